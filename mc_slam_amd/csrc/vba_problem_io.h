// vba_problem_io.h -- the on-disk problem format "VBAP" v2 (v1 files still load) (include/vislam_ba.h: vba_problem_save / _load / _free), plain C++17.
// Compiled into libvislam_ba.so (vislam_ba.hip) and into the sanitizer harness tests/host_structure_check.cpp.
#pragma once
#include "../../include/vislam_ba.h"

#include <algorithm>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

// (the three entry points are declared extern "C" in vislam_ba.h; this header DEFINES them: include it in exactly one
// translation unit of a binary)
// ---- on-disk problem format ----
namespace vba_io {
// v2 = v1 + two ints behind has_kf_fix: solver (VBA_SOLVER_*) and a reserved zero (keeps the doubles 8-byte aligned).  The
// writer always writes v2; the reader takes both (a v1 file has solver = VBA_SOLVER_LDLT, the only solver there was).
struct ProblemFileHeaderV1 {
    char magic[4];
    uint32_t version;
    int32_t variant, n_kf, n_kf_free, n_pt, n_obs, n_imu, algo, its_stage1, its_stage2, protocol, robust, has_kf_fix;
    double K[4], T_cb[7], g_w[3], inv_bg_rw2, inv_ba_rw2, huber_vis, huber_prv, huber_bias, chi2_th, depth_min, rho_min;
};
struct ProblemFileHeader {
    char magic[4];
    uint32_t version;
    int32_t variant, n_kf, n_kf_free, n_pt, n_obs, n_imu, algo, its_stage1, its_stage2, protocol, robust, has_kf_fix, solver, reserved0;
    double K[4], T_cb[7], g_w[3], inv_bg_rw2, inv_ba_rw2, huber_vis, huber_prv, huber_bias, chi2_th, depth_min, rho_min;
};
static_assert(sizeof(ProblemFileHeaderV1) == 56 + 22 * 8 && sizeof(ProblemFileHeader) == 64 + 22 * 8, "VBAP headers are packed");
struct ArrSpec { size_t off; size_t bytes; };
// the arrays of a vba_problem in struct order: (pointer member offset, byte size)
inline std::vector<ArrSpec> problem_arrays(const ProblemFileHeader& hd) {
    const size_t kf = hd.n_kf, pt = hd.n_pt, ob = hd.n_obs, im = hd.n_imu;
    std::vector<ArrSpec> v = {
        {offsetof(vba_problem, kf_pose), kf * 7 * 8}, {offsetof(vba_problem, kf_vel), kf * 3 * 8}, {offsetof(vba_problem, kf_bias), kf * 12 * 8},
        {offsetof(vba_problem, pt), pt * 3 * 8}, {offsetof(vba_problem, pt_ref_kf), pt * 4}, {offsetof(vba_problem, pt_obs_begin), (pt + 1) * 4},
        {offsetof(vba_problem, obs_kf), ob * 4}, {offsetof(vba_problem, obs_uv), ob * 2 * 8}, {offsetof(vba_problem, obs_w), ob * 8},
        {offsetof(vba_problem, imu_kf_i), im * 4}, {offsetof(vba_problem, imu_kf_j), im * 4},
        {offsetof(vba_problem, imu_meas), im * VBA_IMU_MEAS_STRIDE * 8}, {offsetof(vba_problem, imu_info_prv), im * 81 * 8},
        {offsetof(vba_problem, kf_fix), hd.has_kf_fix ? kf : 0}};
    return v;
}
}  // namespace vba_io
using vba_io::ProblemFileHeader; using vba_io::ProblemFileHeaderV1; using vba_io::ArrSpec; using vba_io::problem_arrays;

int vba_problem_save(const char* path, const vba_problem* p) {
    if (!path || !p) return -1;
    ProblemFileHeader hd;
    std::memset(&hd, 0, sizeof hd);
    std::memcpy(hd.magic, "VBAP", 4);
    hd.version = 2;
    hd.variant = p->variant; hd.n_kf = p->n_kf; hd.n_kf_free = p->n_kf_free; hd.n_pt = p->n_pt; hd.n_obs = p->n_obs; hd.n_imu = p->n_imu;
    hd.algo = p->algo; hd.its_stage1 = p->its_stage1; hd.its_stage2 = p->its_stage2; hd.protocol = p->protocol; hd.robust = p->robust;
    hd.has_kf_fix = p->kf_fix ? 1 : 0;
    hd.solver = p->solver;
    std::memcpy(hd.K, p->K, sizeof hd.K); std::memcpy(hd.T_cb, p->T_cb, sizeof hd.T_cb); std::memcpy(hd.g_w, p->g_w, sizeof hd.g_w);
    hd.inv_bg_rw2 = p->inv_bg_rw2; hd.inv_ba_rw2 = p->inv_ba_rw2; hd.huber_vis = p->huber_vis; hd.huber_prv = p->huber_prv;
    hd.huber_bias = p->huber_bias; hd.chi2_th = p->chi2_th; hd.depth_min = p->depth_min; hd.rho_min = p->rho_min;
    FILE* f = std::fopen(path, "wb");
    if (!f) return -2;
    bool ok = std::fwrite(&hd, sizeof hd, 1, f) == 1;
    static const char zeros[16] = {0};
    for (const ArrSpec& a : problem_arrays(hd)) {
        const void* src = *reinterpret_cast<void* const*>(reinterpret_cast<const char*>(p) + a.off);
        if (a.bytes) {
            if (src) ok = ok && std::fwrite(src, 1, a.bytes, f) == a.bytes;
            else for (size_t i = 0; i < a.bytes; i += 16) ok = ok && std::fwrite(zeros, 1, std::min<size_t>(16, a.bytes - i), f) > 0;  // absent optional array
        }
    }
    ok = (std::fclose(f) == 0) && ok;
    return ok ? 0 : -3;
}

int vba_problem_load(const char* path, vba_problem** out) {
    if (!path || !out) return -1;
    *out = nullptr;
    FILE* f = std::fopen(path, "rb");
    if (!f) return -2;
    ProblemFileHeader hd;
    std::memset(&hd, 0, sizeof hd);
    size_t hd_bytes = sizeof hd;
    bool hok = std::fread(&hd, 8, 1, f) == 1 && std::memcmp(hd.magic, "VBAP", 4) == 0 && (hd.version == 1 || hd.version == 2);
    if (hok && hd.version == 2) hok = std::fread(reinterpret_cast<char*>(&hd) + 8, sizeof hd - 8, 1, f) == 1;
    else if (hok) {   // v1: the same fields without solver / reserved0
        ProblemFileHeaderV1 h1;
        hd_bytes = sizeof h1;
        hok = std::fread(reinterpret_cast<char*>(&h1) + 8, sizeof h1 - 8, 1, f) == 1;
        if (hok) {
            std::memcpy(&hd.variant, &h1.variant, 12 * sizeof(int32_t));
            hd.solver = VBA_SOLVER_LDLT; hd.reserved0 = 0;
            std::memcpy(hd.K, h1.K, 22 * sizeof(double));
        }
    }
    if (!hok || hd.n_kf < 0 || hd.n_pt < 0 || hd.n_obs < 0 || hd.n_imu < 0 || hd.n_kf_free < 0 || hd.n_kf_free > hd.n_kf || hd.reserved0 != 0 ||
        (hd.solver != VBA_SOLVER_LDLT && hd.solver != VBA_SOLVER_PCG)) {
        std::fclose(f);
        return -3;
    }
    const std::vector<ArrSpec> arrs = problem_arrays(hd);
    {   // the header must describe exactly the bytes that follow it: nothing is allocated for a file that lies about its sizes
        unsigned long long want = hd_bytes;
        for (const ArrSpec& a : arrs) want += a.bytes;
        const long here = std::ftell(f);
        if (here < 0 || std::fseek(f, 0, SEEK_END) != 0) { std::fclose(f); return -3; }
        const long end = std::ftell(f);
        if (end < 0 || (unsigned long long)end != want || std::fseek(f, here, SEEK_SET) != 0) { std::fclose(f); return -3; }
    }
    size_t total = (sizeof(vba_problem) + 15) / 16 * 16;
    for (const ArrSpec& a : arrs) total += (a.bytes + 15) / 16 * 16;
    char* blk = static_cast<char*>(std::calloc(1, total));
    if (!blk) { std::fclose(f); return -4; }
    vba_problem* p = reinterpret_cast<vba_problem*>(blk);
    p->variant = hd.variant; p->n_kf = hd.n_kf; p->n_kf_free = hd.n_kf_free; p->n_pt = hd.n_pt; p->n_obs = hd.n_obs; p->n_imu = hd.n_imu;
    p->algo = hd.algo; p->its_stage1 = hd.its_stage1; p->its_stage2 = hd.its_stage2; p->protocol = hd.protocol; p->robust = hd.robust;
    p->solver = hd.solver;
    std::memcpy(p->K, hd.K, sizeof hd.K); std::memcpy(p->T_cb, hd.T_cb, sizeof hd.T_cb); std::memcpy(p->g_w, hd.g_w, sizeof hd.g_w);
    p->inv_bg_rw2 = hd.inv_bg_rw2; p->inv_ba_rw2 = hd.inv_ba_rw2; p->huber_vis = hd.huber_vis; p->huber_prv = hd.huber_prv;
    p->huber_bias = hd.huber_bias; p->chi2_th = hd.chi2_th; p->depth_min = hd.depth_min; p->rho_min = hd.rho_min;
    size_t off = (sizeof(vba_problem) + 15) / 16 * 16;
    bool ok = true;
    for (const ArrSpec& a : arrs) {
        void** slot = reinterpret_cast<void**>(blk + a.off);
        if (a.bytes) {
            *slot = blk + off;
            ok = ok && std::fread(blk + off, 1, a.bytes, f) == a.bytes;
            off += (a.bytes + 15) / 16 * 16;
        } else
            *slot = nullptr;
    }
    ok = ok && std::fgetc(f) == EOF;   // nothing may follow the last array
    std::fclose(f);
    if (!ok) { std::free(blk); return -5; }
    // the CSR must be consistent before anybody indexes with it
    if (p->n_pt > 0 && (p->pt_obs_begin[0] != 0 || p->pt_obs_begin[p->n_pt] != p->n_obs)) { std::free(blk); return -6; }
    *out = p;
    return 0;
}

void vba_problem_free(vba_problem* p) { std::free(p); }

