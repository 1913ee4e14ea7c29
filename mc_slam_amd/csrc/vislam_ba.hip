// vislam_ba.hip -- C-ABI (include/vislam_ba.h) of the MI355X local-BA backend: handle, upload (H2D +
// structure build), the lock-step launch schedule of the two-stage solve, download.
//
// Host-side control flow restated from src/Optimizer.cpp:453-517 (two-stage protocol) and
// Thirdparty/g2o/g2o/core/sparse_optimizer.cpp:354-419 (optimize loop); all per-iteration decisions are taken
// on the device (k_ctrl_*), the host only enqueues.  No CPU fallback exists: without a HIP device every entry
// point fails with an error.
#include "../../include/vislam_ba.h"
#include "vba_host_structure.h"
#include "vba_problem_io.h"
#include "vba_kernels_lm.h"
#include "vba_preint.h"
#include "vba_pose.h"
#include "vba_structure.h"
#include "vba_pcg.h"
#include "vba_chain.h"

#include <sched.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <atomic>
#include <memory>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

namespace {

// std::vector whose resize() leaves new elements uninitialised: the concatenated upload arrays are grown first and filled
// by the packing threads afterwards, so every byte is touched once
template <typename T>
struct NoInitAlloc : std::allocator<T> {
    template <typename U> struct rebind { using other = NoInitAlloc<U>; };
    template <typename U, typename... A>
    void construct(U* q, A&&... a) {
        if constexpr (sizeof...(A) == 0) ::new (static_cast<void*>(q)) U;
        else ::new (static_cast<void*>(q)) U(std::forward<A>(a)...);
    }
};
template <typename T> using hvec = std::vector<T, NoInitAlloc<T>>;

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    void* view = nullptr;     // small batches: the array lives inside the upload arena (one H2D for all of them); not owned
    size_t view_bytes = 0;
    void* ptr() const { return view ? view : p; }
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct PinnedBuf {   // persistent pinned host staging (grown on demand)
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = bytes + bytes / 8 + 4096;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
};

// Growable array in pinned host memory with the few std::vector members the upload / download code uses.  The staging
// arrays of a handle persist from call to call, so the H2D / D2H copies are true DMA transfers (no pageable bounce
// buffer) and run concurrently with the kernels of other streams; growth (rare after the first call) re-allocates.
template <typename T>
struct PinVec {
    typedef T value_type;
    T* p = nullptr;
    size_t n = 0, cap = 0;
    bool ok = true;   // false after a failed allocation (checked once per upload / download)
    void reserve(size_t want) {
        if (want <= cap) return;
        const size_t nc = want + want / 4 + 1024;
        void* q = nullptr;
        if (hipHostMalloc(&q, nc * sizeof(T), hipHostMallocDefault) != hipSuccess) { ok = false; return; }
        if (n) memcpy(q, p, n * sizeof(T));
        if (p) (void)hipHostFree(p);
        p = reinterpret_cast<T*>(q);
        cap = nc;
    }
    void resize(size_t m) {
        reserve(m);
        if (m <= cap) n = m;
    }
    void clear() { n = 0; }
    T* data() { return p; }
    const T* data() const { return p; }
    T& operator[](size_t i) { return p[i]; }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        n = cap = 0;
    }
};

// pinned staging of vba_batch_upload (the concatenated arrays of a batch) and vba_batch_download
struct Staging {
    PinVec<double> pose, vel, bias, pt, uv, ow, meas, info;
    PinVec<unsigned char> kffix;
    PinVec<int> ptref, ptobs, obskf, imui, imuj, pair_a, pair_b, pimu_begin, pimu;
    PinVec<int> offpair, pairmask;
    PinVec<unsigned long long> lmask;
    PinVec<int> s_int[14];       // pinned copies of the small host-built lists (tile lists, k_lin2 runs, reference-run lists)
    PinVec<WinDesc> s_desc;
    PinVec<double> dl_pose, dl_vel, dl_bias, dl_pt, dl_chi2;
    PinVec<unsigned char> dl_outl;
    template <typename F> void each(F f) {
        f(pose); f(vel); f(bias); f(pt); f(uv); f(ow); f(meas); f(info); f(kffix);
        f(ptref); f(ptobs); f(obskf); f(imui); f(imuj); f(pair_a); f(pair_b); f(pimu_begin); f(pimu);
        f(offpair); f(pairmask); f(lmask); f(s_desc);
        for (auto& v : s_int) f(v);
        f(dl_pose); f(dl_vel); f(dl_bias); f(dl_pt); f(dl_chi2); f(dl_outl);
    }
    bool ok() { bool r = true; each([&](auto& v) { r = r && v.ok; }); return r; }
    void release() { each([](auto& v) { v.release(); }); }
};

enum {
    BUF_DESC, BUF_CTRL, BUF_POSE, BUF_VEL, BUF_BIAS, BUF_KFR, BUF_POSE0, BUF_VEL0, BUF_BIAS0, BUF_POSEBK, BUF_VELBK,
    BUF_BIASBK, BUF_PT, BUF_PT0, BUF_PTBK, BUF_PTREF, BUF_PTOBS, BUF_OBSKF, BUF_OBSPT, BUF_OBSUV, BUF_OBSW, BUF_LVL,
    BUF_CHI2E, BUF_CHI2F, BUF_DEPTH, BUF_EREC, BUF_PREC, BUF_SLOT, BUF_IMUI, BUF_IMUJ, BUF_IMUMEAS, BUF_IMUINFO, BUF_IMUH, BUF_IMUCHI,
    BUF_S, BUF_LF, BUF_YV, BUF_TLSTEP, BUF_TLPAIR, BUF_TLPANB, BUF_TLPAN, BUF_VEC, BUF_BPOSE, BUF_VARACT, BUF_PAIRA, BUF_PAIRB, BUF_ITEMBEG, BUF_ITEMS, BUF_PIMUBEG, BUF_PIMU,
    BUF_PART, BUF_OUTL, BUF_OUTCHI, BUF_LINBLK, BUF_OFFPAIR, BUF_PAIRMASK, BUF_DBG, BUF_CU, BUF_KFFIX, BUF_TLKB, BUF_TLK, BUF_DVEC, BUF_WINV, BUF_SLOTPERM, BUF_PTPERM,
    BUF_LMASK, BUF_KFSEG, BUF_REFSEG, BUF_ITEMMID, BUF_STKEY, BUF_LMORDER, BUF_SLOTOBS, BUF_PTINV, BUF_KEYSEG, BUF_TSLOT, BUF_ADJBEG, BUF_ADJ, BUF_PCGV, BUF_PCGM, BUF_KFDIR, BUF_MASKQ, BUF_SLOTMASK, BUF_REFQ, BUF_PCGS, BUF_IMUJREC, BUF_ALIVE, BUF_SLOTO, BUF_SLOTREF, BUF_SLOTQ, BUF_RECQ, BUF_TSQ, BUF_RECCNT, BUF_RESULTS, BUF_PRUN0, BUF_PREFBEG, BUF_PREFLIST, BUF_CHAINTAB, BUF_N
};

// every kernel launch of a handle is counted (vba_profile.kernel_launches: launches the last run enqueued)
#define VBA_LAUNCH(...) do { h->n_launch++; hipLaunchKernelGGL(__VA_ARGS__); } while (0)

struct ProfEvt {
    int cls;
    hipEvent_t a, b;
};

struct Handle {
    int device = 0;
    hipStream_t stream = nullptr;
    std::vector<hipStream_t> xstreams;  // extra streams: one per window group of a large batch
    // upload (H2D + structure build) and download (D2H) streams: the run stream itself, except for the lanes of
    // vba_batch_solve, which share the parent's four streams -- run x 2, upload, download -- one per hardware queue
    hipStream_t up_stream = nullptr, dl_stream = nullptr;
    bool owns_streams = true;
    std::string err;
    DevBuf buf[BUF_N];
    DevBuf preint;  // arena of vba_preintegrate
    DevBuf pose_arena;  // arena of vba_pose_optimize
    PinnedBuf pose_host_in, pose_host_out;  // its pinned staging: one H2D and one D2H per call
    // small batches (<= 8 windows): every host-built array of an upload goes through ONE pinned arena and ONE H2D copy into one
    // device arena (a single window is ~25 arrays of a few KB to a few 100 KB: 25 copies cost 0.4 ms of queue latency)
    struct Pending { int id; const void* src; size_t bytes; };
    std::vector<Pending> pending;
    bool arena_on = false;
    DevBuf up_arena;
    PinnedBuf up_arena_host;
    Staging stg;   // pinned staging: upload arrays; download: one D2H per array, windows scattered to the callers' arrays by host threads
    std::vector<Handle*> lanes;   // sub-handles of vba_batch_solve (chunks of a large batch in flight concurrently)
    bool is_lane = false;
    Batch B;
    std::vector<WinDesc> desc;
    PinVec<WinCtrl> hctrl;    // the control blocks after a run (pinned: the copy rides on the run's stream)
    std::vector<int> one_sb;      // n_win == 1: the window's step table (first pair of every factorisation step), for StepOne
    PinnedBuf res_host;           // few windows: control blocks + every result array in ONE block (device: BUF_RESULTS), one D2H copy
    size_t res_bytes = 0, res_off[7] = {0, 0, 0, 0, 0, 0, 0};   // ctrl, pose, vel, bias, pt, outlier flags, chi2
    hipEvent_t up_done = nullptr; // recorded behind an upload that was not waited for on the host (vba_solve)
    bool up_pending = false;
    bool dl_prefetched = false;   // few windows: the run left the result arrays in the download staging already
    int n_win = 0;
    bool any_lin_fallback = false;  // an XYZ window of the batch has a landmark with > 256 observations: k_lin_xyz also runs
    int cur_group = 0; // window group being enqueued (its pinned words)
    int regime_n = 0;  // windows of the uploaded batch: decides WHICH kernels run (few-window / many-window variants), so that
                       // cutting the batch into window groups never changes a summation order
    // launch geometry (maxima over the batch)
    int max_pt_blk = 0, max_imu = 0, max_pairs = 0, max_nb = 0, max_obs_blk = 0, max_kf_blk = 0, max_ns_blk = 0;
    int max_nS = 0, max_its[2] = {0, 0}, max_free = 0, max_lin_blk = 0, max_quads = 1, max_offp = 1, max_pan = 0;
    size_t chain_lds = 0;   // dynamic LDS of k_chol_chain (its per-column tile tables)
    int min_nc = 0, max_nc = 0, max_cu = 0, max_chain_rows = 0, max_split = 0;   // chain columns of the batch's windows (k_chol_chain); tiles of its update launch
    std::vector<int> step_grid;  // workgroups per factorisation step (max over the batch)
    std::vector<int> pan_grid;   // panel tiles per step (max over the batch)
    double tile_updates = 0;     // tile-pair updates per factorisation, summed over the batch
    std::vector<int> win_tiles;  // tile products of one factorisation of window w
    std::vector<long long> win_prod_order;  // per window: tile products under the V/Bias-first and the keyframe order (-1: not evaluated)
    int algo = 0, variant = 2, solver = 0;
    volatile int* stop_host = nullptr;  // pinned, device-visible
    int* stop_dev = nullptr;
    bool profile = false;
    int opt_lin_fallback = 0;  // test hook: XYZ windows without the edge-parallel work split
    int opt_ll_min = 0;        // test hook: batch size from which the left-looking factorisation kernels are used (0: VBA_LL_MIN / 256)
    int opt_chunk = 0, opt_lanes = 0;  // > 0: chunk size / lanes of vba_batch_solve (test hook; defaults from VBA_CHUNK, VBA_LANES)
    int opt_streams = 0;  // > 0: window groups / streams for GN batches (test hook; default from VBA_STREAMS, 1)
    int opt_chol_step = 0;    // test hook: 1 = the first form of the fused factorisation step (k_chol_step) instead of k_chol_step4
    long long n_launch = 0;   // kernel launches enqueued through this handle so far
    int opt_no_chain = 0;     // test hook: 1 = one launch per block column everywhere (vba_debug_set_chain)
    int opt_stop_after = -1;  // test hook: >= 0 -- every window reads the stop flag as 1 from that terminate() poll on (poll_stop)
    std::vector<ProfEvt> evts;
    std::vector<hipEvent_t> evt_pool;
    size_t evt_used = 0;
    vba_profile prof;
    bool uploaded = false;
    bool ran = false;
    bool ll_mode = false;  // left-looking factorisation kernels (batch size at upload >= VBA_LL_MIN)
};

#define HIPCHK(h, call)                                                                          \
    do {                                                                                          \
        hipError_t _e = (call);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(_e);                         \
            return -1;                                                                            \
        }                                                                                         \
    } while (0)

int fail(Handle* h, const std::string& m) {
    static std::mutex mu;   // build_structure runs on several host threads during an upload
    std::lock_guard<std::mutex> lk(mu);
    h->err = m;
    return -1;
}

template <typename T>
T* dp(Handle* h, int id) {
    return reinterpret_cast<T*>(h->buf[id].ptr());
}

hipEvent_t get_evt(Handle* h) {
    if (h->evt_used == h->evt_pool.size()) {
        hipEvent_t e;
        (void)hipEventCreate(&e);
        h->evt_pool.push_back(e);
    }
    return h->evt_pool[h->evt_used++];
}

struct ProfScope {
    Handle* h;
    ProfEvt e;
    bool on;
    ProfScope(Handle* hh, int cls) : h(hh), on(hh->profile) {
        if (on) {
            e.cls = cls;
            e.a = get_evt(h);
            e.b = get_evt(h);
            (void)hipEventRecord(e.a, h->stream);
        }
    }
    ~ProfScope() {
        if (on) {
            (void)hipEventRecord(e.b, h->stream);
            h->evts.push_back(e);
        }
    }
};

// ---- structure build, host half: csrc/vba_host_structure.h (plain C++, also compiled into the sanitizer harness of the tests)
using vba_host::Structure;
using vba_host::vpos_host;
using vba_host::now_ms;
int build_structure(Handle* h, const vba_problem* P, Structure& st, bool two_sided = false) {
    std::string err;
    if (vba_host::build_structure(P, st, err, two_sided)) return fail(h, err);
    if (h->opt_lin_fallback && P->variant != VBA_VARIANT_PRV_IDP) st.linblk.clear();   // test hook: the thread-per-landmark linearisation
    return 0;
}

// a pageable std::vector goes through a pinned copy first: a pageable hipMemcpyAsync is a synchronous, staged transfer
template <typename T>
int h2d_vec(Handle* h, int id, const std::vector<T>& v, PinVec<T>& pin) {
    if (h->arena_on) {
        h->pending.push_back({id, v.data(), v.size() * sizeof(T)});
        return 0;
    }
    pin.clear();
    pin.resize(v.size());
    if (!pin.ok) return fail(h, "out of pinned host memory (upload staging)");
    if (!v.empty()) memcpy(pin.data(), v.data(), v.size() * sizeof(T));
    HIPCHK(h, h->buf[id].ensure(std::max<size_t>(v.size() * sizeof(T), 16)));
    if (!v.empty()) HIPCHK(h, hipMemcpyAsync(h->buf[id].p, pin.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, h->up_stream));
    return 0;
}

template <typename V>
int h2d(Handle* h, int id, const V& v) {
    typedef typename V::value_type T;
    if (h->arena_on) {
        h->pending.push_back({id, v.data(), v.size() * sizeof(T)});
        return 0;
    }
    HIPCHK(h, h->buf[id].ensure(std::max<size_t>(v.size() * sizeof(T), 16)));
    if (!v.empty()) HIPCHK(h, hipMemcpyAsync(h->buf[id].p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, h->up_stream));
    return 0;
}
// arena mode: lay the recorded arrays out (256-B aligned), gather them into the pinned arena, one copy, point the views
int h2d_flush(Handle* h) {
    if (!h->arena_on) return 0;
    size_t total = 0;
    for (auto& q : h->pending) total += (std::max<size_t>(q.bytes, 16) + 255) / 256 * 256;
    HIPCHK(h, h->up_arena.ensure(total + 256));
    HIPCHK(h, h->up_arena_host.ensure(total + 256));
    char* hb = reinterpret_cast<char*>(h->up_arena_host.p);
    char* db = reinterpret_cast<char*>(h->up_arena.p);
    size_t off = 0;
    for (auto& q : h->pending) {
        if (q.bytes) memcpy(hb + off, q.src, q.bytes);
        h->buf[q.id].view = db + off;
        h->buf[q.id].view_bytes = std::max<size_t>(q.bytes, 16);
        off += (std::max<size_t>(q.bytes, 16) + 255) / 256 * 256;
    }
    if (total) HIPCHK(h, hipMemcpyAsync(db, hb, total, hipMemcpyHostToDevice, h->up_stream));
    h->pending.clear();
    return 0;
}
int dalloc(Handle* h, int id, size_t bytes) {
    HIPCHK(h, h->buf[id].ensure(std::max<size_t>(bytes, 16)));
    return 0;
}

void quat_to_R_host(const double* q, double* R) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z, twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}


// Host threads of one handle (packing, structure build, scatter).  One process per GPU: the ranks of a node share its cores, so
// the pool is this rank's share -- cores / LOCAL_WORLD_SIZE, at most 16, at least 2 -- unless VBA_UPLOAD_THREADS says otherwise
// (mc_slam_amd/launch.py exports it per rank).  The cores are those the process may run on (sched_getaffinity: a rank pinned to its
// share by the launcher counts only that share).
int host_threads() {
    static const int n = [] {
        if (const char* e = getenv("VBA_UPLOAD_THREADS")) return std::max(1, atoi(e));
        int cores = (int)std::thread::hardware_concurrency();
        cpu_set_t set;
        CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0) cores = CPU_COUNT(&set);
        // ranks of THIS node that share the cores (torchrun exports LOCAL_WORLD_SIZE; WORLD_SIZE counts the ranks of other nodes too
        // and is not used).  A rank counts as pinned to its share only when the launcher says so (mc_slam_amd/launch.py exports
        // VBA_RANK_CPUS with the cores it bound the rank to): a cpuset-limited container also shows fewer cores than the machine
        // has, and there the ranks still share what it shows.
        int local_world = 1;
        if (const char* e = getenv("LOCAL_WORLD_SIZE")) local_world = std::max(1, atoi(e));
        const bool pinned = getenv("VBA_RANK_CPUS") != nullptr;
        const int share = pinned ? cores : std::max(1, cores / local_world);
        return std::max(std::min(2, std::max(1, cores)), std::min(16, share));
    }();
    return n;
}
// >= this many windows: left-looking factorisation kernels, which never modify S (measured: the right-looking pair is faster
// up to ~256 windows)
bool use_left_looking(const Handle* h, int n) {
    static const int left_looking = getenv("VBA_RIGHT_LOOKING") ? 0 : 1;
    static const int ll_min = getenv("VBA_LL_MIN") ? atoi(getenv("VBA_LL_MIN")) : 256;
    return left_looking && n >= (h->opt_ll_min > 0 ? h->opt_ll_min : ll_min);
}

int do_upload(Handle* h, int n, vba_problem* const* probs, bool defer_sync = false) {
    static const bool timing = getenv("VBA_TIMING") != nullptr;
    const double t_begin = now_ms();
    double t_struct = 0;
    if (n <= 0) return fail(h, "empty batch");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->up_stream));   // (an upload that failed half way may still have copies out of the staging in flight)
    h->uploaded = false;
    h->n_win = n;
    h->regime_n = n;
    for (auto& b : h->buf) { b.view = nullptr; b.view_bytes = 0; }
    h->pending.clear();
    static const int arena_max = getenv("VBA_ARENA_MAX") ? atoi(getenv("VBA_ARENA_MAX")) : 8;
    h->arena_on = n <= arena_max;
    h->desc.assign(n, WinDesc());
    h->win_tiles.assign(n, 0);
    h->win_prod_order.assign(3 * (size_t)n, -1);
    Staging& G = h->stg;
    auto &pose = G.pose, &vel = G.vel, &bias = G.bias, &pt = G.pt, &uv = G.uv, &ow = G.ow, &meas = G.meas, &info = G.info;
    auto& kffix = G.kffix;
    auto &ptref = G.ptref, &ptobs = G.ptobs, &obskf = G.obskf, &imui = G.imui, &imuj = G.imuj, &pair_a = G.pair_a, &pair_b = G.pair_b;
    auto &pimu_begin = G.pimu_begin, &pimu = G.pimu;
    auto &offpair = G.offpair, &pairmask = G.pairmask;
    auto& lmask = G.lmask;
    G.each([](auto& v) { v.clear(); });
    std::vector<int> tlstep, tlpair, tlpanb, tlpan, linblk, tlkb, tlk, adjbeg, adj, prun0, prefbeg, preflist, culist, chaintab;
    const bool pcg = probs[0] && probs[0]->solver == VBA_SOLVER_PCG;
    h->step_grid.clear();
    h->pan_grid.clear();
    h->tile_updates = 0;
    size_t S_tot = 0;
    int kf0 = 0, pt0 = 0, obs0 = 0, imu0 = 0, pair0 = 0, pimu0 = 0, vec0 = 0, part0 = 0;
    long long item0 = 0, mask0 = 0;
    int max_kf = 0;
    h->max_free = 0;
    h->max_lin_blk = 0;
    h->max_quads = 1;
    h->max_pan = 0;
    h->max_offp = 1;
    h->max_pt_blk = h->max_imu = h->max_pairs = h->max_nb = h->max_obs_blk = h->max_kf_blk = h->max_ns_blk = h->max_nS = 0;
    h->max_its[0] = h->max_its[1] = 0;
    h->any_lin_fallback = false;
    h->min_nc = 1 << 30; h->max_nc = 0; h->max_cu = 0; h->chain_lds = 0; h->max_chain_rows = 0; h->max_split = 0;
    // Chain columns in one launch (vba_chain.h): in the left-looking regime (two lean launches for all chain columns of all windows),
    // and for up to 64 windows in the right-looking one (one workgroup per tile row walks the chain, the two chains of the two-sided
    // order side by side).  Every row workgroup redoes the chain's diagonal work, which is only free while compute units idle --
    // measured on MI355X, ms per run with / without: 1 window 2.11 / 2.37, 8: 2.71 / 3.06, 16: 3.16 / 3.72, 32: 4.70 / 5.10,
    // 64: 7.44 / 7.52, 96: 10.6 / 10.0, 128: 13.4 / 12.0.  In between: one launch per block column.
    // VBA_NO_CHAIN: A/B switch, one launch per block column everywhere.
    static const int chain_rl_max = getenv("VBA_CHAIN_RL_MAX") ? atoi(getenv("VBA_CHAIN_RL_MAX")) : 64;
    const bool chain_on = getenv("VBA_NO_CHAIN") == nullptr && !h->opt_no_chain && (use_left_looking(h, n) || n <= chain_rl_max);
    // the two-sided V/Bias-first order (vba_host_structure.h, order 2) is a candidate for every window: its two half-length chains leave
    // half the fill in the PR rows (C3: 408 tile products against 581), and the few-window chain kernel walks them side by side.
    // VBA_ONE_CHAIN: A/B switch, orders 0 and 1 only as before.
    const bool two_sided = getenv("VBA_ONE_CHAIN") == nullptr;
    // Per chunk of windows: (1) the per-window structure (item lists, IMU lists, symbolic tile factorisation: 0.7 ms for a C3
    // window) on a pool of host threads, (2) descriptors and offsets in window order on this thread, (3) the concatenated
    // arrays grown once, (4) the pool again copies every window's arrays to its offsets (2.2 MB per C3 window).
    static const int n_threads = host_threads();
    const int chunk = 8 * n_threads;
    size_t tot_kf = 0, tot_pt = 0, tot_obs = 0, tot_mask = 0;
    {   // one allocation per concatenated array instead of the doubling growth of std::vector
        size_t skf = 0, spt = 0, sobs = 0, simu = 0, spair = 0, smask = 0;
        for (int w = 0; w < n; w++) {
            const vba_problem* P = probs[w];
            if (!P || P->n_kf < 0 || P->n_pt < 0 || P->n_obs < 0 || P->n_imu < 0 || P->n_kf_free < 0) continue;
            skf += P->n_kf; spt += P->n_pt; sobs += P->n_obs; simu += P->n_imu;
            spair += (size_t)P->n_kf_free * (P->n_kf_free + 1) / 2;
            smask += (size_t)P->n_pt * (size_t)((P->n_kf + 63) / 64);
        }
        tot_kf = skf; tot_pt = spt; tot_obs = sobs; tot_mask = smask;
        pose.reserve(7 * skf); vel.reserve(3 * skf); bias.reserve(12 * skf); kffix.reserve(skf);
        pt.reserve(3 * spt); ptref.reserve(spt); ptobs.reserve(spt + n); lmask.reserve(smask);
        obskf.reserve(sobs); uv.reserve(2 * sobs); ow.reserve(sobs);
        imui.reserve(simu); imuj.reserve(simu); meas.reserve(61 * simu); info.reserve(81 * simu);
        pair_a.reserve(spair); pair_b.reserve(spair); offpair.reserve(spair); pairmask.reserve(spair);
        pimu_begin.reserve(spair + n);
    }
    const bool pristine = use_left_looking(h, n);
    h->ll_mode = pristine;
    std::vector<Structure> sts;
    auto run_pool = [&](int cn, const std::function<void(int)>& job) {
        std::atomic<int> next(0);
        auto work = [&]() {
            for (int q = next.fetch_add(1); q < cn; q = next.fetch_add(1)) job(q);
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < std::min(n_threads, cn); t++) pool.emplace_back(work);
        work();
        for (auto& t : pool) t.join();
    };
    // The bulk of a window (observations, landmarks, masks: 1 MB of the 1.06 MB of a C3 window) crosses PCIe WHILE the host works on the
    // next windows: after every packing pass the freshly packed tail of these arrays is copied (their final sizes are known from
    // the pre-pass above, the pinned staging is never re-allocated under a copy).  Before: validate + symbolic (5.4 ms per 384
    // windows), pack (5.0), then ONE copy per array (7.5) one after the other -- the link idle for the first half, the host for the second.
    struct IncCopy { int id; const char* base; size_t esz, total, done; };
    static const int inc_off = getenv("VBA_UPLOAD_NO_OVERLAP") ? 1 : 0;
    const bool inc_on = !h->arena_on && !inc_off;
    std::vector<IncCopy> inc;
    if (inc_on) {
        auto reg = [&](int id, const void* base, size_t esz, size_t total) -> int {
            if (dalloc(h, id, total * esz)) return -1;
            inc.push_back({id, reinterpret_cast<const char*>(base), esz, total, 0});
            return 0;
        };
        if (reg(BUF_OBSUV, uv.data(), 16, tot_obs) || reg(BUF_OBSW, ow.data(), 8, tot_obs) || reg(BUF_OBSKF, obskf.data(), 4, tot_obs) ||
            reg(BUF_PT0, pt.data(), 24, tot_pt) || reg(BUF_PTREF, ptref.data(), 4, tot_pt) || reg(BUF_LMASK, lmask.data(), 8, tot_mask) ||
            reg(BUF_POSE0, pose.data(), 56, tot_kf) || reg(BUF_BIAS0, bias.data(), 96, tot_kf) || reg(BUF_VEL0, vel.data(), 24, tot_kf)) return -1;
    }
    auto inc_push = [&](size_t kf_now, size_t pt_now, size_t obs_now, size_t mask_now) -> int {
        for (auto& a : inc) {
            const size_t now = (a.id == BUF_OBSUV || a.id == BUF_OBSW || a.id == BUF_OBSKF) ? obs_now
                             : (a.id == BUF_PT0 || a.id == BUF_PTREF) ? pt_now : (a.id == BUF_LMASK) ? mask_now : kf_now;
            if (now > a.done) {
                if (now > a.total) return fail(h, "internal: incremental upload past the reserved size");
                HIPCHK(h, hipMemcpyAsync(reinterpret_cast<char*>(h->buf[a.id].p) + a.done * a.esz, a.base + a.done * a.esz, (now - a.done) * a.esz,
                                         hipMemcpyHostToDevice, h->up_stream));
                a.done = now;
            }
        }
        return 0;
    };
    for (int chunk0 = 0; chunk0 < n; chunk0 += chunk) {
        const int cn = std::min(chunk, n - chunk0);
        for (int q = 0; q < cn; q++)
            if (!probs[chunk0 + q]) return fail(h, "null problem");
        // (1)
        sts.assign(cn, Structure());
        std::atomic<int> bad(0);
        const double ts0 = now_ms();
        run_pool(cn, [&](int q) {
            const vba_problem* Q = probs[chunk0 + q];
            if (Q->variant < 0 || Q->variant > 2 || Q->n_kf_free <= 0 || Q->n_kf_free > Q->n_kf || Q->n_pt <= 0 || Q->n_obs <= 0 || Q->n_imu < 0) return;  // reported below
            if (build_structure(h, Q, sts[q], two_sided)) bad.store(1);
        });
        t_struct += now_ms() - ts0;
        if (bad.load()) return -1;
        if (chunk0 == 0) {   // tile lists: extrapolate from the first chunk
            size_t tp = 0;
            for (auto& x : sts) tp += x.tpairs.size() + x.klist.size();
            tlpair.reserve((size_t)(1.1 * tp / cn * n) + 1024); tlk.reserve((size_t)(1.1 * tp / cn * n) + 1024);
        }
        // (2)
        for (int w = chunk0; w < chunk0 + cn; w++) {
            const vba_problem* P = probs[w];
            if (P->variant < 0 || P->variant > 2 || (P->algo != VBA_ALGO_GN && P->algo != VBA_ALGO_LM)) return fail(h, "bad variant / algo");
            if (P->variant == VBA_VARIANT_PRV_IDP && P->algo != VBA_ALGO_GN)
                return fail(h, "inverse-depth landmarks are solved with Gauss-Newton only (as the reference does, src/Optimizer.cpp:136)");
            if (P->variant != VBA_VARIANT_PRV_IDP && P->algo != VBA_ALGO_LM)
                return fail(h, "XYZ landmarks are solved with Levenberg-Marquardt only (as the reference does, src/Optimizer.cpp:1028,3928)");
            if (P->n_kf_free <= 0 || P->n_kf_free > P->n_kf || P->n_pt < 0 || P->n_obs < 0 || P->n_imu < 0) return fail(h, "bad sizes");
            if (P->variant != VBA_VARIANT_SE3_XYZ && P->n_imu > 0 && (!P->imu_kf_i || !P->imu_kf_j || !P->imu_meas || !P->imu_info_prv))
                return fail(h, "n_imu > 0 but an IMU array is NULL");
            if (P->n_pt == 0 || P->n_obs == 0) return fail(h, "a window without landmarks or observations has nothing to optimise");
            if (w > 0 && (P->variant != probs[0]->variant || P->algo != probs[0]->algo || P->solver != probs[0]->solver)) return fail(h, "mixed batch");
            if (P->solver != VBA_SOLVER_LDLT && P->solver != VBA_SOLVER_PCG) return fail(h, "unknown solver");
            if (P->its_stage1 > 30 || P->its_stage2 > 30 || P->its_stage1 < 0 || P->its_stage2 < 0) return fail(h, "its out of range");
            if (P->protocol != VBA_PROTO_LOCAL && P->protocol != VBA_PROTO_SINGLE) return fail(h, "unknown protocol");
            WinDesc& d = h->desc[w];
            d.variant = P->variant; d.algo = P->algo;
            d.protocol = P->protocol; d.robust = P->robust;
            d.win = w;
            d.n_kf = P->n_kf; d.n_free = P->n_kf_free; d.n_pt = P->n_pt; d.n_obs = P->n_obs;
            d.n_imu = (P->variant == VBA_VARIANT_SE3_XYZ) ? 0 : P->n_imu;
            d.pdim = (P->variant == VBA_VARIANT_SE3_XYZ) ? 6 : 15;
            d.np = d.pdim * d.n_free;
            d.nS = sts[w - chunk0].nS;           // (the order decides: the two-sided order pads each of its parts to a tile boundary)
            d.nb = d.nS / VBA_NB;
            d.its[0] = P->its_stage1; d.its[1] = P->its_stage2;
            d.kf0 = kf0; d.pt0 = pt0; d.obs0 = obs0; d.imu0 = imu0;
            d.pair0 = pair0; d.n_pairs = d.n_free * (d.n_free + 1) / 2;
            Structure& st = sts[w - chunk0];
            d.item0 = (int)item0; d.pimu0 = pimu0; d.vec0 = vec0; d.part0 = part0;
            d.mask0 = mask0; d.mwords = st.mwords;
            d.adj0 = (int)adj.size();
            if (pcg) {
                adjbeg.resize((size_t)kf0 + w, 0);   // rows of adj_begin start at kf0 + win, like the keyframe segments
                adjbeg.insert(adjbeg.end(), st.adj_begin.begin(), st.adj_begin.end());
                adj.insert(adj.end(), st.adj.begin(), st.adj.end());
            }
            d.n_part_pt = (d.n_pt + 63) / 64;
            d.n_part_lin = d.n_part_pt;
            d.lin_runs = 0;
            if (!st.linblk.empty()) {   // the work split of the edge-parallel linearisation
                d.lb0 = (int)(linblk.size() / 4);
                linblk.insert(linblk.end(), st.linblk.begin(), st.linblk.end());
                d.n_part_lin = (int)(st.linblk.size() / 4);
                d.lin_runs = 1;
                if (!st.prun0.empty()) {   // inverse depth: the run records of the reference-keyframe terms (ids window-local)
                    prun0.resize((size_t)d.lb0, 0);
                    prun0.insert(prun0.end(), st.prun0.begin(), st.prun0.end() - 1);
                    prefbeg.resize((size_t)kf0 + w, 0);   // rows start at kf0 + win, like the keyframe segments
                    prefbeg.insert(prefbeg.end(), st.pref_begin.begin(), st.pref_begin.end());
                    preflist.resize((size_t)pt0, 0);      // a window has at most n_pt run records: its list starts at pt0
                    preflist.insert(preflist.end(), st.pref_list.begin(), st.pref_list.end());
                }
            } else
                h->any_lin_fallback = true;
            d.S0 = (long long)S_tot;
            for (int i = 0; i < 4; i++) d.K[i] = P->K[i];
            quat_to_R_host(P->T_cb + 3, d.Rcb);
            for (int i = 0; i < 3; i++) { d.tcb[i] = P->T_cb[i]; d.g[i] = P->g_w[i]; }
            d.inv_bg = P->inv_bg_rw2; d.inv_ba = P->inv_ba_rw2;
            d.hub_vis = P->huber_vis; d.hub_prv = P->huber_prv; d.hub_bias = P->huber_bias;
            d.chi2_th = P->chi2_th; d.depth_min = P->depth_min; d.rho_min = P->rho_min;
            d.tl_step0 = (int)tlstep.size(); d.tl_pair0 = (int)tlpair.size(); d.tl_pan0 = (int)tlpan.size();
            tlstep.insert(tlstep.end(), st.step_begin.begin(), st.step_begin.end());
            tlpanb.insert(tlpanb.end(), st.pan_begin.begin(), st.pan_begin.end());
            tlpair.insert(tlpair.end(), st.tpairs.begin(), st.tpairs.end());
            tlpan.insert(tlpan.end(), st.pan.begin(), st.pan.end());
            d.order = st.order;
            d.vp_h = 2147483647; d.vp_vb1 = 0;
            for (int q = 0; q < 3; q++) { d.pad0[q] = 0; d.padn[q] = 0; }
            d.pad0[0] = d.np; d.padn[0] = d.nS - d.np;
            if (d.pdim != 15) { d.vp_pr0 = 0; d.vp_prs = 6; d.vp_vb0 = 0; d.vp_vbs = 0; }
            else if (d.order == 2) {
                int hh, baseB, pr0;
                vba_host::two_sided_layout(d.n_free, hh, baseB, pr0);
                d.vp_pr0 = pr0; d.vp_prs = 6; d.vp_vb0 = 0; d.vp_vbs = 9; d.vp_h = hh; d.vp_vb1 = baseB + 9 * (d.n_free - 1);
                d.pad0[0] = 9 * hh; d.padn[0] = baseB - 9 * hh;
                d.pad0[1] = baseB + 9 * (d.n_free - hh); d.padn[1] = pr0 - d.pad0[1];
                d.pad0[2] = pr0 + 6 * d.n_free; d.padn[2] = d.nS - d.pad0[2];
            }
            else if (d.order) { d.vp_pr0 = 0; d.vp_prs = 15; d.vp_vb0 = 6; d.vp_vbs = 15; }
            else { d.vp_pr0 = 9 * d.n_free; d.vp_prs = 6; d.vp_vb0 = 0; d.vp_vbs = 9; }
            d.nc_split = (chain_on && st.nc > 0) ? st.nc_split : 0;
            h->max_split = std::max(h->max_split, d.nc_split);
            d.tl_kb0 = (int)tlkb.size(); d.tl_k0 = (int)tlk.size();
            tlkb.insert(tlkb.end(), st.kl_begin.begin(), st.kl_begin.end());
            tlk.insert(tlk.end(), st.klist.begin(), st.klist.end());
            d.nc = chain_on ? st.nc : 0;
            d.cu0 = (int)(culist.size() / 4); d.n_cu = d.nc > 0 ? (int)(st.cu.size() / 4) : 0;
            if (d.nc > 0) culist.insert(culist.end(), st.cu.begin(), st.cu.end());
            d.ct0 = (int)(chaintab.size() / 4);
            if (d.nc > 0) chaintab.insert(chaintab.end(), st.chain_tab.begin(), st.chain_tab.end());
            h->min_nc = std::min(h->min_nc, d.nc); h->max_nc = std::max(h->max_nc, d.nc); h->max_cu = std::max(h->max_cu, d.n_cu);
            h->max_chain_rows = std::max(h->max_chain_rows, d.nc > 0 ? d.nb - d.nc : 0);
            h->chain_lds = std::max(h->chain_lds, ((size_t)d.nc * (d.nb - d.nc) + 2 * (size_t)d.nc + 8) * sizeof(short));   // chain_tab_bytes
            if ((int)h->step_grid.size() < d.nb) { h->step_grid.resize(d.nb, 1); h->pan_grid.resize(d.nb, 0); }
            for (int k = 0; k < d.nb; k++) {
                h->step_grid[k] = std::max(h->step_grid[k], std::max(1, st.step_npairs[k]));
                h->pan_grid[k] = std::max(h->pan_grid[k], st.pan_begin[k + 1] - st.pan_begin[k]);
            }
            h->tile_updates += (double)st.tpairs.size();
            h->win_tiles[w] = (int)st.tpairs.size();
            if (n == 1) h->one_sb = st.step_begin;
            for (int q = 0; q < 3; q++) h->win_prod_order[3 * (size_t)w + q] = st.prod_order[q];
            if ((int)st.pair_a.size() != d.n_pairs || (int)st.off_pair.size() != d.n_pairs || (int)st.pair_mask.size() != d.n_pairs ||
                (int)st.pimu_begin.size() != d.n_pairs + 1 || st.lmask.size() != (size_t)d.n_pt * st.mwords)
                return fail(h, "internal: structure sizes");
            {   // the per-window offsets are 32-bit: refuse a batch that would overflow them instead of wrapping
                const long long lim = 2147483647LL - 64;
                if ((long long)obs0 + d.n_obs > lim || item0 + st.item_cap > lim ||
                    (long long)tlpair.size() > lim || (long long)tlk.size() > lim || (long long)vec0 + d.nS > lim)
                    return fail(h, "batch too large for 32-bit offsets: split it into several calls");
            }
            kf0 += d.n_kf; pt0 += d.n_pt; obs0 += d.n_obs; imu0 += d.n_imu;
            pair0 += d.n_pairs; item0 += st.item_cap; pimu0 += (int)(st.pimu.size() / 2);
            mask0 += (long long)d.n_pt * st.mwords;
            max_kf = std::max(max_kf, d.n_kf);
            vec0 += d.nS;
            const int obs_blk = (d.n_obs + 63) / 64;
            // chi2 / computeScale / max-diagonal partials of the linearisation and update kernels, the per-block sums of the final edge
            // pass -- and, with PCG, one p'Sp partial per PCG_ROWS rows of the reduced system (k_pcg_matvec), which grows with the
            // KEYFRAMES of the window, not with its landmarks
            part0 += std::max(std::max(3 * std::max(d.n_part_lin, (d.n_pt + 63) / 64), 2 * obs_blk), pcg ? (d.np + PCG_ROWS - 1) / PCG_ROWS : 0) + 2;
            S_tot += (size_t)d.nS * d.nS;
            h->max_pt_blk = std::max(h->max_pt_blk, (d.n_pt + 63) / 64);
            h->max_lin_blk = std::max(h->max_lin_blk, d.n_part_lin);
            h->max_imu = std::max(h->max_imu, d.n_imu);
            h->max_pairs = std::max(h->max_pairs, d.n_pairs);
            h->max_free = std::max(h->max_free, d.n_free);
            h->max_quads = std::max(h->max_quads, (d.n_pairs - d.n_free + 3) / 4);
            h->max_pan = std::max(h->max_pan, (int)st.pan.size());
            h->max_offp = std::max(h->max_offp, d.n_pairs - d.n_free);
            h->max_nb = std::max(h->max_nb, d.nb);
            h->max_obs_blk = std::max(h->max_obs_blk, obs_blk);
            h->max_kf_blk = std::max(h->max_kf_blk, (d.n_kf + 63) / 64);
            h->max_ns_blk = std::max(h->max_ns_blk, (d.nS + 63) / 64);
            h->max_nS = std::max(h->max_nS, d.nS);
            h->max_its[0] = std::max(h->max_its[0], d.its[0]);
            h->max_its[1] = std::max(h->max_its[1], d.its[1]);
        }
        // (3)
        const int nw = chunk0 + cn;   // windows packed so far
        pose.resize(7 * (size_t)kf0); vel.resize(3 * (size_t)kf0); bias.resize(12 * (size_t)kf0); kffix.resize(kf0);
        pt.resize(3 * (size_t)pt0); ptref.resize(pt0); lmask.resize((size_t)mask0); ptobs.resize((size_t)pt0 + nw);
        obskf.resize(obs0); uv.resize(2 * (size_t)obs0); ow.resize(obs0);
        imui.resize(imu0); imuj.resize(imu0); meas.resize(61 * (size_t)imu0); info.resize(81 * (size_t)imu0);
        pair_a.resize(pair0); pair_b.resize(pair0); offpair.resize(pair0); pairmask.resize(pair0);
        pimu_begin.resize((size_t)pair0 + nw);
        pimu.resize(2 * (size_t)pimu0);
        if (!G.ok()) return fail(h, "out of pinned host memory (upload staging)");
        // (4)
        run_pool(cn, [&](int q) {
            const int w = chunk0 + q;
            const vba_problem* P = probs[w];
            const WinDesc& d = h->desc[w];
            Structure& st = sts[q];
            auto put = [](auto* dst, const auto* src, size_t cnt) { if (cnt) memcpy(dst, src, cnt * sizeof(*dst)); };
            put(pose.data() + 7 * (size_t)d.kf0, P->kf_pose, 7 * (size_t)d.n_kf);
            for (int k = 0; k < d.n_kf; k++) kffix[d.kf0 + k] = P->kf_fix ? (unsigned char)(P->kf_fix[k] & 7) : 0;
            if (P->kf_vel) put(vel.data() + 3 * (size_t)d.kf0, P->kf_vel, 3 * (size_t)d.n_kf);
            else std::fill_n(vel.data() + 3 * (size_t)d.kf0, 3 * (size_t)d.n_kf, 0.0);
            if (P->kf_bias) put(bias.data() + 12 * (size_t)d.kf0, P->kf_bias, 12 * (size_t)d.n_kf);
            else std::fill_n(bias.data() + 12 * (size_t)d.kf0, 12 * (size_t)d.n_kf, 0.0);
            put(pt.data() + 3 * (size_t)d.pt0, P->pt, 3 * (size_t)d.n_pt);
            if (P->pt_ref_kf) put(ptref.data() + d.pt0, P->pt_ref_kf, d.n_pt);
            else std::fill_n(ptref.data() + d.pt0, d.n_pt, 0);
            put(ptobs.data() + d.pt0 + w, P->pt_obs_begin, (size_t)d.n_pt + 1);
            put(obskf.data() + d.obs0, P->obs_kf, d.n_obs);
            put(lmask.data() + d.mask0, st.lmask.data(), st.lmask.size());
            put(uv.data() + 2 * (size_t)d.obs0, P->obs_uv, 2 * (size_t)d.n_obs);
            put(ow.data() + d.obs0, P->obs_w, d.n_obs);
            if (d.n_imu) {
                put(imui.data() + d.imu0, P->imu_kf_i, d.n_imu);
                put(imuj.data() + d.imu0, P->imu_kf_j, d.n_imu);
                put(meas.data() + 61 * (size_t)d.imu0, P->imu_meas, 61 * (size_t)d.n_imu);
                put(info.data() + 81 * (size_t)d.imu0, P->imu_info_prv, 81 * (size_t)d.n_imu);
            }
            put(pair_a.data() + d.pair0, st.pair_a.data(), d.n_pairs);
            put(pair_b.data() + d.pair0, st.pair_b.data(), d.n_pairs);
            put(offpair.data() + d.pair0, st.off_pair.data(), d.n_pairs);
            for (int pi = 0; pi < d.n_pairs; pi++) {
                const bool has_items = (st.pair_mask[pi] & 16) != 0;
                st.pair_mask[pi] &= 15;
                // S stays pristine: a sub-block nothing is ever added to keeps the zero of the upload -- without an IMU edge only
                // the 6x6 PR block of a pair is ever written, and a pair without shared landmarks is not written at all
                if (pristine && st.pair_a[pi] != st.pair_b[pi] && st.pimu_begin[pi + 1] == st.pimu_begin[pi]) st.pair_mask[pi] &= has_items ? 1 : 0;
            }
            put(pairmask.data() + d.pair0, st.pair_mask.data(), d.n_pairs);
            put(pimu_begin.data() + d.pair0 + w, st.pimu_begin.data(), (size_t)d.n_pairs + 1);
            put(pimu.data() + 2 * (size_t)d.pimu0, st.pimu.data(), st.pimu.size());
        });
        if (inc_on && inc_push((size_t)kf0, (size_t)pt0, (size_t)obs0, (size_t)mask0)) return -1;
    }
    if (inc_on && (uv.data() != reinterpret_cast<const double*>(inc[0].base) || pt.data() != reinterpret_cast<const double*>(inc[3].base)))
        return fail(h, "internal: the upload staging moved under an incremental copy");
    h->algo = probs[0]->algo;
    h->variant = probs[0]->variant;
    if (h->chain_lds > 40 * 1024 || h->max_nc > 256) {   // (256: CHAIN_MAX_NC)   // (a window whose tile tables do not fit beside the kernel's 53 KB of tiles: one launch per column)
        for (auto& d : h->desc) { d.nc = 0; d.n_cu = 0; }
        h->min_nc = h->max_nc = h->max_cu = 0;
    }
    if (!pcg) {   // the back-substitution keeps x, its solve blocks and the window's tile lists in LDS (160 KiB per workgroup)
        const size_t shm = ((size_t)h->max_nS + 2 * TRSV_P_DW * 32 + 2 * 32 * 65 + 32) * sizeof(double) + ((size_t)h->max_pan + h->max_nb + 2) * sizeof(int);
        if (shm > 160 * 1024) return fail(h, "window too large for the direct solver (back-substitution workspace > 160 KiB of LDS): use VBA_SOLVER_PCG");
    }
    const double t_pack = now_ms();
    // pads of S: identity on the padded diagonal, written once (the solve never touches them)
    if (h2d_vec(h, BUF_DESC, h->desc, G.s_desc)) return -1;
    if (dalloc(h, BUF_CTRL, sizeof(WinCtrl) * n)) return -1;
    if (!inc_on && (h2d(h, BUF_POSE0, pose) || h2d(h, BUF_VEL0, vel) || h2d(h, BUF_BIAS0, bias) || h2d(h, BUF_PT0, pt))) return -1;
    if (h2d(h, BUF_KFFIX, kffix)) return -1;
    if (dalloc(h, BUF_POSE, pose.size() * 8) || dalloc(h, BUF_VEL, vel.size() * 8) || dalloc(h, BUF_BIAS, bias.size() * 8)) return -1;
    if (dalloc(h, BUF_POSEBK, pose.size() * 8) || dalloc(h, BUF_VELBK, vel.size() * 8) || dalloc(h, BUF_BIASBK, bias.size() * 8)) return -1;
    if (dalloc(h, BUF_KFR, (size_t)kf0 * 12 * 8) || dalloc(h, BUF_PT, pt.size() * 8) || dalloc(h, BUF_PTBK, pt.size() * 8)) return -1;
    if (h2d(h, BUF_PTOBS, ptobs)) return -1;
    if (!inc_on && (h2d(h, BUF_PTREF, ptref) || h2d(h, BUF_OBSKF, obskf) || h2d(h, BUF_LMASK, lmask))) return -1;
    // built on the device (vba_structure.h): record orders, keyframe segments, item lists; + the scratch of the build
    if (dalloc(h, BUF_OBSPT, (size_t)obs0 * 4) || dalloc(h, BUF_SLOTPERM, (size_t)obs0 * 4) || dalloc(h, BUF_PTPERM, (size_t)pt0 * 4)) return -1;
    if (dalloc(h, BUF_KFSEG, ((size_t)kf0 + n) * 4) || dalloc(h, BUF_REFSEG, ((size_t)kf0 + n) * 4) || dalloc(h, BUF_KEYSEG, ((size_t)kf0 + n) * 4)) return -1;
    size_t slotmask_words = 1;
    {
        for (int w = 0; w < n; w++) slotmask_words = std::max(slotmask_words, (size_t)h->desc[w].mwords);
        if (dalloc(h, BUF_MASKQ, (size_t)mask0 * 8) || dalloc(h, BUF_SLOTMASK, (size_t)obs0 * slotmask_words * 8) || dalloc(h, BUF_REFQ, (size_t)pt0 * 4)) return -1;
    }
    if (dalloc(h, BUF_SLOTO, (size_t)obs0 * 4)) return -1;
    if (dalloc(h, BUF_TSLOT, (size_t)obs0 * 4) || dalloc(h, BUF_KFDIR, (size_t)kf0 * 32 * 8)) return -1;
    if (dalloc(h, BUF_SLOTREF, (size_t)obs0 * 4) || dalloc(h, BUF_SLOTQ, (size_t)obs0 * 4) || dalloc(h, BUF_RECQ, (size_t)pt0 * 4) || dalloc(h, BUF_TSQ, (size_t)pt0 * 8 * 4)) return -1;
    if (dalloc(h, BUF_ITEMBEG, ((size_t)pair0 + n) * 4) || dalloc(h, BUF_ITEMMID, ((size_t)pair0 + n) * 4) || dalloc(h, BUF_ITEMS, (size_t)item0 * 8)) return -1;
    if (dalloc(h, BUF_STKEY, (size_t)pt0 * 4) || dalloc(h, BUF_LMORDER, (size_t)pt0 * 4) || dalloc(h, BUF_SLOTOBS, (size_t)obs0 * 4) || dalloc(h, BUF_PTINV, (size_t)pt0 * 4)) return -1;
    if (!inc_on && (h2d(h, BUF_OBSUV, uv) || h2d(h, BUF_OBSW, ow))) return -1;
    const bool idp = probs[0]->variant == VBA_VARIANT_PRV_IDP;
    // (inverse-depth windows evaluate the depth of an edge where they need it, idp_edge_eval: no per-edge copy)
    if (dalloc(h, BUF_LVL, (size_t)obs0) || dalloc(h, BUF_CHI2E, (size_t)obs0 * 8) || dalloc(h, BUF_DEPTH, idp ? 16 : (size_t)obs0 * 8)) return -1;
    if (dalloc(h, BUF_EREC, (size_t)obs0 * (idp ? VBA_EREC1 : VBA_EREC) * 8) || dalloc(h, BUF_PREC, (size_t)pt0 * VBA_PREC * 8)) return -1;
    if (dalloc(h, BUF_SLOT, ((size_t)obs0 + pt0) * (probs[0]->variant == VBA_VARIANT_PRV_IDP ? VBA_SLOT : VBA_SLOT3) * 8)) return -1;
    if (dalloc(h, BUF_CHI2F, (size_t)obs0 * 8)) return -1;
    if (h2d(h, BUF_IMUI, imui) || h2d(h, BUF_IMUJ, imuj) || h2d(h, BUF_IMUMEAS, meas) || h2d(h, BUF_IMUINFO, info)) return -1;
    if (dalloc(h, BUF_IMUH, (size_t)imu0 * VBA_IMUH * 8) || dalloc(h, BUF_IMUCHI, (size_t)imu0 * 4 * 8) || dalloc(h, BUF_IMUJREC, (size_t)imu0 * IMU_JREC * 8)) return -1;
    if (dalloc(h, BUF_S, S_tot * 8) || dalloc(h, BUF_VEC, (size_t)vec0 * 8) || dalloc(h, BUF_BPOSE, (size_t)vec0 * 2 * 8)) return -1;
    if (dalloc(h, BUF_LF, S_tot * 8) || dalloc(h, BUF_YV, (size_t)vec0 * 8)) return -1;
    if (h2d_vec(h, BUF_TLSTEP, tlstep, G.s_int[0]) || h2d_vec(h, BUF_TLPAIR, tlpair, G.s_int[1]) || h2d_vec(h, BUF_TLPANB, tlpanb, G.s_int[2]) ||
        h2d_vec(h, BUF_TLPAN, tlpan, G.s_int[3]) || h2d_vec(h, BUF_TLKB, tlkb, G.s_int[4]) || h2d_vec(h, BUF_TLK, tlk, G.s_int[5]) ||
        h2d_vec(h, BUF_CU, culist, G.s_int[12]) || h2d_vec(h, BUF_CHAINTAB, chaintab, G.s_int[13])) return -1;
    if (dalloc(h, BUF_DVEC, (size_t)vec0 * 8) || dalloc(h, BUF_WINV, (size_t)n * 1024 * 8 * (1 + (size_t)(h->ll_mode ? h->max_nc : 0)))) return -1;
    if (dalloc(h, BUF_VARACT, (size_t)vec0 * 4)) return -1;
    if (h2d(h, BUF_PAIRA, pair_a) || h2d(h, BUF_PAIRB, pair_b)) return -1;
    h->solver = probs[0]->solver;
    if (pcg) {
        adjbeg.resize((size_t)kf0 + n, 0);
        if (h2d_vec(h, BUF_ADJBEG, adjbeg, G.s_int[7]) || h2d_vec(h, BUF_ADJ, adj, G.s_int[8])) return -1;
        if (dalloc(h, BUF_PCGV, (size_t)vec0 * 5 * 8) || dalloc(h, BUF_PCGM, (size_t)kf0 * 450 * 8) || dalloc(h, BUF_PCGS, (size_t)n * 8 * 8)) return -1;
    }
    if (h2d(h, BUF_PIMUBEG, pimu_begin) || h2d(h, BUF_PIMU, pimu) || h2d_vec(h, BUF_LINBLK, linblk, G.s_int[6])) return -1;
    prun0.resize(linblk.size() / 4 + 1, 0); prefbeg.resize((size_t)kf0 + n + 1, 0); preflist.resize((size_t)pt0 + 1, 0);
    if (h2d_vec(h, BUF_PRUN0, prun0, G.s_int[9]) || h2d_vec(h, BUF_PREFBEG, prefbeg, G.s_int[10]) || h2d_vec(h, BUF_PREFLIST, preflist, G.s_int[11])) return -1;
    if (h2d(h, BUF_OFFPAIR, offpair) || h2d(h, BUF_PAIRMASK, pairmask)) return -1;
    if (dalloc(h, BUF_PART, (size_t)part0 * 8) || dalloc(h, BUF_OUTL, (size_t)obs0) || dalloc(h, BUF_OUTCHI, (size_t)obs0 * 8)) return -1;
    h->res_bytes = 0;
    if (n < 4) {   // few windows: everything the download reads lives in ONE block -- one D2H copy behind the run (do_run)
        const size_t sz[7] = {sizeof(WinCtrl) * (size_t)n, pose.size() * 8, vel.size() * 8, bias.size() * 8, pt.size() * 8, (size_t)obs0, (size_t)obs0 * 8};
        const int ids[7] = {BUF_CTRL, BUF_POSE, BUF_VEL, BUF_BIAS, BUF_PT, BUF_OUTL, BUF_OUTCHI};
        size_t off = 0;
        for (int i = 0; i < 7; i++) { h->res_off[i] = off; off += (std::max<size_t>(sz[i], 16) + 255) / 256 * 256; }
        if (dalloc(h, BUF_RESULTS, off)) return -1;
        HIPCHK(h, h->res_host.ensure(off));
        for (int i = 0; i < 7; i++) {
            h->buf[ids[i]].view = reinterpret_cast<char*>(h->buf[BUF_RESULTS].p) + h->res_off[i];
            h->buf[ids[i]].view_bytes = std::max<size_t>(sz[i], 16);
        }
        h->res_bytes = off;
    }
    // S: zero everything once, identity on the pads
    // (PCG reads whole keyframe-pair blocks, also the sub-blocks no factor tile covers and no Schur kernel writes: zero them once)
    if (use_left_looking(h, n) || pcg) HIPCHK(h, hipMemsetAsync(h->buf[BUF_S].p, 0, S_tot * 8, h->up_stream));
    for (int w = 0; w < n && !(use_left_looking(h, n) || pcg); w++) {  // only the pad rows of S must be zero (identity on their diagonal, below)
        const WinDesc& d = h->desc[w];
        if (d.order == 2) {   // pads between the parts: their COLUMNS run through tiles of the factor too -- zero the whole block once
            HIPCHK(h, hipMemsetAsync(dp<double>(h, BUF_S) + d.S0, 0, (size_t)d.nS * d.nS * 8, h->up_stream));
            continue;
        }
        for (int q = 0; q < 3; q++)
            if (d.padn[q] > 0)
                HIPCHK(h, hipMemsetAsync(dp<double>(h, BUF_S) + d.S0 + (size_t)d.pad0[q] * d.nS, 0, (size_t)d.padn[q] * d.nS * 8, h->up_stream));
    }
    HIPCHK(h, hipMemsetAsync(h->buf[BUF_VEC].p, 0, (size_t)vec0 * 8, h->up_stream));
    HIPCHK(h, hipMemsetAsync(h->buf[BUF_YV].p, 0, (size_t)vec0 * 8, h->up_stream));
    HIPCHK(h, hipMemsetAsync(h->buf[BUF_BPOSE].p, 0, (size_t)vec0 * 16, h->up_stream));
    if (h2d_flush(h)) return -1;
    Batch& B = h->B;
    B.desc = dp<WinDesc>(h, BUF_DESC); B.ctrl = dp<WinCtrl>(h, BUF_CTRL); B.n_win = n;
    B.pose = dp<double>(h, BUF_POSE); B.vel = dp<double>(h, BUF_VEL); B.bias = dp<double>(h, BUF_BIAS); B.kfR = dp<double>(h, BUF_KFR);
    B.pose0 = dp<double>(h, BUF_POSE0); B.vel0 = dp<double>(h, BUF_VEL0); B.bias0 = dp<double>(h, BUF_BIAS0);
    B.pose_bk = dp<double>(h, BUF_POSEBK); B.vel_bk = dp<double>(h, BUF_VELBK); B.bias_bk = dp<double>(h, BUF_BIASBK);
    B.pt = dp<double>(h, BUF_PT); B.pt0 = dp<double>(h, BUF_PT0); B.pt_bk = dp<double>(h, BUF_PTBK);
    B.pt_ref = dp<int>(h, BUF_PTREF); B.pt_obs_begin = dp<int>(h, BUF_PTOBS);
    B.obs_kf = dp<int>(h, BUF_OBSKF); B.obs_pt = dp<int>(h, BUF_OBSPT);
    B.obs_uv = dp<double>(h, BUF_OBSUV); B.obs_w = dp<double>(h, BUF_OBSW);
    B.lvl = dp<unsigned char>(h, BUF_LVL); B.chi2_e = dp<double>(h, BUF_CHI2E); B.depth_e = dp<double>(h, BUF_DEPTH);
    B.chi2_f = (probs[0]->variant == VBA_VARIANT_PRV_IDP) ? nullptr : dp<double>(h, BUF_CHI2F);
    B.erec = dp<double>(h, BUF_EREC); B.prec = dp<double>(h, BUF_PREC); B.slot = dp<double>(h, BUF_SLOT); B.kf_fix = dp<unsigned char>(h, BUF_KFFIX);
    B.imu_i = dp<int>(h, BUF_IMUI); B.imu_j = dp<int>(h, BUF_IMUJ);
    B.imu_meas = dp<double>(h, BUF_IMUMEAS); B.imu_info = dp<double>(h, BUF_IMUINFO);
    B.imuH = dp<double>(h, BUF_IMUH); B.imu_chi = dp<double>(h, BUF_IMUCHI); B.imu_jrec = dp<double>(h, BUF_IMUJREC);
    B.S = dp<double>(h, BUF_S); B.vec = dp<double>(h, BUF_VEC); B.bpose = dp<double>(h, BUF_BPOSE);
    B.Lf = dp<double>(h, BUF_LF); B.yv = dp<double>(h, BUF_YV);
    B.l_packed = h->ll_mode ? 1 : 0;
    B.tl_step_begin = dp<int>(h, BUF_TLSTEP); B.tl_pairs = dp<int>(h, BUF_TLPAIR);
    B.tl_pan_begin = dp<int>(h, BUF_TLPANB); B.tl_pan = dp<int>(h, BUF_TLPAN);
    B.tl_kl_begin = dp<int>(h, BUF_TLKB); B.tl_kl = dp<int>(h, BUF_TLK); B.tl_cu = dp<int>(h, BUF_CU); B.tl_ct = dp<int>(h, BUF_CHAINTAB);
    B.dvec = dp<double>(h, BUF_DVEC); B.winv = dp<double>(h, BUF_WINV); B.w_total = n; B.w_stride = h->max_nc;
    B.slot_perm = dp<int>(h, BUF_SLOTPERM); B.pt_perm = dp<int>(h, BUF_PTPERM);
    B.var_act = dp<int>(h, BUF_VARACT);
    B.pair_a = dp<int>(h, BUF_PAIRA); B.pair_b = dp<int>(h, BUF_PAIRB);
    B.item_begin = dp<int>(h, BUF_ITEMBEG); B.items = dp<int>(h, BUF_ITEMS); B.item_mid = dp<int>(h, BUF_ITEMMID);
    B.kf_dir = dp<double>(h, BUF_KFDIR); B.slot_lm = dp<int>(h, BUF_SLOTOBS); B.slot_o = dp<int>(h, BUF_SLOTO); B.rec_lm = dp<int>(h, BUF_PTINV);
    B.adj_begin = dp<int>(h, BUF_ADJBEG); B.adj = dp<int>(h, BUF_ADJ); B.pcg_v = dp<double>(h, BUF_PCGV); B.pcg_m = dp<double>(h, BUF_PCGM); B.pcg_s = dp<double>(h, BUF_PCGS);
    { static const int jac = getenv("VBA_PCG_JACOBI") ? 1 : 0; B.pcg_tri = jac ? 0 : 1; }
    B.lmask = dp<unsigned long long>(h, BUF_LMASK); B.kf_seg = dp<int>(h, BUF_KFSEG); B.ref_seg = dp<int>(h, BUF_REFSEG);
    B.pimu_begin = dp<int>(h, BUF_PIMUBEG); B.pimu = dp<int>(h, BUF_PIMU);
    B.lin_blk = dp<int>(h, BUF_LINBLK);
    B.prun0 = dp<int>(h, BUF_PRUN0); B.pref_begin = dp<int>(h, BUF_PREFBEG); B.pref_list = dp<int>(h, BUF_PREFLIST);
    B.off_pair = dp<int>(h, BUF_OFFPAIR); B.pair_mask = dp<int>(h, BUF_PAIRMASK);
    B.part = dp<double>(h, BUF_PART);
    B.stop_host_word = h->stop_dev;
    B.alive_cnt = h->stop_dev + 64;
    if (dalloc(h, BUF_ALIVE, 14 * 1024 * sizeof(int))) return -1;
    B.alive_dev = dp<int>(h, BUF_ALIVE);
    B.stop_word = (n >= 64) ? B.alive_dev + 1023 : h->stop_dev;   // few windows read the pinned word themselves (no poll launches)
    B.out_outlier = dp<unsigned char>(h, BUF_OUTL); B.out_chi2 = dp<double>(h, BUF_OUTCHI);
    if (dalloc(h, BUF_DBG, 4096)) return -1;
    B.dbg = dp<double>(h, BUF_DBG);
    static_assert(VBA_NB <= 64, "k_init_pads covers the pads with one wave");
    VBA_LAUNCH(k_init_pads, dim3(n), dim3(64), 0, h->up_stream, B);
    {   // the device half of the structure build
        StBuild T;
        T.obs_pt = dp<int>(h, BUF_OBSPT); T.slot_perm = dp<int>(h, BUF_SLOTPERM); T.pt_perm = dp<int>(h, BUF_PTPERM);
        T.kf_seg = dp<int>(h, BUF_KFSEG); T.ref_seg = dp<int>(h, BUF_REFSEG);
        T.item_begin = dp<int>(h, BUF_ITEMBEG); T.item_mid = dp<int>(h, BUF_ITEMMID); T.items = dp<int>(h, BUF_ITEMS);
        T.st_key = dp<int>(h, BUF_STKEY); T.lm_order = dp<int>(h, BUF_LMORDER); T.slot_obs = dp<int>(h, BUF_SLOTOBS); T.pt_inv = dp<int>(h, BUF_PTINV);
        const size_t sh_order = 3 * ((size_t)max_kf + 1) * sizeof(int), sh_row = 2 * (size_t)std::max(1, h->max_free) * sizeof(int);
        if (sh_order > 60000 || sh_row > 60000) return fail(h, "window with too many keyframes for the structure build");
        T.key_seg = dp<int>(h, BUF_KEYSEG); T.tslot = dp<int>(h, BUF_TSLOT);
        T.mask_q = dp<unsigned long long>(h, BUF_MASKQ); T.slot_mask = dp<unsigned long long>(h, BUF_SLOTMASK); T.ref_q = dp<int>(h, BUF_REFQ);
        T.smw = (int)slotmask_words;
        T.row_lds = getenv("VBA_ST_ROW_LDS") ? 1 : 0;   // (read per upload: the test flips it inside one process)
        T.slot_o = dp<int>(h, BUF_SLOTO);
        T.slot_ref = dp<int>(h, BUF_SLOTREF); T.slot_q = dp<int>(h, BUF_SLOTQ); T.rec_q = dp<int>(h, BUF_RECQ); T.tsq = dp<int>(h, BUF_TSQ);
        VBA_LAUNCH(k_st_hist, dim3(n), dim3(n <= 64 ? 1024 : 256), sh_order, h->up_stream, B, T);
        {
            const int max_chunks = std::max(1, h->max_pt_blk);   // 64-landmark blocks of the largest window
            if (dalloc(h, BUF_RECCNT, (size_t)kf0 * max_chunks * 2 * 4)) return -1;
            T.rec_cnt = dp<int>(h, BUF_RECCNT);
            VBA_LAUNCH(k_st_lm_count, dim3(max_chunks, n), dim3(64), 0, h->up_stream, B, T, max_chunks);
            VBA_LAUNCH(k_st_rec_scan, dim3(n), dim3(256), 0, h->up_stream, B, T, max_chunks);
            VBA_LAUNCH(k_st_lm_fill, dim3(max_chunks, n), dim3(64), 0, h->up_stream, B, T, max_chunks);
            VBA_LAUNCH(k_st_rec_count, dim3(max_chunks, n), dim3(64), 0, h->up_stream, B, T, max_chunks);
            VBA_LAUNCH(k_st_rec_scan, dim3(n), dim3(256), 0, h->up_stream, B, T, max_chunks);
            VBA_LAUNCH(k_st_rec_fill, dim3(max_chunks, n), dim3(64), 0, h->up_stream, B, T, max_chunks);
        }
        VBA_LAUNCH(k_st_count, dim3(h->max_free, n), dim3(64), sh_row, h->up_stream, B, T, h->max_free);
        VBA_LAUNCH(k_st_scan, dim3(n), dim3(256), 0, h->up_stream, B, T);
        VBA_LAUNCH(k_st_fill, dim3(h->max_free, n), dim3(64), sh_row, h->up_stream, B, T, h->max_free);
        HIPCHK(h, hipGetLastError());
    }
    const double t_enq = now_ms();
    // vba_solve (one call: upload, run, download) does not come back to the host here: the run stream waits for the upload stream
    // on the device (an event), and the run's kernels queue up behind the structure build instead of behind a host round trip
    h->up_pending = false;
    if (defer_sync && h->up_done) {
        HIPCHK(h, hipEventRecord(h->up_done, h->up_stream));
        h->up_pending = true;
    } else
        HIPCHK(h, hipStreamSynchronize(h->up_stream));
    if (timing) fprintf(stderr, "[vba] chain columns: min %d max %d, update tiles %d, rows %d, ll %d\n", h->min_nc, h->max_nc, h->max_cu, h->max_chain_rows, (int)h->ll_mode);
    if (timing) fprintf(stderr, "[vba] %p t=%.1f upload %d windows: total %.3f ms (structure %.3f, pack %.3f, alloc+H2D enqueue %.3f, sync %.3f)\n", (void*)h, now_ms(), n,
                        now_ms() - t_begin, t_struct, t_pack - t_begin - t_struct, t_enq - t_pack, now_ms() - t_enq);
    h->uploaded = true;
    h->ran = false;
    h->dl_prefetched = false;
    return 0;
}

// The caller's stop flag (g2o's forceStopFlag, sparse_optimizer.h:188) at ITS width: the reference hands over `bool* pbStopFlag` =
// &LocalMapping::mbAbortBA, one byte that the Tracking thread writes (include/Optimizer.h:22-24, src/LocalMapping.cpp:1769-1772);
// a C caller may keep an int.  The byte / word is read, never written.
struct StopRef {
    const volatile void* p = nullptr;
    int width = 0;   // bytes: 1 (vba_*_b) or 4
    bool set() const {
        if (!p) return false;
        return width == 1 ? *reinterpret_cast<const volatile unsigned char*>(p) != 0 : *reinterpret_cast<const volatile int*>(p) != 0;
    }
    explicit operator bool() const { return p != nullptr; }
};
StopRef stop_int(const volatile int* f) { StopRef r; r.p = f; r.width = 4; return r; }
StopRef stop_byte(const volatile unsigned char* f) { StopRef r; r.p = f; r.width = 1; return r; }

// ---- the launch schedule ------------------------------------------------------------------------------
// g2o polls forceStopFlag before every iteration (sparse_optimizer.cpp:376).  The device reads a pinned word; whoever enqueues or
// waits on the host copies the caller's flag into it -- at every iteration it enqueues and while it waits for the device.
inline void forward_stop(Handle* h, StopRef stop_flag) {
    if (stop_flag.set()) *h->stop_host = 1;
}
hipError_t wait_event_forwarding(Handle* h, hipEvent_t ev, StopRef stop_flag) {
    if (!stop_flag) return hipEventSynchronize(ev);
    for (;;) {
        const hipError_t e = hipEventQuery(ev);
        if (e != hipErrorNotReady) return e;
        forward_stop(h, stop_flag);
        std::this_thread::yield();
    }
}
void enqueue_solve_iteration(Handle* h, StopRef stop_flag = StopRef()) {
    const Batch& B = h->B;
    const int n = h->n_win;         // windows of this group: grid sizes
    const int rn = h->regime_n;     // windows of the batch: kernel choice
    const bool idp = h->variant == VBA_VARIANT_PRV_IDP;
    const int ngrp = (n >= 8) ? 8 * ((n + 7) / 8) : n;   // grids whose workgroups schur_map() deals to the XCDs by window
    {
        ProfScope ps(h, VBA_PROF_SCHUR);
        if (idp) {
            static const int fused_schur = getenv("VBA_SCHUR_SPLIT") ? 0 : 1;
            if (rn >= 8 && fused_schur) {
                VBA_LAUNCH(k_schur_all, dim3((h->max_free + h->max_quads) * ngrp), dim3(64), 0, h->stream, B, h->max_free, h->max_quads);
            } else if (fused_schur) {
                VBA_LAUNCH(k_schur_all_w, dim3((h->max_free + h->max_offp) * ngrp), dim3(64), 0, h->stream, B, h->max_free, h->max_offp);
            } else {
            VBA_LAUNCH(k_schur_diag, dim3(h->max_free * ngrp), dim3(64), 0, h->stream, B, h->max_free);
            if (rn >= 8) VBA_LAUNCH(k_schur_off, dim3(h->max_quads * ngrp), dim3(64), 0, h->stream, B, h->max_quads);
            else VBA_LAUNCH(k_schur_off_w, dim3(h->max_offp * ngrp), dim3(64), 0, h->stream, B, h->max_offp);
            }
        } else {
            VBA_LAUNCH(k_dinv, dim3(h->max_pt_blk, n), dim3(64), 0, h->stream, B);
            // (diagonal and off-diagonal pairs in two launches: fusing them as for the inverse-depth records gained nothing at C2)
            VBA_LAUNCH(k_schur_diag3, dim3(h->max_free * ngrp), dim3(64), 0, h->stream, B, h->max_free, 0);
            if (rn >= 8) VBA_LAUNCH(k_schur_off3, dim3(h->max_quads * ngrp), dim3(64), 0, h->stream, B, h->max_quads);
            else VBA_LAUNCH(k_schur_off3_w, dim3(h->max_offp * ngrp), dim3(64), 0, h->stream, B, h->max_offp);
        }
    }
    if (h->solver == VBA_SOLVER_PCG) {
        // Two launches per CG iteration for all windows of the group; the host enqueues BATCHES of iterations and reads one pinned
        // word per batch (did any window go on?) two batches behind the device.  Converged windows exit at the first instruction.
        ProfScope ps(h, VBA_PROF_FACTOR);
        const size_t pcg_shm = (size_t)h->max_free * 16 * sizeof(double);   // the sweeps of the tridiagonal preconditioner
        VBA_LAUNCH(k_pcg_init, dim3(n), dim3(256), pcg_shm, h->stream, B);
        const int per_batch = 32, RING = 16, row_blocks = (h->max_nS + PCG_ROWS - 1) / PCG_ROWS;
        volatile int* ring = h->stop_host + 1024 + 16 * h->cur_group;
        int* ring_dev = h->stop_dev + 1024 + 16 * h->cur_group;
        std::vector<hipEvent_t> ev;
        const int max_batches = (20 * h->max_nS + 50) / per_batch + 2;
        for (int b = 0; b < max_batches; b++) {
            if (b >= 2) {
                (void)wait_event_forwarding(h, ev[b - 2], stop_flag);
                if (ring[(b - 2) % RING] == 0) break;   // every window had converged (or broken down) by the end of batch b-2
            }
            forward_stop(h, stop_flag);
            ring[b % RING] = 0;
            for (int it = 0; it < per_batch; it++) {
                VBA_LAUNCH(k_pcg_matvec, dim3(row_blocks, n), dim3(256), 0, h->stream, B);
                VBA_LAUNCH(k_pcg_step, dim3(n), dim3(256), pcg_shm, h->stream, B, ring_dev + (b % RING));
            }
            ev.push_back(get_evt(h));
            (void)hipEventRecord(ev.back(), h->stream);
        }
        VBA_LAUNCH(k_pcg_finish, dim3(n), dim3(256), 0, h->stream, B);
    } else {
    {
        ProfScope ps(h, VBA_PROF_FACTOR);
        // Two regimes (decided for the whole batch at upload, also for its window groups): from VBA_LL_MIN = 256 windows on the
        // left-looking tile kernels (S stays pristine, the factor is tile-packed), below that one fused right-looking launch per block
        // column.  (Until round 3 there was a third one in between, 64..255 windows: the panel solves and the MFMA updates of a
        // column in two launches, because the fused kernel redid the diagonal tile and two panel solves in every tile-pair workgroup.
        // With the DPP elimination of k_chol_step4 (then: its first form, k_chol_step3) the fused launch wins up to the left-looking threshold -- 64 windows 9.2 ms per step
        // against 12.7, 128: 15.1 / 17.7, 200: 21.8 / 23.1, left-looking at 200: 21.7 -- and the split kernels are gone.)
        // the chain columns [0, nc) of every window in one launch (vba_chain.h); the per-column kernels start behind them
        const int k_first = h->max_nc > 0 ? std::min(h->min_nc, h->max_nc) : 0;
        if (h->max_nc > 0) {
            if (h->ll_mode) {
                VBA_LAUNCH(k_chol_chain_diag, dim3(n), dim3(64), 0, h->stream, B);
                if (h->max_chain_rows > 0) VBA_LAUNCH(k_chol_chain_panel, dim3(h->max_chain_rows * ngrp), dim3(64), 0, h->stream, B, h->max_chain_rows);
            }
            else {
                VBA_LAUNCH(k_chol_chain_rows, dim3(std::max(1, h->max_chain_rows), n), dim3(h->max_split > 0 ? 512 : 256), 0, h->stream, B);   // (two chains: two halves)
                if (h->max_cu > 0) VBA_LAUNCH(k_chol_chain_upd, dim3(h->max_cu, n), dim3(512), 0, h->stream, B);
            }
        }
        if (h->ll_mode) {
            for (int k = k_first; k < h->max_nb; k++) {  // every tile read once, updated in registers, written once
                VBA_LAUNCH(k_chol_diag_ll2, dim3(n), dim3(64), 0, h->stream, B, k);
                if (h->pan_grid[k] > 0) VBA_LAUNCH(k_chol_panel_ll, dim3(h->pan_grid[k] * ngrp), dim3(64), 0, h->stream, B, k, h->pan_grid[k]);
            }
        } else {
            // form 1 (test hook vba_debug_set_chol_step / VBA_CHOL_STEP=1): the first version of the step -- diagonal tile, then the
            // panel solves, v_readlane broadcasts; kept as the cross-check of the hand-written DPP instruction stream
            static const int env_form = getenv("VBA_CHOL_STEP") ? atoi(getenv("VBA_CHOL_STEP")) : 0;
            const int step_form = h->opt_chol_step > 0 ? h->opt_chol_step : (env_form > 0 ? env_form : 4);
            for (int k = k_first; k < h->max_nb; k++) {
                if (step_form == 1) VBA_LAUNCH(k_chol_step, dim3(h->step_grid[k], n), dim3(64), 0, h->stream, B, k);
                else if (h->n_win == 1 && n == 1 && (int)h->one_sb.size() > k + 1) {   // one window: descriptor and step table ride in the kernel arguments
                    const WinDesc& d0 = h->desc[0];
                    StepOne so;
                    so.algo = d0.algo; so.nS = d0.nS; so.nb = d0.nb; so.vec0 = d0.vec0; so.S0 = d0.S0;
                    so.pair_off = d0.tl_pair0 + h->one_sb[k]; so.npair = h->one_sb[k + 1] - h->one_sb[k];
                    VBA_LAUNCH(k_chol_step4<true>, dim3(h->step_grid[k], 1), dim3(128), 0, h->stream, B, k, so);
                } else VBA_LAUNCH(k_chol_step4<false>, dim3(h->step_grid[k], n), dim3(128), 0, h->stream, B, k, StepOne());
            }
        }
    }
    {
        ProfScope ps(h, VBA_PROF_TRSV);
        static const int trsv_old = getenv("VBA_TRSV_OLD") ? 1 : 0;
        if (h->ll_mode || trsv_old) {
            const size_t shm = ((size_t)h->max_nS + 256 + 32 * 33) * sizeof(double);
            VBA_LAUNCH(k_trsv, dim3(n), dim3(256), shm, h->stream, B);
        } else {   // row-major factor: a solving wave + seven waves that work one column ahead
            const size_t shm = ((size_t)h->max_nS + 2 * TRSV_P_DW * 32 + 2 * 32 * 65 + 32) * sizeof(double) + ((size_t)h->max_pan + h->max_nb + 2) * sizeof(int);
            VBA_LAUNCH(k_trsv_p, dim3(n), dim3(512), shm, h->stream, B);
        }
    }
    }
    {
        ProfScope ps(h, VBA_PROF_UPDATE);
        if (idp) VBA_LAUNCH(k_update, dim3(h->max_pt_blk + h->max_kf_blk, n), dim3(64), 0, h->stream, B, h->max_pt_blk);
        else VBA_LAUNCH(k_update_xyz, dim3(h->max_pt_blk + h->max_kf_blk, n), dim3(64), 0, h->stream, B, h->max_pt_blk);
    }
}

void enqueue_lin(Handle* h, int mode) {
    ProfScope ps(h, VBA_PROF_LINEARIZE);
    if (h->variant == VBA_VARIANT_PRV_IDP) {
        const size_t shm = LIN2_LDS;
        static const int fuse_imu = getenv("VBA_LIN_IMU_SPLIT") ? 0 : 1;
        if (fuse_imu && h->max_imu > 0 && h->regime_n < 64) {   // few windows: edges and IMU factors in one launch
            VBA_LAUNCH(k_lin2_imu, dim3(h->max_lin_blk + h->max_imu, h->n_win), dim3(256), shm, h->stream, h->B, h->max_lin_blk, mode);
            return;
        }
        VBA_LAUNCH(k_lin2, dim3(h->max_lin_blk, h->n_win), dim3(256), shm, h->stream, h->B, h->max_lin_blk, mode);
    } else {
        VBA_LAUNCH(k_lin_xyz_e, dim3(h->max_lin_blk, h->n_win), dim3(256), 0, h->stream, h->B, mode);
        if (h->any_lin_fallback) VBA_LAUNCH(k_lin_xyz, dim3(h->max_pt_blk, h->n_win), dim3(64), 0, h->stream, h->B, h->max_pt_blk, mode);
    }
    if (h->max_imu > 0 && h->regime_n < 64) {   // few windows: latency matters, one launch
        VBA_LAUNCH(k_lin_imu_pair, dim3(h->max_imu, h->n_win), dim3(64), 0, h->stream, h->B, mode);
    } else if (h->max_imu > 0) {   // the IMU factors: a lane per keyframe pair for the Lie-group part, then a wave per pair for J^T Omega J
        VBA_LAUNCH(k_lin_imu_res, dim3((h->max_imu + 63) / 64, h->n_win), dim3(64), 0, h->stream, h->B, mode);
        if (mode == LIN_FULL) VBA_LAUNCH(k_lin_imu_hess, dim3(h->max_imu, h->n_win), dim3(64), 0, h->stream, h->B);
    }
}

// One group of windows of a batch with its own stream (the whole batch is the only group unless VBA_STREAMS > 1)
struct Group {
    Batch B;
    int n_win;
    hipStream_t stream;
    volatile int* alive;  // pinned words of this group: [stage * 32 + it]
    bool dead;
};

// Levenberg-Marquardt schedule (levenberg.cpp:61-164) of a batch cut into window groups, device-resident.
// The launch stream of a group is a sequence of SLOT GROUPS [outer, trial]:
//   outer = linearise + computeLambdaInit + the bookkeeping that opens an outer iteration   -- for windows that owe no trial
//   trial = damp, Schur, factor, solve, update, re-evaluate, accept / reject (+ restore)    -- for windows that owe one
// Every kernel is gated per window on WinCtrl (active, lm_need_trial), so each window consumes the slots that apply to it:
// the usual outer iteration takes one [outer, trial]; a window whose step is rejected skips the next group's outer slot (its
// workgroups exit at once) and retries in that group's trial slot -- windows drift apart by whole slots, never inside one, and a
// window that needs no retry never pays for one (a speculative second trial slot per group cost 6 % at C2: ~20 launches whose
// 300 k workgroups only exit).  The host learns through one
// pinned word per slot group whether any window of the group of windows is still going, and stays two slot groups ahead of
// the device (as the Gauss-Newton schedule does): no host round trip per trial, none per outer iteration on the critical path.
int enqueue_schedule_lm(Handle* h, std::vector<Group>& groups, StopRef stop_flag) {
    const int big_blk = std::max(std::max(h->max_kf_blk, h->max_pt_blk), h->max_obs_blk);
    const int kp_blk = std::max(h->max_kf_blk, h->max_pt_blk);
    const Batch B_all = h->B;
    const int n_all = h->n_win;
    hipStream_t main_stream = h->stream;
    auto use = [&](const Group& g) { h->B = g.B; h->n_win = g.n_win; h->stream = g.stream; h->cur_group = (int)(&g - &groups[0]); };
    auto stage_begin = [&](Group& g, int stage) {
        ProfScope ps(h, VBA_PROF_MISC);
        if (h->regime_n >= 64) VBA_LAUNCH(k_poll_stop, dim3(1), dim3(1), 0, g.stream, g.B);
        VBA_LAUNCH(k_stage_clear, dim3(h->max_ns_blk, g.n_win), dim3(64), 0, g.stream, g.B, stage);
        if (stage == 1) VBA_LAUNCH(k_classify, dim3(h->max_obs_blk, g.n_win), dim3(64), 0, g.stream, g.B);
        VBA_LAUNCH(k_stage_mark, dim3(h->max_free + (h->max_imu + 63) / 64, g.n_win), dim3(64), 0, g.stream, g.B, h->max_free);
    };
    auto stage_end = [&](Group& g) {
        if (h->variant == VBA_VARIANT_PRV_IDP) return;
        ProfScope ps(h, VBA_PROF_MISC);
        VBA_LAUNCH(k_depth_xyz, dim3(std::max(h->max_obs_blk, 1), g.n_win), dim3(64), 0, g.stream, g.B);
    };
    auto outer = [&](Group& g) {   // linearise + computeLambdaInit of one outer iteration
        enqueue_lin(h, LIN_FULL);
        const int ngrp = (g.n_win >= 8) ? 8 * ((g.n_win + 7) / 8) : g.n_win;
        {   // H_pp diagonal for computeLambdaInit (the block it writes into S is rewritten by the first trial): a Schur diagonal pass
            ProfScope ps(h, VBA_PROF_SCHUR);
            VBA_LAUNCH(k_schur_diag3, dim3(h->max_free * ngrp), dim3(64), 0, g.stream, g.B, h->max_free, 1);
        }
        ProfScope ps(h, VBA_PROF_CONTROL);
        if (h->regime_n >= 64) VBA_LAUNCH(k_poll_stop, dim3(1), dim3(1), 0, g.stream, g.B);
        VBA_LAUNCH(k_ctrl_lm_outer, dim3(g.n_win), dim3(64), 0, g.stream, g.B);
    };
    auto trial = [&](Group& g, int* alive_dev, int* alive_mirror) {
        {
            ProfScope ps(h, VBA_PROF_MISC);
            VBA_LAUNCH(k_backup, dim3(kp_blk, g.n_win), dim3(64), 0, g.stream, g.B);
        }
        enqueue_solve_iteration(h, stop_flag);
        enqueue_lin(h, LIN_ERR_TRIAL);
        {
            ProfScope ps(h, VBA_PROF_CONTROL);
            if (h->regime_n >= 64) VBA_LAUNCH(k_poll_stop, dim3(1), dim3(1), 0, g.stream, g.B);
            VBA_LAUNCH(k_ctrl_lm_trial, dim3(g.n_win), dim3(64), 0, g.stream, g.B, alive_dev, alive_mirror);
        }
        {
            ProfScope ps(h, VBA_PROF_MISC);   // pop of a rejected step (with k_backup, the push)
            VBA_LAUNCH(k_restore, dim3(kp_blk, g.n_win), dim3(64), 0, g.stream, g.B);
        }
    };
    auto finish = [&](Group& g) {
        ProfScope ps(h, VBA_PROF_MISC);
        if (h->variant != VBA_VARIANT_PRV_IDP)
            VBA_LAUNCH(k_chi2_fresh_xyz, dim3(std::max(h->max_obs_blk, 1), g.n_win), dim3(64), 0, g.stream, g.B);
        VBA_LAUNCH(k_final_edges, dim3(std::max(h->max_obs_blk, 1), g.n_win), dim3(64), 0, g.stream, g.B);
        VBA_LAUNCH(k_final_sum, dim3(g.n_win), dim3(64), 0, g.stream, g.B);
    };
    for (auto& g : groups) {
        use(g);
        ProfScope ps(h, VBA_PROF_MISC);
        VBA_LAUNCH(k_reset, dim3(std::max(1, std::min(32, big_blk / 4)), g.n_win), dim3(256), 0, g.stream, g.B);
    }
    const int RING = 32;   // pinned alive words per window group (its 64-word block: [0, RING) used here)
    int rc = 0;
    for (int stage = 0; stage < 2 && rc == 0; stage++) {
        for (auto& g : groups) { use(g); stage_begin(g, stage); }
        if (h->max_its[stage] > 0) {
            // upper bound of the slot groups a stage can need: every outer iteration may take up to 10 trials
            const int max_groups = std::min(478, 10 * h->max_its[stage] + 2);
            std::vector<std::vector<hipEvent_t>> ev(groups.size());
            for (auto& g : groups) g.dead = false;
            for (int j = 0; j < max_groups; j++) {
                bool any = false;
                for (size_t gi = 0; gi < groups.size(); gi++) {
                    Group& g = groups[gi];
                    if (g.dead) continue;
                    if (j >= 2) {
                        if (wait_event_forwarding(h, ev[gi][j - 2], stop_flag) != hipSuccess) { rc = -1; break; }
                        if (g.alive[(j - 2) % RING] == 0) { g.dead = true; continue; }   // nobody went on after slot group j-2
                    }
                    any = true;
                    use(g);
                    forward_stop(h, stop_flag);
                    g.alive[j % RING] = 0;   // the word's previous user (group j - RING) was consumed long ago
                    int* alive_dev = h->stop_dev + (g.alive - h->stop_host) + (j % RING);
                    outer(g);
                    trial(g, alive_dev, g.B.alive_dev + 64 + stage * 480 + j);   // a mirror word of its own per slot group (never reused inside a run)
                    ev[gi].push_back(get_evt(h));
                    if (hipEventRecord(ev[gi][j], g.stream) != hipSuccess) { rc = -1; break; }
                }
                if (!any || rc) break;
            }
        }
        for (auto& g : groups) { use(g); stage_end(g); }
    }
    for (auto& g : groups) {
        if (rc) break;
        use(g);
        finish(g);
    }
    h->B = B_all; h->n_win = n_all; h->stream = main_stream; h->cur_group = 0;
    return rc;
}

// The two-stage schedule of a batch cut into window groups.  The groups are independent; their launches are enqueued
// INTERLEAVED, iteration by iteration, each on its own stream, so that while one group sits in the latency-bound block
// columns of its factorisation another one streams through its bandwidth-bound linearise / Schur kernels.
int enqueue_schedule(Handle* h, std::vector<Group>& groups, StopRef stop_flag) {
    if (h->algo == VBA_ALGO_LM) return enqueue_schedule_lm(h, groups, stop_flag);
    const int big_blk = std::max(std::max(h->max_kf_blk, h->max_pt_blk), h->max_obs_blk);
    const Batch B_all = h->B;
    const int n_all = h->n_win;
    hipStream_t main_stream = h->stream;
    auto use = [&](const Group& g) { h->B = g.B; h->n_win = g.n_win; h->stream = g.stream; h->cur_group = (int)(&g - &groups[0]); };
    auto restore = [&]() { h->B = B_all; h->n_win = n_all; h->stream = main_stream; h->cur_group = 0; };
    int rc = 0;
    for (auto& g : groups) {
        use(g);
        ProfScope ps(h, VBA_PROF_MISC);
        VBA_LAUNCH(k_reset, dim3(std::max(1, std::min(32, big_blk / 4)), g.n_win), dim3(256), 0, g.stream, g.B);
    }
    for (int stage = 0; stage < 2 && rc == 0; stage++) {
        for (auto& g : groups) {
            use(g);
            ProfScope ps(h, VBA_PROF_MISC);
            if (h->regime_n >= 64) VBA_LAUNCH(k_poll_stop, dim3(1), dim3(1), 0, g.stream, g.B);
        VBA_LAUNCH(k_stage_clear, dim3(h->max_ns_blk, g.n_win), dim3(64), 0, g.stream, g.B, stage);
            if (stage == 1) VBA_LAUNCH(k_classify, dim3(h->max_obs_blk, g.n_win), dim3(64), 0, g.stream, g.B);
            VBA_LAUNCH(k_stage_mark, dim3(h->max_free + (h->max_imu + 63) / 64, g.n_win), dim3(64), 0, g.stream, g.B, h->max_free);
        }
        {
            // The host stays at most two iterations ahead of the device: before enqueuing iteration it of a group it waits
            // for that group's control kernel of iteration it-2 and stops enqueuing for the group once none of its windows is
            // iterating any more (the |dchi2| < 1e-3 stop usually ends stage 2 after 3 of its 10 iterations).  The device
            // never starves: one full iteration is always queued behind the one being waited for.
            const int nit = h->max_its[stage];
            std::vector<std::vector<hipEvent_t>> ev(groups.size(), std::vector<hipEvent_t>(nit, nullptr));
            const bool word_report = h->n_win == 1 && groups.size() == 1 && !h->profile && nit <= 32;   // see k_ctrl_gn
            const bool pace = nit <= 32;  // also when profiling: the launch counts (and hence the per-launch averages) then equal those of a normal run
            // how far ahead: two iterations for batches (the device must never wait for the host); ONE for a handful of windows,
            // where an iteration is a chain of ~30 short launches that the host enqueues three times faster than the device runs
            // them, and every launch enqueued for a window that has already converged (1.7 us each, 30 per iteration) is latency
            static const int env_depth = getenv("VBA_PACE_DEPTH") ? atoi(getenv("VBA_PACE_DEPTH")) : 0;
            const int depth = env_depth > 0 ? env_depth : (h->regime_n < 8 ? 1 : 2);
            for (auto& g : groups) g.dead = false;
            for (int it = 0; it < nit; it++) {
                bool any = false;
                for (size_t gi = 0; gi < groups.size(); gi++) {
                    Group& g = groups[gi];
                    if (g.dead) continue;
                    if (pace && it >= depth) {
                        const int slot = stage * 32 + it - depth;
                        if (word_report) {   // the control kernel writes 1 (stopped) / 2 (goes on) into the pinned word when it is done
                            long spins = 0;
                            while (g.alive[slot] == 0) {
                                forward_stop(h, stop_flag);
                                if ((++spins & 1023) == 0 && hipStreamQuery(g.stream) != hipErrorNotReady) break;   // drained or failed: nothing will write it
                                std::this_thread::yield();
                            }
                            if (g.alive[slot] != 2) { g.dead = true; continue; }
                        } else {
                            (void)wait_event_forwarding(h, ev[gi][it - depth], stop_flag);
                            if (g.alive[slot] == 0) { g.dead = true; continue; }
                        }
                    }
                    any = true;
                    use(g);
                    forward_stop(h, stop_flag);   // InterruptBA raised while the host paces itself: the device sees it at its next poll
                    enqueue_lin(h, LIN_FULL);
                    {
                        ProfScope ps(h, VBA_PROF_CONTROL);
                        if (h->regime_n >= 64) VBA_LAUNCH(k_poll_stop, dim3(1), dim3(1), 0, g.stream, g.B);
                        VBA_LAUNCH(k_ctrl_gn, dim3(g.n_win), dim3(64), 0, g.stream, g.B, 0, word_report ? stage * 32 + it : -1);
                    }
                    if (pace && !word_report) {
                        ev[gi][it] = get_evt(h);
                        (void)hipEventRecord(ev[gi][it], g.stream);
                    }
                    enqueue_solve_iteration(h, stop_flag);
                }
                if (!any) break;
            }
            for (auto& g : groups) {
                use(g);
                enqueue_lin(h, LIN_ERR);
                ProfScope ps(h, VBA_PROF_CONTROL);
                VBA_LAUNCH(k_ctrl_gn, dim3(g.n_win), dim3(64), 0, g.stream, g.B, 1, -1);
            }
        }
        if (h->variant != VBA_VARIANT_PRV_IDP)
            for (auto& g : groups) {
                use(g);
                ProfScope ps(h, VBA_PROF_MISC);
                VBA_LAUNCH(k_depth_xyz, dim3(std::max(h->max_obs_blk, 1), g.n_win), dim3(64), 0, g.stream, g.B);
            }
    }
    for (auto& g : groups) {
        if (rc) break;
        use(g);
        ProfScope ps(h, VBA_PROF_MISC);
        if (h->variant != VBA_VARIANT_PRV_IDP)
            VBA_LAUNCH(k_chi2_fresh_xyz, dim3(std::max(h->max_obs_blk, 1), g.n_win), dim3(64), 0, g.stream, g.B);
        VBA_LAUNCH(k_final_edges, dim3(std::max(h->max_obs_blk, 1), g.n_win), dim3(64), 0, g.stream, g.B);
        VBA_LAUNCH(k_final_sum, dim3(g.n_win), dim3(64), 0, g.stream, g.B);
    }
    restore();
    return rc;
}

int do_run(Handle* h, StopRef stop_flag) {
    static const bool timing = getenv("VBA_TIMING") != nullptr;
    const double t_run0 = timing ? now_ms() : 0.0;
    if (!h->uploaded) return fail(h, "vba_batch_run before vba_batch_upload");
    HIPCHK(h, hipSetDevice(h->device));
    h->B.dbg_stop_after = h->opt_stop_after;
    const Batch B = h->B;
    const int n = h->n_win;
    *h->stop_host = stop_flag.set() ? 1 : 0;
    for (int i = 64; i < 1024; i++) h->stop_host[i] = 0;
    const long long launch0 = h->n_launch;
    h->evts.clear();
    h->evt_used = 0;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    if (h->profile) {
        ev_begin = get_evt(h);
        ev_end = get_evt(h);
        (void)hipEventRecord(ev_begin, h->stream);
    }
    // Large Gauss-Newton batches can be cut into groups of windows, each with its own stream (enqueue_schedule).
    // (Profiling runs and LM, which needs a host decision per trial, use one group.)
    // Measured on MI355X, C3 windows, windows/s with 1 / 2 / 4 / 8 groups: 64 windows 5.1k / 5.6k / 5.8k / 4.1k; 256: 7.5k / 8.0k /
    // 8.5k / 6.1k; 512: 8.9k / 9.2k / 9.9k / 8.6k; 1024: 9.7k / 10.1k / 10.2k / 10.0k; 2048: 10.2k / 10.4k / 10.3k / 10.1k.
    // LM (C2 windows, 1 / 2 / 4 groups): 256 windows 5.9k / 6.2k / 6.5k, 2048: 6.4k / 6.6k / 6.8k.
    static const int env_streams = getenv("VBA_STREAMS") ? atoi(getenv("VBA_STREAMS")) : 0;
    int want = h->opt_streams > 0 ? h->opt_streams : env_streams;
    static const int lane_streams = getenv("VBA_LANE_STREAMS") ? atoi(getenv("VBA_LANE_STREAMS")) : 2;
    if (want <= 0 && h->is_lane) want = lane_streams;   // several lanes share the chip: fewer window groups each
    // default policy (16..48 windows: 2 groups +5..10 %, 4 groups -40 %; from 64 windows on 4 groups -- round 3, 16 distinct ragged
    // windows with 3+1 .. 5+3 iterations: 4096 windows 14.0-14.2 k/s with 2 groups, 14.6-14.8 k with 4; 2048 windows 13.7 k either way)
    if (want <= 0) want = (n >= 64) ? 4 : (n >= 16) ? 2 : 1;
    const int max_streams = std::min(std::min(14, want), (int)h->xstreams.size() + (h->owns_streams ? 11 : 1));
    int ngroups = 1;
    if (!h->profile && max_streams > 1 && n >= 8)
        ngroups = std::max(1, std::min(max_streams, n / 8));   // a group never falls below the 8 windows of the XCD-aware mapping
    while ((int)h->xstreams.size() < ngroups - 1) {
        hipStream_t st;
        HIPCHK(h, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        h->xstreams.push_back(st);
    }
    std::vector<Group> groups(ngroups);
    std::vector<hipEvent_t> done(ngroups);
    for (int g = 0; g < ngroups; g++) {
        const int w0 = (int)((long long)n * g / ngroups), w1 = (int)((long long)n * (g + 1) / ngroups);
        groups[g].B = B;
        groups[g].B.desc = B.desc + w0;
        groups[g].B.ctrl = B.ctrl + w0;
        groups[g].B.n_win = w1 - w0;
        groups[g].B.alive_cnt = h->stop_dev + 64 + 64 * g;
        groups[g].B.alive_dev = dp<int>(h, BUF_ALIVE) + 1024 * g;
        groups[g].B.stop_word = (h->regime_n >= 64) ? groups[g].B.alive_dev + 1023 : h->stop_dev;
        groups[g].n_win = w1 - w0;
        groups[g].stream = (g == 0) ? h->stream : h->xstreams[g - 1];
        groups[g].alive = h->stop_host + 64 + 64 * g;
        groups[g].dead = false;
    }
    if (h->up_pending) {   // (vba_solve: the upload was not waited for on the host)
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->up_done, 0));
        h->up_pending = false;
    }
    HIPCHK(h, hipMemsetAsync(h->buf[BUF_ALIVE].p, 0, 14 * 1024 * sizeof(int), h->stream));   // the mirror words of this run
    // the other streams start after everything already queued on the main stream (upload, previous run)
    if (ngroups > 1) {
        hipEvent_t e0 = get_evt(h);
        HIPCHK(h, hipEventRecord(e0, h->stream));
        for (int g = 1; g < ngroups; g++) HIPCHK(h, hipStreamWaitEvent(groups[g].stream, e0, 0));
    }
    int rc = enqueue_schedule(h, groups, stop_flag);
    for (int g = 0; g < ngroups && rc == 0; g++) {
        done[g] = get_evt(h);
        if (hipEventRecord(done[g], groups[g].stream) != hipSuccess) rc = -1;
    }
    if (rc) return fail(h, h->err.empty() ? "enqueue failed" : h->err);
    if (h->profile) (void)hipEventRecord(ev_end, h->stream);
    HIPCHK(h, hipGetLastError());
    // behind the last kernel, on the main stream: the control blocks and -- for a few windows, where every synchronous copy of the
    // download is a 20-us round trip on a 3-ms solve -- the result arrays, into pinned staging (do_download only scatters them)
    for (int g = 1; g < ngroups; g++) HIPCHK(h, hipStreamWaitEvent(h->stream, done[g], 0));
    h->hctrl.resize(n);
    if (!h->hctrl.ok) return fail(h, "out of pinned host memory (control blocks)");
    h->dl_prefetched = false;
    if (h->res_bytes) {   // few windows: ONE copy brings the control blocks and every result array (do_upload laid them out in one block)
        HIPCHK(h, hipMemcpyAsync(h->res_host.p, h->buf[BUF_RESULTS].p, h->res_bytes, hipMemcpyDeviceToHost, h->stream));
        h->dl_prefetched = true;
    } else
        HIPCHK(h, hipMemcpyAsync(h->hctrl.data(), B.ctrl, sizeof(WinCtrl) * n, hipMemcpyDeviceToHost, h->stream));
    hipEvent_t ev_all = get_evt(h);
    HIPCHK(h, hipEventRecord(ev_all, h->stream));
    // wait, forwarding the caller's stop flag (g2o forceStopFlag) into the device-visible word
    if (stop_flag) {
        while (hipEventQuery(ev_all) == hipErrorNotReady) {
            if (stop_flag.set()) *h->stop_host = 1;
            std::this_thread::yield();
        }
    }
    HIPCHK(h, hipEventSynchronize(ev_all));
    if (h->dl_prefetched) memcpy(h->hctrl.data(), h->res_host.p, sizeof(WinCtrl) * n);
    h->prof.kernel_launches = h->n_launch - launch0;
    if (h->profile) {
        vba_profile& pf = h->prof;
        memset(&pf, 0, sizeof pf);
        pf.kernel_launches = h->n_launch - launch0;
        for (auto& e : h->evts) {
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e.a, e.b);
            pf.ms[e.cls] += ms;
            pf.launches[e.cls] += 1;
        }
        float tot = 0;
        (void)hipEventElapsedTime(&tot, ev_begin, ev_end);
        pf.total_ms = tot;
        // algorithmic bytes: 32 B observation record + 36 B landmark per linearisation pass;
        // reduced system written once and read once per solve
        for (int w = 0; w < n; w++) {
            const WinDesc& d = h->desc[w];
            const WinCtrl& c = h->hctrl[w];
            double passes = 0, solves = 0;
            for (int s = 0; s < 2; s++)
                if (c.its_done[s] > 0) { passes += c.its_done[s] + 1; solves += c.its_done[s]; }
            pf.bytes[VBA_PROF_LINEARIZE] += passes * (32.0 * d.n_obs + 36.0 * d.n_pt + 432.0 * d.n_free);
            // reduced system ("write n_p^2 8 B + read once by solver"): the Schur class writes S once, the factorisation
            // reads S and writes L once, the two triangular solves read L once each (half the square each)
            pf.bytes[VBA_PROF_SCHUR] += solves * ((double)d.np * d.np * 8.0);
            pf.bytes[VBA_PROF_FACTOR] += solves * ((double)d.np * d.np * 8.0 * 2.0);
            pf.bytes[VBA_PROF_TRSV] += solves * ((double)d.np * d.np * 8.0);
            pf.bytes[VBA_PROF_UPDATE] += solves * (36.0 * d.n_pt + 432.0 * d.n_free);   // per point 28 B read + 8 B write, per free KF 432 B
            pf.factor_flops += solves * h->win_tiles[w] * (2.0 * VBA_NB * VBA_NB * VBA_NB);
        }
    }
    h->ran = true;
    if (timing) fprintf(stderr, "[vba] %p t=%.1f run %d windows: %.3f ms\n", (void*)h, now_ms(), n, now_ms() - t_run0);
    return 0;
}

int do_download(Handle* h, int n, vba_problem* const* inout, vba_result* const* out) {
    static const bool timing = getenv("VBA_TIMING") != nullptr;
    const double t_dl0 = timing ? now_ms() : 0.0;
    if (!h->ran) return fail(h, "vba_batch_download before vba_batch_run");
    if (n != h->n_win) return fail(h, "window count mismatch");
    HIPCHK(h, hipSetDevice(h->device));
    const Batch& B = h->B;
    // Many windows: every result array crosses PCIe ONCE into host staging and host threads scatter it to the callers'
    // arrays (per-window copies cost ~12 synchronous hipMemcpy calls per window, 0.25 ms).  Few windows: the run has left them in the staging already (do_run).
    const bool staged = n >= 4 || h->dl_prefetched;
    if (staged && !h->dl_prefetched) {
        bool want_state = false, want_outl = false, want_chi2 = false;
        for (int w = 0; w < n; w++) {
            if (inout && inout[w] && h->hctrl[w].status != VBA_ABORTED_BEFORE) want_state = true;
            if (out && out[w] && out[w]->obs_outlier) want_outl = true;
            if (out && out[w] && out[w]->obs_chi2) want_chi2 = true;
        }
        const WinDesc& dl = h->desc[n - 1];
        const size_t nkf = (size_t)dl.kf0 + dl.n_kf, npt = (size_t)dl.pt0 + dl.n_pt, nobs = (size_t)dl.obs0 + dl.n_obs;
        const bool vi = h->variant != VBA_VARIANT_SE3_XYZ;
        Staging& G = h->stg;
        if (want_state) { G.dl_pose.resize(7 * nkf); G.dl_pt.resize(3 * npt); }
        if (want_state && vi) { G.dl_vel.resize(3 * nkf); G.dl_bias.resize(12 * nkf); }
        if (want_outl) G.dl_outl.resize(nobs);
        if (want_chi2) G.dl_chi2.resize(nobs);
        if (!G.ok()) return fail(h, "out of pinned host memory (download staging)");
        if (want_state) {
            HIPCHK(h, hipMemcpyAsync(G.dl_pose.data(), B.pose, 56 * nkf, hipMemcpyDeviceToHost, h->dl_stream));
            HIPCHK(h, hipMemcpyAsync(G.dl_pt.data(), B.pt, 24 * npt, hipMemcpyDeviceToHost, h->dl_stream));
            if (vi) {
                HIPCHK(h, hipMemcpyAsync(G.dl_vel.data(), B.vel, 24 * nkf, hipMemcpyDeviceToHost, h->dl_stream));
                HIPCHK(h, hipMemcpyAsync(G.dl_bias.data(), B.bias, 96 * nkf, hipMemcpyDeviceToHost, h->dl_stream));
            }
        }
        if (want_outl) HIPCHK(h, hipMemcpyAsync(G.dl_outl.data(), B.out_outlier, nobs, hipMemcpyDeviceToHost, h->dl_stream));
        if (want_chi2) HIPCHK(h, hipMemcpyAsync(G.dl_chi2.data(), B.out_chi2, 8 * nobs, hipMemcpyDeviceToHost, h->dl_stream));
        HIPCHK(h, hipStreamSynchronize(h->dl_stream));
    }
    // where the staged arrays are: the per-array staging of a big batch, or the one block a small one came back in
    const char* rb = reinterpret_cast<const char*>(h->res_host.p);
    const bool one = h->dl_prefetched;
    const double* s_pose = one ? reinterpret_cast<const double*>(rb + h->res_off[1]) : h->stg.dl_pose.data();
    const double* s_vel = one ? reinterpret_cast<const double*>(rb + h->res_off[2]) : h->stg.dl_vel.data();
    const double* s_bias = one ? reinterpret_cast<const double*>(rb + h->res_off[3]) : h->stg.dl_bias.data();
    const double* s_pt = one ? reinterpret_cast<const double*>(rb + h->res_off[4]) : h->stg.dl_pt.data();
    const unsigned char* s_outl = one ? reinterpret_cast<const unsigned char*>(rb + h->res_off[5]) : h->stg.dl_outl.data();
    const double* s_chi2 = one ? reinterpret_cast<const double*>(rb + h->res_off[6]) : h->stg.dl_chi2.data();
    std::atomic<int> next(0), bad(0);
    auto work = [&]() {
        for (int w = next.fetch_add(1); w < n; w = next.fetch_add(1)) {
            const WinDesc& d = h->desc[w];
            const WinCtrl& c = h->hctrl[w];
            vba_problem* P = inout ? inout[w] : nullptr;
            vba_result* R = out ? out[w] : nullptr;
            auto get = [&](void* dst, const void* dev, const void* host, size_t bytes) {
                if (!bytes) return;
                if (staged) memcpy(dst, host, bytes);
                else if (hipMemcpy(dst, dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) bad.store(1);
            };
            if (P && c.status != VBA_ABORTED_BEFORE) {
                get(P->kf_pose, B.pose + 7 * (size_t)d.kf0, s_pose + 7 * (size_t)d.kf0, 56 * (size_t)d.n_free);
                if (d.pdim == 15) {
                    if (P->kf_vel) get(P->kf_vel, B.vel + 3 * (size_t)d.kf0, s_vel + 3 * (size_t)d.kf0, 24 * (size_t)d.n_free);
                    if (P->kf_bias) get(P->kf_bias, B.bias + 12 * (size_t)d.kf0, s_bias + 12 * (size_t)d.kf0, 96 * (size_t)d.n_free);
                }
                get(P->pt, B.pt + 3 * (size_t)d.pt0, s_pt + 3 * (size_t)d.pt0, 24 * (size_t)d.n_pt);
            }
            if (R) {
                R->chi2_vis = c.chi2_vis; R->chi2_prv = c.chi2_prv; R->chi2_bias = c.chi2_bias;
                R->its_done[0] = c.its_done[0]; R->its_done[1] = c.its_done[1];
                R->n_outliers = c.n_outliers; R->status = c.status;
                R->n_trace = c.n_trace;
                for (int i = 0; i < c.n_trace && i < VBA_TRACE_MAX; i++) R->chi2_trace[i] = c.trace[i];
                R->lambda_final = c.lambda;
                R->lin_iterations = c.lin_its;
                if (c.status != VBA_ABORTED_BEFORE && d.n_obs) {
                    if (R->obs_outlier) get(R->obs_outlier, B.out_outlier + d.obs0, s_outl + d.obs0, (size_t)d.n_obs);
                    if (R->obs_chi2) get(R->obs_chi2, B.out_chi2 + d.obs0, s_chi2 + d.obs0, 8 * (size_t)d.n_obs);
                }
            }
        }
    };
    {
        const int nt = staged ? std::max(1, std::min(host_threads(), n / 8)) : 1;
        std::vector<std::thread> pool;
        for (int t = 1; t < nt; t++) pool.emplace_back(work);
        work();
        for (auto& t : pool) t.join();
    }
    if (bad.load()) return fail(h, "hipMemcpy (download) failed");
    if (timing) fprintf(stderr, "[vba] %p t=%.1f download %d windows: %.3f ms\n", (void*)h, now_ms(), n, now_ms() - t_dl0);
    return 0;
}

}  // namespace

extern "C" {

// parent == nullptr: a handle of its own (four streams, created NOW, before anything ran: created after a first solve they do
// not run concurrently with it -- measured: 64 windows in 4 groups 16.4 ms instead of 11.3 ms when a one-window solve came
// first; the runtime binds streams to its hardware queues when they are created).
// parent != nullptr: a lane of vba_batch_solve.  It owns device buffers, pinned staging and control words, but SHARES the
// parent's four streams, one role each: [0] run, [1] run (second window group), [2] upload (H2D + structure build), [3]
// download (D2H).  The runtime multiplexes streams onto four hardware queues; with streams of their own the lanes' uploads
// landed in the queue of another lane's solve and stalled it behind their transfers (head-of-line blocking: 512 windows
// solved in 70 ms instead of 52).  Only one lane solves at a time (run token), so the run streams are never contended.
static int make_handle(int device, Handle* parent, Handle** out) {
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return -2;  // no CPU fallback
    Handle* h = new Handle();
    h->device = device;
    if (hipSetDevice(device) != hipSuccess) { delete h; return -3; }
    if (parent) {
        if (parent->xstreams.size() < 3) { delete h; return -3; }
        h->owns_streams = false;
        h->is_lane = true;
        h->stream = parent->stream;
        h->xstreams.push_back(parent->xstreams[0]);
        h->up_stream = parent->xstreams[1];
        h->dl_stream = parent->xstreams[2];
    } else {
        if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { delete h; return -3; }
        for (int i = 0; i < 3; i++) {
            hipStream_t st;
            if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) break;
            h->xstreams.push_back(st);
        }
        h->up_stream = h->dl_stream = h->stream;
    }
    void* hp = nullptr;
    if (hipHostMalloc(&hp, 8192, hipHostMallocMapped) != hipSuccess) { delete h; return -4; }   // [0,1024) run control words, [1024,2048) PCG rings
    memset(hp, 0, 8192);
    h->stop_host = reinterpret_cast<volatile int*>(hp);
    *h->stop_host = 0;
    void* dpw = nullptr;
    if (hipHostGetDevicePointer(&dpw, hp, 0) != hipSuccess) { delete h; return -5; }
    h->stop_dev = reinterpret_cast<int*>(dpw);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_lin2), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)LIN2_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_lin2_imu), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)LIN2_LDS);
    // the back-substitution keeps x (nS doubles) in LDS: maps of more than ~5 600 pose dofs need more than the default 64 KiB
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_trsv), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_trsv_p), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipEventCreateWithFlags(&h->up_done, hipEventDisableTiming);
    memset(&h->prof, 0, sizeof h->prof);
    *out = h;
    return 0;
}

int vba_create(int device, void** handle) {
    if (!handle) return -1;
    *handle = nullptr;
    Handle* h = nullptr;
    const int rc = make_handle(device, nullptr, &h);
    if (rc == 0) *handle = h;
    return rc;
}

int vba_destroy(void* handle) {
    Handle* h = reinterpret_cast<Handle*>(handle);
    if (!h) return -1;
    for (Handle* l : h->lanes) (void)vba_destroy(l);
    h->lanes.clear();
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    (void)hipStreamSynchronize(h->up_stream);
    (void)hipStreamSynchronize(h->dl_stream);
    h->stg.release();
    h->hctrl.release();
    h->res_host.release();
    h->up_arena.release();
    h->up_arena_host.release();
    for (auto& b : h->buf) b.release();
    h->preint.release();
    h->pose_arena.release();
    h->pose_host_in.release();
    h->pose_host_out.release();
    for (auto e : h->evt_pool) (void)hipEventDestroy(e);
    if (h->up_done) (void)hipEventDestroy(h->up_done);
    if (h->owns_streams)
        for (auto st : h->xstreams) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    if (h->stop_host) (void)hipHostFree((void*)h->stop_host);
    if (h->owns_streams) (void)hipStreamDestroy(h->stream);
    delete h;
    return 0;
}

const char* vba_last_error(void* handle) {
    Handle* h = reinterpret_cast<Handle*>(handle);
    return h ? h->err.c_str() : "null handle";
}

int vba_batch_upload(void* handle, int32_t n, vba_problem* const* problems) {
    Handle* h = reinterpret_cast<Handle*>(handle);
    if (!h) return -1;
    return do_upload(h, n, problems);
}
int vba_batch_run(void* handle, const volatile int* stop_flag) {
    Handle* h = reinterpret_cast<Handle*>(handle);
    if (!h) return -1;
    return do_run(h, stop_int(stop_flag));
}
int vba_batch_run_b(void* handle, const volatile unsigned char* stop_flag) {
    Handle* h = reinterpret_cast<Handle*>(handle);
    if (!h) return -1;
    return do_run(h, stop_byte(stop_flag));
}
int vba_batch_download(void* handle, int32_t n, vba_problem* const* inout, vba_result* const* out) {
    Handle* h = reinterpret_cast<Handle*>(handle);
    if (!h) return -1;
    return do_download(h, n, inout, out);
}

}  // extern "C"
namespace {
int solve_one(void* handle, vba_problem* inout, vba_result* out, StopRef stop_flag) {
    Handle* h = reinterpret_cast<Handle*>(handle);
    if (!h || !inout || !out) return -1;
    if (stop_flag.set()) {  // src/Optimizer.cpp:453-455: return before anything is built
        out->status = VBA_ABORTED_BEFORE;
        out->its_done[0] = out->its_done[1] = 0;
        out->n_outliers = 0; out->n_trace = 0; out->lin_iterations = 0;
        out->chi2_vis = out->chi2_prv = out->chi2_bias = 0;
        return 0;
    }
    vba_problem* ps[1] = {inout};
    vba_result* rs[1] = {out};
    if (do_upload(h, 1, ps, true)) return -1;
    if (do_run(h, stop_flag)) return -1;
    return do_download(h, 1, ps, rs);
}
int batch_solve(void* handle, int32_t n, vba_problem* const* inout, vba_result* const* out, StopRef stop_flag);
}  // namespace
extern "C" {
int vba_solve(void* handle, vba_problem* inout, vba_result* out, const volatile int* stop_flag) { return solve_one(handle, inout, out, stop_int(stop_flag)); }
int vba_solve_b(void* handle, vba_problem* inout, vba_result* out, const volatile unsigned char* stop_flag) { return solve_one(handle, inout, out, stop_byte(stop_flag)); }
int vba_batch_solve(void* handle, int32_t n, vba_problem* const* inout, vba_result* const* out, const volatile int* stop_flag) {
    return batch_solve(handle, n, inout, out, stop_int(stop_flag));
}
int vba_batch_solve_b(void* handle, int32_t n, vba_problem* const* inout, vba_result* const* out, const volatile unsigned char* stop_flag) {
    return batch_solve(handle, n, inout, out, stop_byte(stop_flag));
}
}  // extern "C"

// Fresh windows in, solved windows out: the batch is cut into chunks and several chunks are in flight at once, each on its
// own lane (a sub-handle with its own streams, device buffers and pinned staging), so that the host-side packing, the H2D
// transfer and the structure build of chunk k+1 and the D2H + scatter of chunk k-1 run while chunk k is being solved.
// Windows are independent (one function-local optimiser per call in the reference, src/Optimizer.cpp:130).  A chunk runs the
// kernels a batch of its size runs (the choice depends on the window count: thresholds 8 / 64 / 256), so chunks of the
// default size give bit for bit what one big upload + run + download gives; across a threshold the sums run in another
// fixed order and the results agree to rounding.
namespace {
int batch_solve(void* handle, int32_t n, vba_problem* const* inout, vba_result* const* out, StopRef stop_flag) {
    Handle* h = reinterpret_cast<Handle*>(handle);
    if (!h) return -1;
    if (n <= 0 || !inout) return fail(h, "vba_batch_solve: bad arguments");
    // measured on MI355X, 4096 fresh C3 windows (scripts/e2e_sweep.py, resident 12.1-13.0k windows/s): chunk x lanes 512x4 7.8k windows/s,
    // 768x3 8.1k, 1024x2 9.05k, 1024x3 8.97k, 1365x2 9.3-9.4k, 1536x2 9.4k, 1700x2 9.6k, 2048x2 (no ramp) 7.7k -- one lane solves while
    // the other packs / transfers / builds its structure / scatters; every chunk pays the fixed cost of its ~1700 launches again
    static const int env_lanes = getenv("VBA_LANES") ? atoi(getenv("VBA_LANES")) : 2;
    static const int env_chunk = getenv("VBA_CHUNK") ? atoi(getenv("VBA_CHUNK")) : 1536;
    const int chunk_max = std::max(1, h->opt_chunk > 0 ? h->opt_chunk : env_chunk);
    // chunk boundaries: a ramp at the start, then equal chunks (no tiny tail).  Uploads go one at a time in chunk order (below) at
    // ~57 us per window, a chunk of s windows solves in ~14 + 0.075 s ms: chunk k+1 is on the device before chunk k's solve ends when
    // the uploads of chunks 2..k+1 fit into the solves of chunks 1..k -- sizes c, 2c, 3.25c, 4.5c with c a quarter of VBA_CHUNK (384,
    // 768, 1248, 1696 for 4096 windows: measured timeline in DESIGN.md section 6).  No chunk falls below 256 windows when the batch has
    // that many: the kernel choice of a chunk (section "regime") then equals the batch's.
    std::vector<int> cbeg(1, 0);
    {
        static const int ramp = getenv("VBA_NO_RAMP") ? 0 : 1;
        int left = n;
        if (const char* e = getenv("VBA_CHUNKS")) {   // experiment: explicit chunk sizes "384,1024,1664" (the rest goes into one last chunk)
            for (const char* q = e; *q && left > 0;) {
                const int c = std::min(left, std::max(1, atoi(q)));
                cbeg.push_back(cbeg.back() + c);
                left -= c;
                while (*q && *q != ',') q++;
                if (*q == ',') q++;
            }
            if (left > 0) cbeg.push_back(cbeg.back() + left);
            left = 0;
        }
        const int c = std::max(256, chunk_max / 4);
        const int steps[4] = {c, 2 * c, 13 * c / 4, 9 * c / 2};
        int cap = chunk_max;
        if (ramp && chunk_max >= 1024) {
            cap = steps[3];
            for (int i = 0; i < 4 && left >= steps[i] + 256; i++) {
                cbeg.push_back(cbeg.back() + steps[i]);
                left -= steps[i];
            }
        }
        const int rest = (left > 0) ? std::max(1, (left + cap - 1) / cap) : 0;
        const int base = cbeg.back();
        for (int q = 1; q <= rest; q++) cbeg.push_back(base + (int)((long long)left * q / rest));
    }
    const int n_lanes = std::max(1, std::min(h->opt_lanes > 0 ? h->opt_lanes : env_lanes, (int)cbeg.size() - 1));
    if (cbeg.size() == 2) {
        if (do_upload(h, n, inout) || do_run(h, stop_flag)) return -1;
        return do_download(h, n, inout, out);
    }
    while ((int)h->lanes.size() < n_lanes) {
        Handle* l = nullptr;
        if (make_handle(h->device, h, &l) != 0) return fail(h, "vba_batch_solve: could not create a lane");
        l->opt_ll_min = h->opt_ll_min;
        l->opt_no_chain = h->opt_no_chain;
        l->opt_stop_after = h->opt_stop_after;
        h->lanes.push_back(l);
    }
    const int n_chunks2 = (int)cbeg.size() - 1;
    std::atomic<int> next(0), bad(0);
    // Lanes that start together stay in step (all pack, then all solve, then all scatter: the GPU idles while the hosts pack).
    // A run token breaks the symmetry: only `run_slots` lanes may be inside the solve at a time, the others pack / transfer /
    // build the structure of their next chunk or scatter their last one meanwhile.
    static const int env_slots = getenv("VBA_RUN_SLOTS") ? atoi(getenv("VBA_RUN_SLOTS")) : 1;
    int run_free = std::max(1, std::min(env_slots, n_lanes));
    std::mutex run_mu;
    std::condition_variable run_cv;
    auto run_gated = [&](Handle* lane) -> int {
        {
            std::unique_lock<std::mutex> lk(run_mu);
            run_cv.wait(lk, [&] { return run_free > 0; });
            run_free--;
        }
        const int rc = do_run(lane, stop_flag);
        {
            std::lock_guard<std::mutex> lk(run_mu);
            run_free++;
        }
        run_cv.notify_one();
        return rc;
    };
    static const bool timing = getenv("VBA_TIMING") != nullptr;
    const double t_call = now_ms();
    int up_turn = 0;
    std::mutex up_mu;
    std::condition_variable up_cv;
    auto work = [&](Handle* lane) {
        for (int c = next.fetch_add(1); c < n_chunks2 && !bad.load(); c = next.fetch_add(1)) {
            const int w0 = cbeg[c], cn = cbeg[c + 1] - w0;
            if (cn <= 0) {
                { std::unique_lock<std::mutex> lk(up_mu); up_cv.wait(lk, [&] { return up_turn == c || bad.load(); }); up_turn = c + 1; }
                up_cv.notify_all();
                continue;
            }
            {   // uploads go one at a time, in chunk order: the first chunk gets every host thread and the whole link (lanes that
                // start together share them and the device waits for the slower of two half-speed uploads), and a third lane
                // can have chunk c+1 on the device before chunk c's solve ends
                std::unique_lock<std::mutex> lk(up_mu);
                up_cv.wait(lk, [&] { return up_turn == c || bad.load(); });
            }
            const double t0 = now_ms();
            int rc = bad.load() ? -1 : do_upload(lane, cn, inout + w0);
            {
                std::lock_guard<std::mutex> lk(up_mu);
                up_turn = c + 1;
            }
            up_cv.notify_all();
            const double t1 = now_ms();
            if (!rc) rc = run_gated(lane);
            const double t2 = now_ms();
            if (!rc) rc = do_download(lane, cn, inout + w0, out ? out + w0 : nullptr);
            if (timing) fprintf(stderr, "[vba_batch_solve] chunk %d (%d windows): upload %.1f..%.1f  run ..%.1f  download ..%.1f ms\n", c, cn, t0 - t_call, t1 - t_call, t2 - t_call, now_ms() - t_call);
            if (rc) {
                {   // the message and `bad` change together, under the mutex the waiting lanes evaluate their predicate under: the first
                    // failing lane writes the message (two lanes failing together cannot both), and a lane that has just found
                    // `up_turn == c || bad` false cannot miss this wake-up
                    std::lock_guard<std::mutex> lk(up_mu);
                    if (!bad.load()) h->err = "vba_batch_solve, windows " + std::to_string(w0) + ".." + std::to_string(w0 + cn - 1) + ": " + lane->err;
                    bad.store(1);
                }
                up_cv.notify_all();   // lanes waiting for their upload turn see `bad`
                return;
            }
        }
    };
    std::vector<std::thread> pool;
    for (int l = 1; l < n_lanes; l++) pool.emplace_back(work, h->lanes[l]);
    work(h->lanes[0]);
    for (auto& t : pool) t.join();
    return bad.load() ? -1 : 0;
}
}  // namespace

extern "C" {

// ---- test / diagnostic hooks: NOT part of include/vislam_ba.h and not in the shipped library.  `make` builds a second flavour,
// libvislam_ba_hooks.so (-DVBA_TEST_HOOKS), that the tests load when they need to look inside (tests/test_abi_exports.py checks
// that libvislam_ba.so exports exactly the header).
#ifdef VBA_TEST_HOOKS
// test/debug hook (not part of include/vislam_ba.h): raw copy out of one device buffer of the last batch
int vba_debug_copy(void* handle, int32_t buf_id, uint64_t offset_bytes, void* dst, uint64_t nbytes) {
    Handle* h = reinterpret_cast<Handle*>(handle);
    if (!h || buf_id < 0 || buf_id >= BUF_N) return -1;
    const DevBuf& b = h->buf[buf_id];
    if (offset_bytes + nbytes > (b.view ? b.view_bytes : b.cap)) return -1;
    (void)hipSetDevice(h->device);
    return hipMemcpy(dst, reinterpret_cast<char*>(b.ptr()) + offset_bytes, nbytes, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
// diagnostic (bench.py --workload c3s): tile products of window w's symbolic factorisation under both elimination orders and the
// order chosen: out[5] = {V/Bias-first, keyframe by keyframe, chosen order, products of the chosen lists, two-sided V/Bias-first}
int vba_debug_tile_products(void* handle, int32_t w, int64_t* out) {
    Handle* h = reinterpret_cast<Handle*>(handle);
    if (!h || !out || w < 0 || w >= h->n_win || (size_t)w >= h->win_tiles.size()) return -1;
    out[0] = h->win_prod_order[3 * (size_t)w]; out[1] = h->win_prod_order[3 * (size_t)w + 1];
    out[2] = h->desc[w].order; out[3] = h->win_tiles[w]; out[4] = h->win_prod_order[3 * (size_t)w + 2];
    return 0;
}
int vba_debug_set_streams(void* handle, int32_t n) {
    Handle* h = reinterpret_cast<Handle*>(handle);
    if (!h) return -1;
    h->opt_streams = n;
    return 0;
}
// every window reads the stop flag as 1 from its n-th terminate() poll on (n < 0: off); the oracle's vba_oracle_solve_ex counts alike
int vba_debug_set_stop_after(void* handle, int32_t n) {
    Handle* h = reinterpret_cast<Handle*>(handle);
    if (!h) return -1;
    h->opt_stop_after = n;
    for (Handle* l : h->lanes) l->opt_stop_after = n;
    return 0;
}
// 1: the first form of the fused factorisation step (v_readlane broadcasts, panel solves after the diagonal tile); anything else: k_chol_step4
int vba_debug_set_chol_step(void* handle, int32_t form) {
    Handle* h = reinterpret_cast<Handle*>(handle);
    if (!h) return -1;
    h->opt_chol_step = form;
    return 0;
}
int vba_debug_set_lin_fallback(void* handle, int32_t on) {
    Handle* h = reinterpret_cast<Handle*>(handle);
    if (!h) return -1;
    h->opt_lin_fallback = on;
    return 0;
}
int vba_debug_set_ll_min(void* handle, int32_t n) {
    Handle* h = reinterpret_cast<Handle*>(handle);
    if (!h) return -1;
    h->opt_ll_min = n;
    for (Handle* l : h->lanes) l->opt_ll_min = n;
    return 0;
}
int vba_debug_set_chunking(void* handle, int32_t chunk, int32_t lanes) {
    Handle* h = reinterpret_cast<Handle*>(handle);
    if (!h) return -1;
    h->opt_chunk = chunk;
    h->opt_lanes = lanes;
    return 0;
}
int vba_debug_buf_id(const char* name) {
    static const char* names[] = {"DESC", "CTRL", "POSE", "VEL", "BIAS", "KFR", "POSE0", "VEL0", "BIAS0", "POSEBK", "VELBK", "BIASBK", "PT", "PT0",
        "PTBK", "PTREF", "PTOBS", "OBSKF", "OBSPT", "OBSUV", "OBSW", "LVL", "CHI2E", "CHI2F", "DEPTH", "EREC", "PREC", "SLOT", "IMUI", "IMUJ",
        "IMUMEAS", "IMUINFO", "IMUH", "IMUCHI", "S", "LF", "YV", "TLSTEP", "TLPAIR", "TLPANB", "TLPAN", "VEC", "BPOSE", "VARACT", "PAIRA",
        "PAIRB", "ITEMBEG", "ITEMS", "PIMUBEG", "PIMU", "PART", "OUTL", "OUTCHI", "LINBLK", "OFFPAIR", "PAIRMASK", "DBG", "CU", "KFFIX", "TLKB", "TLK", "DVEC", "WINV", "SLOTPERM", "PTPERM",
        "LMASK", "KFSEG", "REFSEG", "ITEMMID", "STKEY", "LMORDER", "SLOTOBS", "PTINV", "KEYSEG", "TSLOT", "ADJBEG", "ADJ", "PCGV", "PCGM", "KFDIR", "MASKQ", "SLOTMASK", "REFQ", "PCGS", "IMUJREC", "ALIVE", "SLOTO", "SLOTREF", "SLOTQ", "RECQ", "TSQ", "RECCNT", "RESULTS", "PRUN0", "PREFBEG", "PREFLIST", "CHAINTAB"};
    static_assert(sizeof(names) / sizeof(names[0]) == BUF_N, "buffer name table out of date");
    for (int i = 0; i < BUF_N; i++)
        if (!strcmp(names[i], name)) return i;
    return -1;
}

// chain columns of the factorisation (vba_chain.h): 0 = one launch per block column everywhere, 1 = the default policy
int vba_debug_set_chain(void* handle, int32_t on) {
    Handle* h = reinterpret_cast<Handle*>(handle);
    if (!h) return -1;
    h->opt_no_chain = on ? 0 : 1;
    for (Handle* l : h->lanes) l->opt_no_chain = h->opt_no_chain;
    return 0;
}
#endif  // VBA_TEST_HOOKS

// the size of the host thread pool of a handle in this process (this rank's share of the cores: host_threads above)
int vba_host_threads(void) { return host_threads(); }

int vba_preintegrate(void* handle, int32_t n_edges, const int32_t* sample_begin, const double* gyr, const double* acc,
                     const double* dt, double gyr_meas_cov, double acc_meas_cov, double* imu_meas, double* cov_pvphi,
                     double* imu_info_prv) {
    Handle* h = reinterpret_cast<Handle*>(handle);
    if (!h) return -1;
    if (n_edges <= 0 || !sample_begin || !gyr || !acc || !dt || !imu_meas || !cov_pvphi) return fail(h, "vba_preintegrate: bad arguments");
    HIPCHK(h, hipSetDevice(h->device));
    const int ns = sample_begin[n_edges];
    for (int e = 0; e < n_edges; e++)
        if (sample_begin[e] > sample_begin[e + 1] || sample_begin[e] < 0) return fail(h, "vba_preintegrate: sample_begin is not a CSR");
    // a small private arena: inputs | outputs
    const size_t b_sb = ((size_t)(n_edges + 1) * 4 + 255) / 256 * 256, b_v = ((size_t)ns * 24 + 255) / 256 * 256, b_d = ((size_t)ns * 8 + 255) / 256 * 256;
    const size_t b_m = (size_t)n_edges * 61 * 8, b_c = (size_t)n_edges * 81 * 8;
    const size_t total = b_sb + 2 * b_v + b_d + b_m + 2 * b_c + 1024;
    HIPCHK(h, h->preint.ensure(total));
    char* base = reinterpret_cast<char*>(h->preint.p);
    int* d_sb = reinterpret_cast<int*>(base);
    double* d_g = reinterpret_cast<double*>(base + b_sb);
    double* d_a = reinterpret_cast<double*>(base + b_sb + b_v);
    double* d_dt = reinterpret_cast<double*>(base + b_sb + 2 * b_v);
    double* d_m = reinterpret_cast<double*>(base + b_sb + 2 * b_v + b_d);
    double* d_c = d_m + (size_t)n_edges * 61;
    double* d_i = d_c + (size_t)n_edges * 81;
    HIPCHK(h, hipMemcpyAsync(d_sb, sample_begin, (size_t)(n_edges + 1) * 4, hipMemcpyHostToDevice, h->stream));
    if (ns > 0) {
        HIPCHK(h, hipMemcpyAsync(d_g, gyr, (size_t)ns * 24, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(d_a, acc, (size_t)ns * 24, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(d_dt, dt, (size_t)ns * 8, hipMemcpyHostToDevice, h->stream));
    }
    VBA_LAUNCH(k_preint, dim3(n_edges), dim3(128), 0, h->stream, n_edges, d_sb, d_g, d_a, d_dt, gyr_meas_cov, acc_meas_cov,
                       d_m, d_c, imu_info_prv ? d_i : nullptr);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(imu_meas, d_m, b_m, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(cov_pvphi, d_c, b_c, hipMemcpyDeviceToHost, h->stream));
    if (imu_info_prv) HIPCHK(h, hipMemcpyAsync(imu_info_prv, d_i, b_c, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

namespace {
// Matrix::inverse() of the small dense matrices of the set-up code (Gauss-Jordan, partial pivoting)
bool inverse_host(int n, const double* A, double* Ai) {
    double M[15][30];
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) { M[i][j] = A[i * n + j]; M[i][n + j] = (i == j) ? 1.0 : 0.0; }
    for (int c = 0; c < n; c++) {
        int p = c;
        for (int r = c + 1; r < n; r++)
            if (std::fabs(M[r][c]) > std::fabs(M[p][c])) p = r;
        if (p != c)
            for (int j = 0; j < 2 * n; j++) std::swap(M[c][j], M[p][j]);
        if (!(std::fabs(M[c][c]) > 0.0) || !std::isfinite(M[c][c])) return false;   // singular or non-finite: no information matrix
        const double inv = 1.0 / M[c][c];
        for (int j = 0; j < 2 * n; j++) M[c][j] *= inv;
        for (int r = 0; r < n; r++) {
            if (r == c) continue;
            const double f = M[r][c];
            if (f == 0.0) continue;
            for (int j = 0; j < 2 * n; j++) M[r][j] -= f * M[c][j];
        }
    }
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            Ai[i * n + j] = M[i][n + j];
            if (!std::isfinite(Ai[i * n + j])) return false;
        }
    return true;
}
}  // namespace

int vba_pose_optimize(void* handle, int32_t n_frames, vba_frame_problem* const* inout, vba_frame_result* const* out) {
    Handle* h = reinterpret_cast<Handle*>(handle);
    if (!h) return -1;
    if (n_frames <= 0 || !inout || !out) return fail(h, "vba_pose_optimize: bad arguments");
    HIPCHK(h, hipSetDevice(h->device));
    size_t n_tot = 0;
    for (int f = 0; f < n_frames; f++) {
        const vba_frame_problem* F = inout[f];
        if (!F || !out[f] || F->n_obs < 0 || (F->n_obs > 0 && (!F->obs_pw || !F->obs_uv || !F->obs_w || !out[f]->outlier)))
            return fail(h, "vba_pose_optimize: bad frame");
        if (F->last_is_frame < 0 || F->last_is_frame > 2) return fail(h, "vba_pose_optimize: unknown frame kind");
        if (F->last_is_frame == VBA_FRAME_FRAME && F->n_obs_last < 0) return fail(h, "vba_pose_optimize: negative n_obs_last");
        if (F->last_is_frame == VBA_FRAME_FRAME && F->n_obs_last > 0 && (!F->last_pw || !F->last_uv || !F->last_w)) return fail(h, "vba_pose_optimize: bad last frame");
        n_tot += (size_t)F->n_obs + (F->last_is_frame == VBA_FRAME_FRAME ? (size_t)F->n_obs_last : 0);
    }
    // one device arena and one pinned staging block with the same layout: [desc | pw | uv | w] go up in one copy,
    // [out | lvl] come back in one
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t b_desc = up(sizeof(FrameDesc) * n_frames), b_pw = up((3 * n_tot + 3) * 8), b_uv = up((2 * n_tot + 2) * 8), b_w = up((n_tot + 1) * 8);
    const size_t b_in = b_desc + b_pw + b_uv + b_w;
    const size_t b_out = up(sizeof(FrameOut) * n_frames), b_lvl = up(n_tot + 1), b_err = up((2 * n_tot + 2) * 8);
    HIPCHK(h, h->pose_arena.ensure(b_in + b_out + b_lvl + b_err));
    HIPCHK(h, h->pose_host_in.ensure(b_in));
    HIPCHK(h, h->pose_host_out.ensure(b_out + b_lvl));
    char* hin = reinterpret_cast<char*>(h->pose_host_in.p);
    FrameDesc* desc = reinterpret_cast<FrameDesc*>(hin);
    double* pw = reinterpret_cast<double*>(hin + b_desc);
    double* uv = reinterpret_cast<double*>(hin + b_desc + b_pw);
    double* ww = reinterpret_cast<double*>(hin + b_desc + b_pw + b_uv);
    {   // offsets first, then the frames are packed by a few host threads
        size_t o = 0;
        for (int f = 0; f < n_frames; f++) {
            const vba_frame_problem* F = inout[f];
            FrameDesc& d = desc[f];
            std::memset(&d, 0, sizeof d);
            d.last_is_frame = F->last_is_frame;
            d.n_obs = F->n_obs;
            d.n_last = (d.last_is_frame == VBA_FRAME_FRAME) ? F->n_obs_last : 0;
            d.obs0 = (int)o; o += d.n_obs;
            d.last0 = (int)o; o += d.n_last;
        }
    }
    std::atomic<int> bad_cov(0);
    auto pack = [&](int f) {
        const vba_frame_problem* F = inout[f];
        FrameDesc& d = desc[f];
        d.compute_marg = F->compute_marg ? 1 : 0;
        size_t o = (size_t)d.obs0;
        std::memcpy(&pw[3 * o], F->obs_pw, 24 * (size_t)d.n_obs);
        std::memcpy(&uv[2 * o], F->obs_uv, 16 * (size_t)d.n_obs);
        std::memcpy(&ww[o], F->obs_w, 8 * (size_t)d.n_obs);
        o = (size_t)d.last0;
        if (d.n_last) {
            std::memcpy(&pw[3 * o], F->last_pw, 24 * (size_t)d.n_last);
            std::memcpy(&uv[2 * o], F->last_uv, 16 * (size_t)d.n_last);
            std::memcpy(&ww[o], F->last_w, 8 * (size_t)d.n_last);
        }
        std::memcpy(d.nav, F->nav, sizeof d.nav);
        std::memcpy(d.nav_last, F->nav_last, sizeof d.nav_last);
        std::memcpy(d.prior_nav, F->prior_nav, sizeof d.prior_nav);
        std::memcpy(d.prior_info, F->prior_info, sizeof d.prior_info);
        std::memcpy(d.K, F->K, sizeof d.K);
        quat_to_R_host(F->T_cb + 3, d.Rcb);
        for (int i = 0; i < 3; i++) { d.tcb[i] = F->T_cb[i]; d.g[i] = F->g_w[i]; }
        std::memcpy(d.meas, F->imu_meas, sizeof d.meas);
        if (F->last_is_frame != VBA_FRAME_VISION && !inverse_host(9, F->imu_cov_pvphi, d.info_pvr))   // Matrix9d InvCovPVR = imupreint.getCovPVPhi().inverse(), :2103
            bad_cov.store(1);
        d.inv_bg = F->inv_bg_rw2; d.inv_ba = F->inv_ba_rw2;
        d.hub_prior = (double)(float)std::sqrt(30.5779); d.hub_pvr = (double)(float)std::sqrt(21.666);
        d.hub_bias = (double)(float)std::sqrt(16.812); d.hub_mono = (double)(float)std::sqrt(5.991);
    };
    {
        const int nt = (n_frames >= 256) ? std::max(1, std::min(8, host_threads())) : 1;
        std::atomic<int> next(0);
        auto work = [&]() { for (int f = next.fetch_add(1); f < n_frames; f = next.fetch_add(1)) pack(f); };
        std::vector<std::thread> pool;
        for (int t = 1; t < nt; t++) pool.emplace_back(work);
        work();
        for (auto& t : pool) t.join();
    }
    if (bad_cov.load()) return fail(h, "vba_pose_optimize: imu_cov_pvphi is singular or not finite");
    char* base = reinterpret_cast<char*>(h->pose_arena.p);
    PoseBatch B;
    B.desc = reinterpret_cast<const FrameDesc*>(base);
    B.pw = reinterpret_cast<const double*>(base + b_desc);
    B.uv = reinterpret_cast<const double*>(base + b_desc + b_pw);
    B.w = reinterpret_cast<const double*>(base + b_desc + b_pw + b_uv);
    B.out = reinterpret_cast<FrameOut*>(base + b_in);
    B.lvl = reinterpret_cast<unsigned char*>(base + b_in + b_out);
    B.err = reinterpret_cast<double*>(base + b_in + b_out + b_lvl);
    B.n_frames = n_frames;
    HIPCHK(h, hipMemcpyAsync(base, hin, b_in, hipMemcpyHostToDevice, h->stream));
    VBA_LAUNCH(k_pose_opt, dim3(n_frames), dim3(64), 0, h->stream, B);
    HIPCHK(h, hipGetLastError());
    char* hout = reinterpret_cast<char*>(h->pose_host_out.p);
    HIPCHK(h, hipMemcpyAsync(hout, base + b_in, b_out + b_lvl, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const FrameOut* res = reinterpret_cast<const FrameOut*>(hout);
    const unsigned char* lvl = reinterpret_cast<const unsigned char*>(hout + b_out);
    for (int f = 0; f < n_frames; f++) {
        vba_frame_problem* F = inout[f];
        vba_frame_result* R = out[f];
        const FrameDesc& d = desc[f];
        const FrameOut& r = res[f];
        R->n_inliers = r.n_inliers; R->status = r.status;
        for (int k = 0; k < 4; k++) { R->its_done[k] = r.its[k]; R->chi2_round[k] = r.chi2_round[k]; }
        std::memcpy(R->marg_cov_inv, r.marg, sizeof r.marg);
        std::memcpy(F->nav, r.nav, sizeof r.nav);
        for (int i = 0; i < d.n_obs; i++) R->outlier[i] = lvl[d.obs0 + i];
        if (R->outlier_last)
            for (int i = 0; i < d.n_last; i++) R->outlier_last[i] = lvl[d.last0 + i];
    }
    return 0;
}

int vba_set_profile(void* handle, int32_t enable) {
    Handle* h = reinterpret_cast<Handle*>(handle);
    if (!h) return -1;
    h->profile = enable != 0;
    return 0;
}
int vba_get_profile(void* handle, vba_profile* out) {
    Handle* h = reinterpret_cast<Handle*>(handle);
    if (!h || !out) return -1;
    *out = h->prof;
    return 0;
}

}  // extern "C"
