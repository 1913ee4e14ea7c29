// vba_host_structure.h -- host half of the structure build (plain C++17, no HIP): included by vislam_ba.hip and by the
// sanitizer harness tests/host_structure_check.cpp (g++ -fsanitize=address,undefined, tests/test_host_structure.py).
#pragma once
#include "../../include/vislam_ba.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#ifndef VBA_NB
#define VBA_NB 32
#endif

namespace vba_host {

inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ---- structure build (g2o BlockSolver::buildStructure analogue, block_solver.hpp:143-295) -------------
// Host half: ONE walk over a window's index arrays validates them and leaves, per landmark, the bitmask of its observing
// keyframes; from the masks come the keyframe-pair occupancy and with it the symbolic tile factorisation, the IMU lists and
// the write masks.  Everything that needs a sort or a per-pair item list is built on the device from the raw arrays
// (vba_structure.h).
struct Structure {
    std::vector<int> pair_a, pair_b, pimu_begin, pimu;
    std::vector<int> step_begin, tpairs, pan_begin, pan;  // tile lists of the factorisation
    std::vector<int> step_npairs;
    std::vector<int> off_pair, pair_mask;
    std::vector<unsigned long long> lmask;  // [n_pt][mwords] observing keyframes of every landmark
    std::vector<int> adj_begin, adj;        // PCG: per free keyframe the other free keyframes its block row of S is non-zero for
    std::vector<int> linblk;                // k_lin2 work split (inverse-depth windows): (p0, p1, e0, e1) per workgroup
    // inverse-depth windows: a workgroup of k_lin2 sums the reference-keyframe terms (G0, g0) of its landmarks over RUNS of
    // consecutive landmarks that share the reference keyframe and writes one record per run instead of one per landmark
    std::vector<int> prun0;                 // per workgroup: id of its first run record (the window has prun0.back() of them)
    std::vector<int> pref_begin, pref_list; // per keyframe the run records it is the reference of (ids ascending)
    int mwords = 1;
    long long item_cap = 0;            // upper bound of the off-diagonal items (exact when every keyframe is free)
    int order = 0;                     // elimination order of the reduced system (see below)
    long long prod_order[3] = {-1, -1, -1}; // tile products of the symbolic factorisation under orders 0 / 1 / 2 (-1: not evaluated)
    std::vector<int> kl_begin, klist;  // left-looking factorisation: per column entry the steps k < J that update it
    // Chain columns: the leading block columns J < nc whose row of L has no tile left of (J, J-1) -- the V/Bias blocks of the IMU chain
    // under the V/Bias-first order.  One launch walks them per window (k_chol_chain) instead of one launch per column.  Few-window
    // regime: `cu` lists the tiles (I,J), J >= nc, that collect updates from chain columns -- (I << 16 | J, first and end position of the chain columns in its k list, 0) per tile
    // (k_chol_chain_upd brings them up to date in one launch before the per-column steps take over).
    int nc = 0;
    int nc_split = 0;                  // > 0: chain columns [0, nc_split) and [nc_split, nc) do not depend on each other (no tile couples
                                       // column nc_split - 1 to column nc_split): the few-window chain kernel walks them side by side
    int nS = 0;                        // rows of the reduced system under the chosen order (a multiple of the tile size)
    std::vector<int> cu;
    std::vector<int> chain_tab;        // per chain column J: (mask lo, mask hi, rides, 0) -- bit q of the mask: tile (nc + q, J) is in the factor;
                                       // rides: tile (J+1, J) is in the factor and J+1 is a chain column
};

// Order 2 ("two-sided", few windows): the V/Bias blocks first like order 0, but as TWO chains that meet in the middle -- keyframes
// 0 .. h-1 ascending from position 0, keyframes nf-1 .. h descending from the next tile boundary -- and the PR blocks from the tile
// boundary behind them.  The chains share no tile, so the chain kernel walks them side by side (7 + 7 block columns in 7 rounds
// instead of 13 in a row for 49 keyframes); the one coupling between them (keyframes h-1 and h) lands in the last tile of the
// second chain, which becomes an ordinary column behind the chains.  Pads between the parts are identity rows like the tail pad.
inline void two_sided_layout(int nf, int& h, int& baseB, int& pr0) {
    int best = 1 << 30;
    h = nf / 2; baseB = 0; pr0 = 0;
    for (int c = std::max(1, nf / 2 - 3); c <= std::min(nf - 1, nf / 2 + 3); c++) {
        const int TA = (9 * c + VBA_NB - 1) / VBA_NB, TB = (9 * (nf - c) + VBA_NB - 1) / VBA_NB;
        const int cost = 64 * std::max(TA, TB - 1) + (TA + TB);
        if (cost < best) { best = cost; h = c; baseB = VBA_NB * TA; pr0 = VBA_NB * (TA + TB); }
    }
}
inline int vpos_host(int order, int pdim, int nf, int a, int r) {
    if (pdim != 15) return 6 * a + r;
    if (order == 2) {
        int h, baseB, pr0;
        two_sided_layout(nf, h, baseB, pr0);
        return r < 6 ? pr0 + 6 * a + r : (a < h ? 9 * a : baseB + 9 * (nf - 1 - a)) + (r - 6);
    }
    return order ? 15 * a + r : (r < 6 ? 9 * nf + 6 * a + r : 9 * a + (r - 6));
}
inline int order_rows(int order, int pdim, int nf) {   // rows of the reduced system before rounding up to the tile size
    if (pdim == 15 && order == 2) {
        int h, baseB, pr0;
        two_sided_layout(nf, h, baseB, pr0);
        return pr0 + 6 * nf;
    }
    return pdim * nf;
}

inline int build_structure(const vba_problem* P, Structure& st, std::string& err, bool prefer_two_sided = false) {
    auto fail = [&err](int, const char* m) { err = m; return -1; };
    const int h = 0;
    static const bool timing = getenv("VBA_TIMING") != nullptr;
    const double t_b0 = timing ? now_ms() : 0.0;
    const int nf = P->n_kf_free, npairs = nf * (nf + 1) / 2, nkf = P->n_kf;
    auto pidx = [nf](int a, int b) { return a * nf - a * (a - 1) / 2 + (b - a); };
    st.pair_a.resize(npairs);
    st.pair_b.resize(npairs);
    st.off_pair.clear();
    st.off_pair.reserve(npairs);
    for (int a = 0; a < nf; a++)
        for (int b = a; b < nf; b++) {
            st.pair_a[pidx(a, b)] = a;
            st.pair_b[pidx(a, b)] = b;
        }
    // The off-diagonal pairs in the order the Schur gather deals them to its waves (four pairs per wave, in lock-step): by
    // DISTANCE b - a, then by a.  Co-visibility falls off with the distance between two keyframes, so the four pairs of a wave
    // then carry similar numbers of items; in (a, b) order a wave held (a,a+1) .. (a,a+4) with 300 .. 150 items and ran at a
    // quarter of its lanes.  (Which wave takes a pair does not enter any sum: results are unchanged bit for bit.)
    for (int dd = 1; dd < nf; dd++)
        for (int a = 0; a + dd < nf; a++) st.off_pair.push_back(pidx(a, a + dd));
    const bool idp = P->variant == VBA_VARIANT_PRV_IDP;
    const int mw = (nkf + 63) / 64;
    st.mwords = mw;
    st.lmask.assign((size_t)P->n_pt * mw, 0ull);
    // occ[a] = the keyframes that share a landmark with free keyframe a (observer or reference), as a bitmask
    std::vector<unsigned long long> occ((size_t)nf * mw, 0ull), tmp(mw);
    st.item_cap = 0;
    if (P->n_pt > 0 && P->pt_obs_begin[0] != 0) return fail(h, "pt_obs_begin is not a valid CSR");
    for (int p = 0; p < P->n_pt; p++) {
        const int o0 = P->pt_obs_begin[p], o1 = P->pt_obs_begin[p + 1];
        if (o0 > o1 || o0 < 0 || o1 > P->n_obs) return fail(h, "pt_obs_begin is not a valid CSR");
        unsigned long long* M = &st.lmask[(size_t)p * mw];
        int rf = -1;
        if (idp) {
            rf = P->pt_ref_kf[p];
            if (rf < 0 || rf >= nkf) return fail(h, "pt_ref_kf out of range");
        }
        for (int o = o0; o < o1; o++) {
            const int kf = P->obs_kf[o];
            if (kf < 0 || kf >= nkf) return fail(h, "obs_kf out of range");
            if (kf == rf) return fail(h, "observation from the reference keyframe is not an edge");
            const unsigned long long bit = 1ull << (kf & 63);
            if (M[kf >> 6] & bit) return fail(h, "a landmark is observed twice from one keyframe");
            M[kf >> 6] |= bit;
        }
        const long long m = o1 - o0;
        st.item_cap += idp ? m * (m + 1) / 2 : m * (m - 1) / 2;
        // every free keyframe of the track (reference included) shares this landmark with every other one
        for (int wd = 0; wd < mw; wd++) tmp[wd] = M[wd];
        if (rf >= 0) tmp[rf >> 6] |= 1ull << (rf & 63);
        for (int wd = 0; wd < mw; wd++) {
            unsigned long long bits = tmp[wd];
            while (bits) {
                const int a = 64 * wd + __builtin_ctzll(bits);
                bits &= bits - 1;
                if (a >= nf) break;
                for (int w2 = 0; w2 < mw; w2++) occ[(size_t)a * mw + w2] |= tmp[w2];
            }
        }
    }
    if (P->pt_obs_begin[P->n_pt] != P->n_obs) return fail(h, "pt_obs_begin does not cover the observations");
    auto pair_vis = [&](int a, int b) { return (occ[(size_t)a * mw + (b >> 6)] >> (b & 63)) & 1ull; };
    {
        // work split of the edge-parallel linearisation (k_lin2, k_lin_xyz_e): runs of consecutive landmarks with <= 256 edges and
        // <= 64 landmarks per workgroup (one 16-B record per workgroup: first / end landmark, first / end edge).  XYZ windows with a
        // longer track fall back to the thread-per-landmark kernel (linblk left empty).
        st.linblk.clear();
        int p = 0;
        while (p < P->n_pt) {
            const int p_first = p;
            int ne = 0, np2 = 0;
            while (p < P->n_pt && np2 < 64) {
                const int k = P->pt_obs_begin[p + 1] - P->pt_obs_begin[p];
                if (k > 256) {
                    if (idp) return fail(h, "a landmark with more than 256 observations is not supported");
                    st.linblk.clear();
                    p = P->n_pt + 1;   // leave both loops
                    break;
                }
                if (ne + k > 256) break;
                ne += k; np2++; p++;
            }
            if (p > P->n_pt) break;
            st.linblk.push_back(p_first); st.linblk.push_back(p);
            st.linblk.push_back(P->pt_obs_begin[p_first]); st.linblk.push_back(P->pt_obs_begin[p]);
        }
    }
    st.prun0.clear(); st.pref_begin.clear(); st.pref_list.clear();
    if (idp) {
        const int nblk = (int)(st.linblk.size() / 4);
        std::vector<int> run_ref;
        st.prun0.reserve(nblk + 1);
        for (int lb = 0; lb < nblk; lb++) {
            st.prun0.push_back((int)run_ref.size());
            for (int p = st.linblk[4 * lb]; p < st.linblk[4 * lb + 1]; p++) {
                const int rf = P->pt_ref_kf ? P->pt_ref_kf[p] : 0;
                if (p == st.linblk[4 * lb] || rf != (P->pt_ref_kf ? P->pt_ref_kf[p - 1] : 0)) run_ref.push_back(rf);
            }
        }
        st.prun0.push_back((int)run_ref.size());
        st.pref_begin.assign(nkf + 1, 0);
        for (int rf : run_ref) st.pref_begin[rf + 1]++;
        for (int k = 0; k < nkf; k++) st.pref_begin[k + 1] += st.pref_begin[k];
        st.pref_list.resize(run_ref.size());
        std::vector<int> fill(st.pref_begin.begin(), st.pref_begin.end() - 1);
        for (int id = 0; id < (int)run_ref.size(); id++) st.pref_list[fill[run_ref[id]]++] = id;
    }
    const double t_b1 = timing ? now_ms() : 0.0;
    // IMU edges per block pair: (edge, role) with role bit0: a is keyframe j of the edge, bit1: b is keyframe j
    const int nimu = (P->variant == VBA_VARIANT_SE3_XYZ) ? 0 : P->n_imu;
    st.pimu_begin.assign(npairs + 1, 0);
    for (int pass = 0; pass < 2; pass++) {
        std::vector<int> fill;
        if (pass) {
            for (int i = 0; i < npairs; i++) st.pimu_begin[i + 1] += st.pimu_begin[i];
            st.pimu.assign(2 * (size_t)st.pimu_begin[npairs], 0);
            fill.assign(st.pimu_begin.begin(), st.pimu_begin.end() - 1);
        }
        auto put = [&](int pi, int k, int role) {
            if (!pass) { st.pimu_begin[pi + 1]++; return; }
            st.pimu[2 * (size_t)fill[pi]] = k;
            st.pimu[2 * (size_t)fill[pi] + 1] = role;
            fill[pi]++;
        };
        for (int k = 0; k < nimu; k++) {
            const int i = P->imu_kf_i[k], j = P->imu_kf_j[k];
            if (i < 0 || j < 0 || i >= nkf || j >= nkf || i == j) return fail(h, "imu keyframe index out of range");
            if (i < nf) put(pidx(i, i), k, 0);
            if (j < nf) put(pidx(j, j), k, 3);
            if (i < nf && j < nf) {
                if (i < j) put(pidx(i, j), k, 2);
                else put(pidx(j, i), k, 1);
            }
        }
    }
    if (P->solver == VBA_SOLVER_PCG) {   // block rows of S: a shares a landmark or an IMU edge with b
        st.adj_begin.assign(nf + 1, 0);
        st.adj.clear();
        for (int a = 0; a < nf; a++) {
            for (int b = 0; b < nf; b++) {
                if (b == a) continue;
                const int pi = (a < b) ? pidx(a, b) : pidx(b, a);
                if (pair_vis(a, b) || st.pimu_begin[pi + 1] > st.pimu_begin[pi]) st.adj.push_back(b);
            }
            st.adj_begin[a + 1] = (int)st.adj.size();
        }
    }
    // symbolic factorisation on 32x32 tiles (the tile-level analogue of SimplicialLDLT::analyzePattern,
    // linear_solver_eigen.h:147-152): which tiles of L can be nonzero.  Two elimination orders are tried and the cheaper
    // one (tile products of the factorisation) kept -- g2o lets AMD pick an order; the reduced system here is either
    //   order 0: all V/Bias blocks first, PR blocks last -- the IMU chain stays a narrow band, the PR block fills in
    //            completely: best when most keyframes share landmarks with most others (the usual local window)
    //   order 1: keyframe by keyframe [PR_a V_a Bias_a] -- a block band whose width is the co-visibility span: best
    //            for long, thin windows and for maps
    const int pdim = (P->variant == VBA_VARIANT_SE3_XYZ) ? 6 : 15;
    auto symbolic = [&](int order, bool lists) -> long long {
    const int np = order_rows(order, pdim, nf), nS = ((np + VBA_NB - 1) / VBA_NB) * VBA_NB, nb = nS / VBA_NB;
    st.nS = nS;
    // first and last tile of a keyframe's PR run (sub-block 0) and of its V/Bias run (sub-block 1) under this order
    std::vector<int> trun((size_t)4 * nf);
    for (int a = 0; a < nf; a++) {
        trun[4 * a] = vpos_host(order, pdim, nf, a, 0) / VBA_NB;
        trun[4 * a + 1] = vpos_host(order, pdim, nf, a, 5) / VBA_NB;
        trun[4 * a + 2] = pdim == 15 ? vpos_host(order, pdim, nf, a, 6) / VBA_NB : 0;
        trun[4 * a + 3] = pdim == 15 ? vpos_host(order, pdim, nf, a, 14) / VBA_NB : 0;
    }
    st.tpairs.clear(); st.pan.clear();
    std::vector<unsigned char> T((size_t)nb * nb, 0);
    for (int i = 0; i < nb; i++) T[(size_t)i * nb + i] = 1;
    for (int pi = 0; pi < npairs; pi++) {
        const int a = st.pair_a[pi], b = st.pair_b[pi];
        const bool vis = pair_vis(a, b);
        const bool imu = st.pimu_begin[pi + 1] > st.pimu_begin[pi];
        if (!vis && !imu && a != b) continue;
        // a keyframe's PR dofs (0..5) and V/Bias dofs (6..14) are two contiguous runs: each touches at most two tiles
        const int nsub = ((imu || a == b) && pdim == 15) ? 2 : 1;
        for (int sr = 0; sr < nsub; sr++)
            for (int sc = 0; sc < nsub; sc++) {
                const int ti0 = trun[4 * a + 2 * sr], ti1 = trun[4 * a + 2 * sr + 1];
                const int tj0 = trun[4 * b + 2 * sc], tj1 = trun[4 * b + 2 * sc + 1];
                for (int ti = ti0; ti <= ti1; ti++)
                    for (int tj = tj0; tj <= tj1; tj++) T[(size_t)std::max(ti, tj) * nb + std::min(ti, tj)] = 1;
            }
    }
    st.step_begin.assign(nb + 1, 0);
    st.pan_begin.assign(nb + 1, 0);
    st.step_npairs.assign(nb, 0);
    std::vector<int> pk;
    long long cost = 0;
    for (int k = 0; k < nb; k++) {
        pk.clear();
        for (int I = k + 1; I < nb; I++)
            if (T[(size_t)I * nb + k]) pk.push_back(I);
        for (int I : pk) st.tpairs.push_back((I << 16) | I);  // diagonal pairs first: pair 0 owns y_k
        for (size_t i = 0; i < pk.size(); i++)
            for (size_t j = 0; j < i; j++) {
                st.tpairs.push_back((pk[i] << 16) | pk[j]);
                T[(size_t)pk[i] * nb + pk[j]] = 1;  // fill
            }
        for (int I : pk) st.pan.push_back(I);
        st.step_begin[k + 1] = (int)st.tpairs.size();
        st.pan_begin[k + 1] = (int)st.pan.size();
        st.step_npairs[k] = st.step_begin[k + 1] - st.step_begin[k];
    }
    cost = (long long)st.tpairs.size();   // one tile product per update pair (= the k-list entries of the left-looking form)
    if (!lists) return cost;
    // left-looking lists: column entries in the order (J,J), then (I,J) for I in the panel of J; K(I,J) = {k < J : L_Ik, L_Jk != 0}
    st.kl_begin.clear();
    st.klist.clear();
    for (int J = 0; J < nb; J++) {
        for (int e = -1; e < st.pan_begin[J + 1] - st.pan_begin[J]; e++) {
            const int I = (e < 0) ? J : st.pan[st.pan_begin[J] + e];
            st.kl_begin.push_back((int)st.klist.size());
            for (int k = 0; k < J; k++)
                if (T[(size_t)I * nb + k] && T[(size_t)J * nb + k]) st.klist.push_back(k);
        }
    }
    st.kl_begin.push_back((int)st.klist.size());
    {   // chain columns (see Structure::nc): T holds L's pattern
        static const int chain_min = getenv("VBA_CHAIN_MIN") ? atoi(getenv("VBA_CHAIN_MIN")) : 4;
        int nc = 0;
        for (int J = 0; J < nb; J++) {
            bool ok = true;
            for (int k = 0; k + 1 < J && ok; k++) ok = !T[(size_t)J * nb + k];
            if (!ok) break;
            nc = J + 1;
        }
        if (nc == nb && nc > 0) nc--;   // (the last column has nothing to its right: leave it to the per-column kernels)
        st.nc = (chain_min > 0 && nc >= chain_min) ? nc : 0;
        st.nc_split = 0;
        for (int J = 1; J < st.nc; J++)    // a column whose predecessor does not couple to it, nearest to the middle
            if (!T[(size_t)J * nb + (J - 1)] && std::abs(2 * J - st.nc) < std::abs(2 * st.nc_split - st.nc)) st.nc_split = J;
        if (nb - nc > 64) st.nc = 0;      // (the row masks of chain_tab are 64 bits wide)
        if (!st.nc) st.nc_split = 0;
        st.chain_tab.clear();
        for (int J = 0; J < st.nc; J++) {
            unsigned long long mask = 0;
            int ride = 0;
            for (int i = st.pan_begin[J]; i < st.pan_begin[J + 1]; i++) {
                const int I = st.pan[i];
                if (I == J + 1 && J + 1 < st.nc) ride = 1;
                else if (I >= st.nc) mask |= 1ull << (I - st.nc);
            }
            st.chain_tab.push_back((int)(unsigned)(mask & 0xffffffffull));
            st.chain_tab.push_back((int)(unsigned)(mask >> 32));
            st.chain_tab.push_back(ride);
            st.chain_tab.push_back(0);
        }
        st.cu.clear();
        for (int J = st.nc; J < nb && st.nc > 0; J++)
            for (int e = -1; e < st.pan_begin[J + 1] - st.pan_begin[J]; e++) {
                const int I = (e < 0) ? J : st.pan[st.pan_begin[J] + e];
                const int ent = st.pan_begin[J] + J + 1 + e;
                if (st.kl_begin[ent + 1] > st.kl_begin[ent] && st.klist[st.kl_begin[ent]] < st.nc) {
                    int ke = st.kl_begin[ent + 1];
                    while (ke > st.kl_begin[ent] && st.klist[ke - 1] >= st.nc) ke--;   // (ascending list: the chain columns are its head)
                    st.cu.push_back((I << 16) | J);
                    st.cu.push_back(st.kl_begin[ent]);
                    st.cu.push_back(ke);
                    st.cu.push_back(0);
                }
            }
    }
    // which sub-blocks of a keyframe pair's block can land in a tile the factorisation reads (T now holds L's pattern)
    st.pair_mask.assign(npairs, 0);
    for (int pi = 0; pi < npairs; pi++) {
        const int a = st.pair_a[pi], b = st.pair_b[pi];
        int mask = 0;
        const int nsub = (pdim == 15) ? 2 : 1;
        for (int sr = 0; sr < nsub; sr++)
            for (int sc = 0; sc < nsub; sc++) {
                const int ti0 = trun[4 * a + 2 * sr], ti1 = trun[4 * a + 2 * sr + 1];
                const int tj0 = trun[4 * b + 2 * sc], tj1 = trun[4 * b + 2 * sc + 1];
                bool any = false;
                for (int ti = ti0; ti <= ti1; ti++)
                    for (int tj = tj0; tj <= tj1; tj++) any = any || T[(size_t)std::max(ti, tj) * nb + std::min(ti, tj)];
                if (any) mask |= 1 << ((sr ? 2 : 0) + (sc ? 1 : 0));
            }
        if (pair_vis(a, b)) mask |= 16;   // bit 4: the two keyframes share a landmark (the pair has items)
        st.pair_mask[pi] = mask;
    }
    return cost;
    };
    static const int env_order = getenv("VBA_ORDER") ? atoi(getenv("VBA_ORDER")) : -1;
    st.order = 0;
    if (pdim == 15) {
        if (env_order >= 0) st.order = (env_order >= 2 && (nf < 4 || P->solver == VBA_SOLVER_PCG)) ? 0 : std::min(env_order, 2);   // (PCG walks keyframe-pair blocks: orders 0 / 1)
        else {
            const long long c0 = symbolic(0, false), c1 = symbolic(1, false);
            const long long c2 = (prefer_two_sided && nf >= 12 && P->solver != VBA_SOLVER_PCG) ? symbolic(2, false) : -1;
            st.prod_order[0] = c0; st.prod_order[1] = c1; st.prod_order[2] = c2;
            const long long cv = (c2 >= 0 && c2 < c0) ? c2 : c0;   // the better V/Bias-first variant
            // keyframe order only on a clear win for a local window (a batch that mixes patterns pays for each, and the chain kernels make
            // a V/Bias-first tile product cheaper); for a map (> 64 keyframes: a chain of dozens of block columns walked in sequence,
            // few windows per call) the fewer products decide -- C4, two kinds of graph (products 8.1 k keyframe order, 14.6 k two-sided):
            // all keyframe order 839 solves/s, all two-sided 827, one kind each 798
            const bool kf_order = (nf > 64) ? (c1 < cv) : (10 * c1 < 7 * cv);
            st.order = kf_order ? 1 : (cv == c0 ? 0 : 2);
            if (timing) fprintf(stderr, "[vba] tile products of the factorisation: V/Bias-first %lld, keyframe order %lld, two-sided %lld\n", c0, c1, c2);
        }
    }
    symbolic(st.order, true);
    st.off_pair.resize(npairs, 0);  // padded to the pair stride
    if (timing) fprintf(stderr, "[vba] structure (host): validation + masks %.3f ms, IMU lists + symbolic factorisation %.3f ms\n", t_b1 - t_b0, now_ms() - t_b1);
    return 0;
}


}  // namespace vba_host
