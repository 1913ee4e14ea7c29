// vba_structure.h -- the data-dependent half of g2o's BlockSolver::buildStructure (block_solver.hpp:143-295) on the device.
//
// The host validates a window's index arrays in one walk (and leaves, per landmark, the bitmask of its observing
// keyframes); everything that needs a sort or a per-pair list is built here, from the raw arrays the caller handed over:
//   lm_order            landmarks by the first keyframe of their track (stable)
//   slot_perm / kf_seg  record position of every observation: KEYFRAME-major, landmarks in lm_order inside a keyframe
//   pt_perm / ref_seg   record position of every landmark: by reference keyframe, lm_order inside (inverse-depth windows)
//   item lists          per off-diagonal keyframe pair (a < b) the (record_a, record_b) of every landmark both see:
//                       [item_begin, item_mid) pairs of two observation records, [item_mid, item_end) pairs that involve
//                       the landmark's reference keyframe (these also carry a direct H_pp term)
// All of it is a stable counting sort or a ranked compaction, done with wave ballots: a wave walks a sequence 64
// entries at a time, `ballot(predicate)` gives the members of the bucket among them, the popcount below a lane its rank
// -- no atomics decide an order, so the structure (and with it every summation order of the solve) is reproducible.
// The diagonal pair (a,a) needs no list: its items are exactly the records of keyframe a, i.e. two index ranges.
#pragma once
#include "vba_kernels.h"

struct StBuild {   // what the build writes (the solve reads the same arrays through Batch, as const)
    int *obs_pt, *slot_perm, *pt_perm, *kf_seg, *ref_seg, *item_begin, *item_mid, *items;
    int *st_key, *lm_order, *slot_obs, *pt_inv;   // scratch: first keyframe of a track, landmark at rank q, observation in slot s, landmark in record r
};

typedef unsigned long long u64_t;

DEVI u64_t wave_or64(u64_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o, 64);
    return v;
}
DEVI u64_t lanes_below() { return (1ull << (threadIdx.x & 63)) - 1ull; }

// record position of landmark p's observation from keyframe b (the host has checked that there is exactly one)
DEVI int st_slot_of(const Batch& B, const StBuild& T, const WinDesc& d, const int* ob, int p, int b) {
    for (int o = ob[p]; o < ob[p + 1]; o++)
        if (B.obs_kf[d.obs0 + o] == b) return T.slot_perm[d.obs0 + o];
    return 0;
}

// One workgroup per window: histograms -> segment starts, then the two ranked walks.
__global__ void __launch_bounds__(256) k_st_order(Batch B, StBuild T) {
    extern __shared__ int sh[];
    const int w = blockIdx.x;
    const WinDesc& d = B.desc[w];
    const int nk = d.n_kf, npt = d.n_pt, t = threadIdx.x, mw = d.mwords;
    const bool idp = d.variant == 2;
    int* hk = sh;                 // landmarks per first keyframe
    int* ho = sh + (nk + 1);      // observations per observing keyframe
    int* hr = ho + (nk + 1);      // landmarks per reference keyframe
    for (int i = t; i < 3 * (nk + 1); i += 256) sh[i] = 0;
    __syncthreads();
    const int* ob = B.pt_obs_begin + d.pt0 + d.win;
    const u64_t* LM = B.lmask + d.mask0;
    for (int p = t; p < npt; p += 256) {
        const int ref = B.pt_ref[d.pt0 + p];
        int key = idp ? ref : nk - 1;
        for (int wd = 0; wd < mw; wd++) {
            const u64_t m = LM[(size_t)p * mw + wd];
            if (m) { key = min(key, 64 * wd + (int)__builtin_ctzll(m)); break; }
        }
        T.st_key[d.pt0 + p] = key;
        atomicAdd(&hk[key + 1], 1);   // integer counts: the result does not depend on the order of the adds
        if (idp) atomicAdd(&hr[ref + 1], 1);
        for (int o = ob[p]; o < ob[p + 1]; o++) T.obs_pt[d.obs0 + o] = p;
    }
    for (int o = t; o < d.n_obs; o += 256) atomicAdd(&ho[B.obs_kf[d.obs0 + o] + 1], 1);
    __syncthreads();
    if (t < 3) {
        int* h = sh + t * (nk + 1);
        for (int i = 0; i < nk; i++) h[i + 1] += h[i];
    }
    __syncthreads();
    for (int i = t; i <= nk; i += 256) {
        T.kf_seg[d.kf0 + d.win + i] = ho[i];
        T.ref_seg[d.kf0 + d.win + i] = idp ? hr[i] : 0;
    }
    const int wave = t >> 6, lane = t & 63;
    const u64_t lt = lanes_below();
    // A. landmarks by (first keyframe, index): one wave per bucket walks the landmarks in index order
    for (int k = wave; k < nk; k += 4) {
        int base = hk[k];
        const int end = hk[k + 1];
        for (int c = 0; c < npt && base < end; c += 64) {
            const int p = c + lane;
            const bool has = p < npt && T.st_key[d.pt0 + p] == k;
            const u64_t m = __ballot(has);
            if (has) T.lm_order[d.pt0 + base + __popcll(m & lt)] = p;
            base += __popcll(m);
        }
    }
    __syncthreads();
    // B. observation records by (observing keyframe, lm_order) and landmark records by (reference keyframe, lm_order): one
    //    wave per keyframe walks the landmarks in lm_order
    for (int k = wave; k < nk; k += 4) {
        int bo = ho[k], br = hr[k];
        const int eo = ho[k + 1], er = hr[k + 1];
        const int kw = k >> 6, kb = k & 63;
        for (int c = 0; c < npt && (bo < eo || br < er); c += 64) {
            const int q = c + lane;
            const bool valid = q < npt;
            const int p = valid ? T.lm_order[d.pt0 + q] : 0;
            const bool haso = valid && ((LM[(size_t)p * mw + kw] >> kb) & 1ull);
            const u64_t mo = __ballot(haso);
            if (haso) {
                const int slot = bo + __popcll(mo & lt);
                int o = ob[p];
                while (B.obs_kf[d.obs0 + o] != k) o++;
                T.slot_perm[d.obs0 + o] = slot;
                T.slot_obs[d.obs0 + slot] = o;
            }
            bo += __popcll(mo);
            const bool hasr = valid && idp && B.pt_ref[d.pt0 + p] == k;
            const u64_t mr = __ballot(hasr);
            if (hasr) {
                const int r = br + __popcll(mr & lt);
                T.pt_perm[d.pt0 + p] = r;
                T.pt_inv[d.pt0 + r] = p;
            }
            br += __popcll(mr);
        }
    }
    if (!idp)   // XYZ landmarks have no reference keyframe: their point records stay in landmark order
        for (int p = t; p < npt; p += 256) T.pt_perm[d.pt0 + p] = p;
}

// One wave per (window, free keyframe a): walks the records of a -- its observation records in slot order, then the landmark
// records it is the reference of -- and, for every later free keyframe b some of those landmarks are also seen from (or
// referenced by), ranks the members.  FILL = false counts per pair, FILL = true writes the items at the scanned offsets.
template <bool FILL>
DEVI void st_row_body(const Batch& B, const StBuild& T, int* c0, int* c1) {
    const int w = blockIdx.y, a = blockIdx.x;
    const WinDesc& d = B.desc[w];
    const int nf = d.n_free;
    if (a >= nf) return;
    const int lane = threadIdx.x, mw = d.mwords;
    const bool idp = d.variant == 2;
    const int rowbase = a * nf - a * (a - 1) / 2 - a;   // pair index of (a, b) = rowbase + b
    int* ib = T.item_begin + d.pair0 + d.win;
    int* im = T.item_mid + d.pair0 + d.win;
    for (int b = lane; b < nf; b += 64) {
        c0[b] = (FILL && b > a) ? ib[rowbase + b] : 0;
        c1[b] = (FILL && b > a) ? im[rowbase + b] : 0;
    }
    __syncthreads();
    const int* kseg = T.kf_seg + d.kf0 + d.win;
    const int* rseg = T.ref_seg + d.kf0 + d.win;
    const int* ob = B.pt_obs_begin + d.pt0 + d.win;
    const u64_t* LM = B.lmask + d.mask0;
    const u64_t lt = lanes_below();
    int2* items = reinterpret_cast<int2*>(T.items) + d.item0;
    // bits of mask word wd that name a keyframe b with a < b < nf
    auto range = [&](int wd) -> u64_t {
        const int lo = 64 * wd;
        u64_t r = ~0ull;
        if (a + 1 > lo) r &= (a + 1 - lo >= 64) ? 0ull : (~0ull << (a + 1 - lo));
        if (nf < lo + 64) r &= (nf <= lo) ? 0ull : (~0ull >> (lo + 64 - nf));
        return r;
    };
    // 1. observation records of a
    for (int c = kseg[a]; c < kseg[a + 1]; c += 64) {
        const int slot = c + lane;
        const bool valid = slot < kseg[a + 1];
        const int o = valid ? T.slot_obs[d.obs0 + slot] : 0;
        const int p = valid ? T.obs_pt[d.obs0 + o] : 0;
        const int r = (valid && idp) ? B.pt_ref[d.pt0 + p] : -1;
        for (int wd = a >> 6; wd < mw; wd++) {
            const u64_t rg = range(wd);
            if (!rg) continue;
            const u64_t Mr = valid ? (LM[(size_t)p * mw + wd] & rg) : 0ull;
            const u64_t rb = (r > a && r < nf && (r >> 6) == wd) ? (1ull << (r & 63)) : 0ull;
            u64_t U = wave_or64(Mr | rb);
            while (U) {
                const int bb = __builtin_ctzll(U);
                U &= U - 1;
                const int b = 64 * wd + bb;
                const bool h0 = (Mr >> bb) & 1ull, h1 = (rb >> bb) & 1ull;
                const u64_t m0 = __ballot(h0), m1 = __ballot(h1);
                if (FILL) {
                    if (h0) items[c0[b] + __popcll(m0 & lt)] = make_int2(slot, st_slot_of(B, T, d, ob, p, b));
                    if (h1) items[c1[b] + __popcll(m1 & lt)] = make_int2(slot, d.n_obs + T.pt_perm[d.pt0 + p]);
                }
                __syncthreads();   // one wave: orders the LDS reads above before lane 0's update
                if (lane == 0) { c0[b] += __popcll(m0); c1[b] += __popcll(m1); }
                __syncthreads();
            }
        }
    }
    // 2. landmark records a is the reference keyframe of: reference items only
    if (idp)
        for (int c = rseg[a]; c < rseg[a + 1]; c += 64) {
            const int rec = c + lane;
            const bool valid = rec < rseg[a + 1];
            const int p = valid ? T.pt_inv[d.pt0 + rec] : 0;
            for (int wd = a >> 6; wd < mw; wd++) {
                const u64_t rg = range(wd);
                if (!rg) continue;
                const u64_t Mr = valid ? (LM[(size_t)p * mw + wd] & rg) : 0ull;
                u64_t U = wave_or64(Mr);
                while (U) {
                    const int bb = __builtin_ctzll(U);
                    U &= U - 1;
                    const int b = 64 * wd + bb;
                    const bool h1 = (Mr >> bb) & 1ull;
                    const u64_t m1 = __ballot(h1);
                    if (FILL && h1) items[c1[b] + __popcll(m1 & lt)] = make_int2(d.n_obs + rec, st_slot_of(B, T, d, ob, p, b));
                    __syncthreads();
                    if (lane == 0) c1[b] += __popcll(m1);
                    __syncthreads();
                }
            }
        }
    if (!FILL) {
        for (int b = lane; b < nf; b += 64) {
            if (b > a) { ib[rowbase + b] = c0[b] + c1[b]; im[rowbase + b] = c0[b]; }
            else if (b == a) { ib[rowbase + b] = 0; im[rowbase + b] = 0; }   // the diagonal pair has no list
        }
    }
}
__global__ void __launch_bounds__(64) k_st_count(Batch B, StBuild T, int max_free) {
    extern __shared__ int shc[];
    st_row_body<false>(B, T, shc, shc + max_free);
}
__global__ void __launch_bounds__(64) k_st_fill(Batch B, StBuild T, int max_free) {
    extern __shared__ int shc[];
    st_row_body<true>(B, T, shc, shc + max_free);
}

// per-pair counts -> offsets (exclusive scan over the pairs of a window); item_mid = first reference item of the pair
__global__ void __launch_bounds__(256) k_st_scan(Batch B, StBuild T) {
    __shared__ int part[256];
    const int w = blockIdx.x, t = threadIdx.x;
    const WinDesc& d = B.desc[w];
    const int n = d.n_pairs;
    int* ib = T.item_begin + d.pair0 + d.win;
    int* im = T.item_mid + d.pair0 + d.win;
    const int per = (n + 255) / 256, s = min(n, t * per), e = min(n, s + per);
    int sum = 0;
    for (int i = s; i < e; i++) sum += ib[i];
    part[t] = sum;
    __syncthreads();
    if (t == 0) {
        int run = 0;
        for (int i = 0; i < 256; i++) { const int v = part[i]; part[i] = run; run += v; }
        ib[n] = run;
        im[n] = run;
    }
    __syncthreads();
    int run = part[t];
    for (int i = s; i < e; i++) {
        const int cnt = ib[i], n0 = im[i];
        ib[i] = run;
        im[i] = run + n0;
        run += cnt;
    }
}
