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

typedef unsigned long long u64_t;

struct StBuild {   // what the build writes (the solve reads the same arrays through Batch, as const)
    int *obs_pt, *slot_perm, *pt_perm, *kf_seg, *ref_seg, *item_begin, *item_mid, *items;
    int *st_key, *lm_order, *slot_obs, *pt_inv;   // scratch: first keyframe of a track, landmark at rank q, landmark of the record in slot s, landmark in record r
    int *key_seg, *tslot;                         // scratch: landmarks per first keyframe (starts); per landmark its slots in keyframe order
    u64_t *mask_q, *slot_mask;                    // scratch: the landmark masks in lm_order / per slot record, so that the ranked
                                                  // walks read them in sequence instead of gathering them through an index
    int *ref_q;                                   // scratch: reference keyframe of the landmark at rank q
    int *slot_o;                                  // observation of the record in slot s (k_stage_mark walks a keyframe's edges through it)
    int smw;                                      // words per slot_mask entry (the largest mwords of the batch)
};

DEVI u64_t wave_or64(u64_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o, 64);
    return v;
}
DEVI u64_t lanes_below() { return (1ull << (threadIdx.x & 63)) - 1ull; }

// record position of landmark p's observation from keyframe b: the landmark's slots are kept in KEYFRAME order behind its
// CSR row (tslot), so the position inside the row is the number of observing keyframes below b
DEVI int st_slot_of(const StBuild& T, const WinDesc& d, const u64_t* LM, const int* ob, int p, int b) {
    const u64_t* M = LM + (size_t)p * d.mwords;
    int r = __popcll(M[b >> 6] & ((1ull << (b & 63)) - 1ull));
    for (int wd = 0; wd < (b >> 6); wd++) r += __popcll(M[wd]);
    return T.tslot[d.obs0 + ob[p] + r];
}

// The first eight slots of a landmark's row, fetched together BEFORE the walk over the partner keyframes: the lookups inside
// that walk then cost no memory round trip (tracks longer than eight fall back to the table)
struct SlotRow {
    int v[8];
    DEVI void load(const StBuild& T, const WinDesc& d, const int* ob, int p, bool valid) {
        const int o0 = valid ? ob[p] : 0, n = valid ? ob[p + 1] - o0 : 0;
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = (i < n) ? T.tslot[d.obs0 + o0 + i] : 0;
    }
    DEVI int at(const StBuild& T, const WinDesc& d, const u64_t* LM, const int* ob, int p, int b) const {
        const u64_t* M = LM + (size_t)p * d.mwords;
        int r = __popcll(M[b >> 6] & ((1ull << (b & 63)) - 1ull));
        for (int wd = 0; wd < (b >> 6); wd++) r += __popcll(M[wd]);
        if (r >= 8) return T.tslot[d.obs0 + ob[p] + r];
        int x = v[0];
#pragma unroll
        for (int i = 1; i < 8; i++) x = (r == i) ? v[i] : x;
        return x;
    }
};

// 1. One workgroup per window: first keyframe of every track, observation -> landmark, and the three histograms with their
//    prefix sums = the segment starts (landmarks per first keyframe, observations per observer, landmarks per reference).
__global__ void __launch_bounds__(1024) k_st_hist(Batch B, StBuild T) {
    extern __shared__ int sh[];
    const int w = blockIdx.x;
    const WinDesc& d = B.desc[w];
    const int nk = d.n_kf, npt = d.n_pt, t = threadIdx.x, mw = d.mwords, nt = blockDim.x;
    const bool idp = d.variant == 2;
    int* hk = sh;                 // landmarks per first keyframe
    int* ho = sh + (nk + 1);      // observations per observing keyframe
    int* hr = ho + (nk + 1);      // landmarks per reference keyframe
    for (int i = t; i < 3 * (nk + 1); i += nt) sh[i] = 0;
    __syncthreads();
    const int* ob = B.pt_obs_begin + d.pt0 + d.win;
    const u64_t* LM = B.lmask + d.mask0;
    for (int p = t; p < npt; p += nt) {
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
        if (!idp) T.pt_perm[d.pt0 + p] = p;   // XYZ landmarks have no reference keyframe: point records stay in landmark order
    }
    for (int o = t; o < d.n_obs; o += nt) atomicAdd(&ho[B.obs_kf[d.obs0 + o] + 1], 1);
    __syncthreads();
    if (t < 3) {
        int* h = sh + t * (nk + 1);
        for (int i = 0; i < nk; i++) h[i + 1] += h[i];
    }
    __syncthreads();
    for (int i = t; i <= nk; i += nt) {
        T.key_seg[d.kf0 + d.win + i] = hk[i];
        T.kf_seg[d.kf0 + d.win + i] = ho[i];
        T.ref_seg[d.kf0 + d.win + i] = idp ? hr[i] : 0;
    }
}

// 2. landmarks by (first keyframe, index): one wave per (window, bucket) walks the landmarks in index order
__global__ void __launch_bounds__(64) k_st_rank_lm(Batch B, StBuild T) {
    const int w = blockIdx.y, k = blockIdx.x;
    const WinDesc& d = B.desc[w];
    if (k >= d.n_kf) return;
    const int npt = d.n_pt, lane = threadIdx.x;
    int base = T.key_seg[d.kf0 + d.win + k];
    const int end = T.key_seg[d.kf0 + d.win + k + 1];
    const u64_t lt = lanes_below();
    for (int c = 0; c < npt && base < end; c += 64) {
        const int p = c + lane;
        const bool has = p < npt && T.st_key[d.pt0 + p] == k;
        const u64_t m = __ballot(has);
        if (has) {
            const int q = base + __popcll(m & lt);
            T.lm_order[d.pt0 + q] = p;
            T.ref_q[d.pt0 + q] = B.pt_ref[d.pt0 + p];
            for (int wd = 0; wd < d.mwords; wd++) T.mask_q[d.mask0 + (size_t)q * d.mwords + wd] = B.lmask[d.mask0 + (size_t)p * d.mwords + wd];
        }
        base += __popcll(m);
    }
}

// 3. observation records by (observing keyframe, lm_order) and landmark records by (reference keyframe, lm_order): one wave
//    per (window, keyframe) walks the landmarks in lm_order
__global__ void __launch_bounds__(64) k_st_rank_rec(Batch B, StBuild T) {
    const int w = blockIdx.y, k = blockIdx.x;
    const WinDesc& d = B.desc[w];
    if (k >= d.n_kf) return;
    const int npt = d.n_pt, lane = threadIdx.x, mw = d.mwords;
    const bool idp = d.variant == 2;
    const int* ob = B.pt_obs_begin + d.pt0 + d.win;
    const u64_t* LM = B.lmask + d.mask0;
    int bo = T.kf_seg[d.kf0 + d.win + k], br = T.ref_seg[d.kf0 + d.win + k];
    const int eo = T.kf_seg[d.kf0 + d.win + k + 1], er = T.ref_seg[d.kf0 + d.win + k + 1];
    const int kw = k >> 6, kb = k & 63;
    const u64_t lt = lanes_below();
    // a track that contains k starts at k or before: only the landmarks of the first k + 1 buckets of lm_order can be members
    const int q_end = min(npt, T.key_seg[d.kf0 + d.win + k + 1]);
    for (int c = 0; c < q_end && (bo < eo || br < er); c += 64) {
        const int q = c + lane;
        const bool valid = q < q_end;
        const u64_t* M = T.mask_q + d.mask0 + (size_t)(valid ? q : 0) * mw;   // masks in lm_order: read in sequence
        const bool haso = valid && ((M[kw] >> kb) & 1ull);
        const bool hasr = valid && idp && T.ref_q[d.pt0 + q] == k;
        const int p = (haso || hasr) ? T.lm_order[d.pt0 + q] : 0;
        const u64_t mo = __ballot(haso);
        if (haso) {
            const int slot = bo + __popcll(mo & lt);
            int o = ob[p];
            while (B.obs_kf[d.obs0 + o] != k) o++;
            T.slot_perm[d.obs0 + o] = slot;
            T.slot_obs[d.obs0 + slot] = p;   // the landmark of the record in this slot
            T.slot_o[d.obs0 + slot] = o;
            for (int wd = 0; wd < mw; wd++) T.slot_mask[(size_t)(d.obs0 + slot) * T.smw + wd] = M[wd];
            int r = __popcll(M[kw] & ((1ull << kb) - 1ull));
            for (int wd = 0; wd < kw; wd++) r += __popcll(M[wd]);
            T.tslot[d.obs0 + ob[p] + r] = slot;   // the landmark's slots in keyframe order (st_slot_of)
        }
        bo += __popcll(mo);
        const u64_t mr = __ballot(hasr);
        if (hasr) {
            const int r = br + __popcll(mr & lt);
            T.pt_perm[d.pt0 + p] = r;
            T.pt_inv[d.pt0 + r] = p;
        }
        br += __popcll(mr);
    }
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
        const int p = valid ? T.slot_obs[d.obs0 + slot] : 0;
        const int r = (valid && idp) ? B.pt_ref[d.pt0 + p] : -1;
        SlotRow sr;
        if (FILL) sr.load(T, d, ob, p, valid);
        for (int wd = a >> 6; wd < mw; wd++) {
            const u64_t rg = range(wd);
            if (!rg) continue;
            const u64_t Mr = valid ? (T.slot_mask[(size_t)(d.obs0 + slot) * T.smw + wd] & rg) : 0ull;
            const u64_t rb = (r > a && r < nf && (r >> 6) == wd) ? (1ull << (r & 63)) : 0ull;
            u64_t U = wave_or64(Mr | rb);
            while (U) {
                const int bb = __builtin_ctzll(U);
                U &= U - 1;
                const int b = 64 * wd + bb;
                const bool h0 = (Mr >> bb) & 1ull, h1 = (rb >> bb) & 1ull;
                const u64_t m0 = __ballot(h0), m1 = __ballot(h1);
                if (FILL) {
                    if (h0) items[c0[b] + __popcll(m0 & lt)] = make_int2(slot, sr.at(T, d, LM, ob, p, b));
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
            SlotRow sr;
            if (FILL) sr.load(T, d, ob, p, valid);
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
                    if (FILL && h1) items[c1[b] + __popcll(m1 & lt)] = make_int2(d.n_obs + rec, sr.at(T, d, LM, ob, p, b));
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
