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
    int *rec_cnt;                                 // scratch: per chunk of 64 landmarks (lm_order) and keyframe the observed / referenced counts, then their prefix
    int *slot_ref, *slot_q, *rec_q, *tsq;         // scratch, so that the pair-row walk gathers nothing through a landmark index: reference
                                                  // keyframe and lm_order rank of the landmark in slot s, rank of the landmark in record r,
                                                  // and per RANK the first eight slots of the landmark's row (neighbouring records of a
                                                  // keyframe are neighbouring ranks: their rows share cache lines)
    int smw;                                      // words per slot_mask entry (the largest mwords of the batch)
    int row_lds;                                  // test hook (VBA_ST_ROW_LDS): the pair-row walks keep their counts in LDS also for windows of <= 64 keyframes
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
// The first eight record positions of a landmark (rank-major, tsq), as two vector registers passed by value: position r of the
// landmark's track is picked by name.  (A struct with an int[8] member, and later one with two int4 members, was kept in scratch memory.)
DEVI void slot_row_load(const StBuild& T, const WinDesc& d, int q, bool valid, int4& lo, int4& hi) {   // q: lm_order rank of the landmark
    const int4* row = reinterpret_cast<const int4*>(T.tsq + 8 * (size_t)(d.pt0 + (valid ? q : 0)));
    lo = row[0]; hi = row[1];   // entries beyond the track length are never selected
}
// m0: word 0 of the landmark's mask, already in a register (windows of <= 64 keyframes need nothing else)
DEVI int slot_row_at(int4 lo, int4 hi, const StBuild& T, const WinDesc& d, const u64_t* LM, const int* ob, int p, int b, u64_t m0) {
    int r;
    if (d.mwords == 1) r = __popcll(m0 & ((1ull << (b & 63)) - 1ull));
    else {
        const u64_t* M = LM + (size_t)p * d.mwords;
        r = __popcll(M[b >> 6] & ((1ull << (b & 63)) - 1ull));
        for (int wd = 0; wd < (b >> 6); wd++) r += __popcll(M[wd]);
    }
    if (r >= 8) return T.tslot[d.obs0 + ob[p] + r];
    int x = lo.x;
    x = (r == 1) ? lo.y : x; x = (r == 2) ? lo.z : x; x = (r == 3) ? lo.w : x;
    x = (r == 4) ? hi.x : x; x = (r == 5) ? hi.y : x; x = (r == 6) ? hi.z : x; x = (r == 7) ? hi.w : x;
    return x;
}

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
    for (int o0 = t; o0 < d.n_obs; o0 += 8 * nt) {   // eight loads in flight per thread (a loop of load -> LDS atomic runs one at a time)
        int kf[8];
#pragma unroll
        for (int i = 0; i < 8; i++) kf[i] = (o0 + i * nt < d.n_obs) ? B.obs_kf[d.obs0 + o0 + i * nt] : -1;
#pragma unroll
        for (int i = 0; i < 8; i++)
            if (kf[i] >= 0) atomicAdd(&ho[kf[i] + 1], 1);
    }
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

// 2. landmarks by (first keyframe, index), stable: one wave per chunk of 64 landmarks in index order counts its members per bucket
//    (k_st_lm_count), k_st_rec_scan turns the counts into prefixes over the chunks, k_st_lm_fill ranks.  Same scratch (rec_cnt,
//    column 0) and same shape as the record ranking below.
__global__ void __launch_bounds__(64) k_st_lm_count(Batch B, StBuild T, int max_chunks) {
    const int w = blockIdx.y, c = blockIdx.x;
    const WinDesc& d = B.desc[w];
    if (64 * c >= d.n_pt) return;
    const int lane = threadIdx.x, nk = d.n_kf, p = 64 * c + lane;
    const int key = (p < d.n_pt) ? T.st_key[d.pt0 + p] : -1;
    int* cnt = T.rec_cnt + 2 * ((size_t)d.kf0 * max_chunks + (size_t)c * nk);
    for (int wd = 0; wd < (nk + 63) / 64; wd++) {
        const u64_t kb = (key >= 0 && (key >> 6) == wd) ? (1ull << (key & 63)) : 0ull;
        u64_t U = wave_or64(kb);
        int co = 0;
        while (U) {
            const int bb = __builtin_ctzll(U);
            U &= U - 1;
            const int no = __popcll(__ballot((kb >> bb) & 1ull));
            if (lane == bb) co = no;
        }
        const int k = 64 * wd + lane;
        if (k < nk) { cnt[2 * k] = co; cnt[2 * k + 1] = 0; }
    }
}
__global__ void __launch_bounds__(64) k_st_lm_fill(Batch B, StBuild T, int max_chunks) {
    const int w = blockIdx.y, c = blockIdx.x;
    const WinDesc& d = B.desc[w];
    if (64 * c >= d.n_pt) return;
    const int lane = threadIdx.x, nk = d.n_kf, mw = d.mwords, p = 64 * c + lane;
    const int key = (p < d.n_pt) ? T.st_key[d.pt0 + p] : -1;
    const int* pre = T.rec_cnt + 2 * ((size_t)d.kf0 * max_chunks + (size_t)c * nk);
    const int* kseg = T.key_seg + d.kf0 + d.win;
    const u64_t lt = lanes_below();
    for (int wd = 0; wd < (nk + 63) / 64; wd++) {
        const u64_t kb = (key >= 0 && (key >> 6) == wd) ? (1ull << (key & 63)) : 0ull;
        u64_t U = wave_or64(kb);
        while (U) {
            const int bb = __builtin_ctzll(U);
            U &= U - 1;
            const int k = 64 * wd + bb;
            const bool has = (kb >> bb) & 1ull;
            const u64_t m = __ballot(has);
            if (has) {
                const int q = kseg[k] + pre[2 * k] + __popcll(m & lt);
                T.lm_order[d.pt0 + q] = p;
                T.ref_q[d.pt0 + q] = B.pt_ref[d.pt0 + p];
                for (int v = 0; v < mw; v++) T.mask_q[d.mask0 + (size_t)q * mw + v] = B.lmask[d.mask0 + (size_t)p * mw + v];
            }
        }
    }
}

// 3. observation records by (observing keyframe, lm_order) and landmark records by (reference keyframe, lm_order).  One wave per
//    CHUNK of 64 landmarks in lm_order (a lane per landmark: its mask, reference, CSR row and the keyframes of its first eight
//    observations are loaded once), three launches:
//      k_st_rec_count  per chunk and keyframe: how many of the chunk's landmarks the keyframe observes / is the reference of
//                      (one ballot per keyframe that occurs in the chunk)
//      k_st_rec_scan   per window and keyframe: exclusive prefix over the chunks
//      k_st_rec_fill   the same ballots again; record position = segment start + prefix of the chunk + rank inside the ballot
//    (Round 2 first had one wave per (window, KEYFRAME) scan all landmark masks -- n_kf x the mask reads -- and look the observation
//    up per incidence through the landmark's CSR row: 2 scattered lines per incidence instead of 2 per landmark, 9.7 ms per 4096 windows.)
//    cnt: [chunk][n_kf][2] ints per window at 2 * kf0 * max_chunks (max_chunks = 64-landmark blocks of the largest window of the batch)
struct RecLane {   // what a lane knows about its landmark
    bool valid;
    int q, p, ref, ob0, nob;
    int okf[8];
};
DEVI void rec_lane_load(const Batch& B, const StBuild& T, const WinDesc& d, int c, RecLane& L, bool full) {
    const int lane = threadIdx.x;
    L.q = 64 * c + lane;
    L.valid = L.q < d.n_pt;
    L.ref = (L.valid && d.variant == 2) ? T.ref_q[d.pt0 + L.q] : -1;
    L.p = 0; L.ob0 = 0; L.nob = 0;
    if (full && L.valid) {
        const int* ob = B.pt_obs_begin + d.pt0 + d.win;
        L.p = T.lm_order[d.pt0 + L.q];
        L.ob0 = ob[L.p];
        L.nob = ob[L.p + 1] - L.ob0;
#pragma unroll
        for (int i = 0; i < 8; i++) L.okf[i] = (i < L.nob) ? B.obs_kf[d.obs0 + L.ob0 + i] : -1;
    }
}
__global__ void __launch_bounds__(64) k_st_rec_count(Batch B, StBuild T, int max_chunks) {
    const int w = blockIdx.y, c = blockIdx.x;
    const WinDesc& d = B.desc[w];
    if (64 * c >= d.n_pt) return;
    const int lane = threadIdx.x, mw = d.mwords, nk = d.n_kf;
    RecLane L;
    rec_lane_load(B, T, d, c, L, false);
    const u64_t* M = T.mask_q + d.mask0 + (size_t)(L.valid ? L.q : 0) * mw;
    int* cnt = T.rec_cnt + 2 * ((size_t)d.kf0 * max_chunks + (size_t)c * nk);
    for (int wd = 0; wd < mw; wd++) {
        const u64_t m = L.valid ? M[wd] : 0ull;
        const u64_t rb = (L.ref >= 0 && (L.ref >> 6) == wd) ? (1ull << (L.ref & 63)) : 0ull;
        u64_t U = wave_or64(m | rb);
        int co = 0, cr = 0;   // lane l: the counts of keyframe 64 wd + l
        while (U) {
            const int bb = __builtin_ctzll(U);
            U &= U - 1;
            const int no = __popcll(__ballot((m >> bb) & 1ull)), nr = __popcll(__ballot((rb >> bb) & 1ull));
            if (lane == bb) { co = no; cr = nr; }
        }
        const int k = 64 * wd + lane;
        if (k < nk) { cnt[2 * k] = co; cnt[2 * k + 1] = cr; }
    }
}
__global__ void __launch_bounds__(256) k_st_rec_scan(Batch B, StBuild T, int max_chunks) {
    const int w = blockIdx.x;
    const WinDesc& d = B.desc[w];
    const int nk = d.n_kf, nch = (d.n_pt + 63) / 64;
    int* cnt = T.rec_cnt + 2 * (size_t)d.kf0 * max_chunks;
    for (int k2 = threadIdx.x; k2 < 2 * nk; k2 += 256) {   // (keyframe, obs / ref) columns: exclusive prefix over the chunks
        int run = 0;
        // eight chunks per trip, their counts requested together: the column is one dependent chain of ~80 loads otherwise (the
        // stores in between keep the compiler from hoisting them; 19 us per launch for a single C3 window)
        for (int c0 = 0; c0 < nch; c0 += 8) {
            int v[8];
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = (c0 + i < nch) ? cnt[(size_t)(c0 + i) * 2 * nk + k2] : 0;
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (c0 + i < nch) { cnt[(size_t)(c0 + i) * 2 * nk + k2] = run; run += v[i]; }
        }
    }
}
__global__ void __launch_bounds__(64) k_st_rec_fill(Batch B, StBuild T, int max_chunks) {
    const int w = blockIdx.y, c = blockIdx.x;
    const WinDesc& d = B.desc[w];
    if (64 * c >= d.n_pt) return;
    const int lane = threadIdx.x, mw = d.mwords, nk = d.n_kf;
    RecLane L;
    rec_lane_load(B, T, d, c, L, true);
    const u64_t* M = T.mask_q + d.mask0 + (size_t)(L.valid ? L.q : 0) * mw;
    const int* pre = T.rec_cnt + 2 * ((size_t)d.kf0 * max_chunks + (size_t)c * nk);
    const int* kseg = T.kf_seg + d.kf0 + d.win;
    const int* rseg = T.ref_seg + d.kf0 + d.win;
    const u64_t lt = lanes_below();
    int below = 0;   // observing keyframes of this landmark in the mask words already walked
    for (int wd = 0; wd < mw; wd++) {
        const u64_t m = L.valid ? M[wd] : 0ull;
        const u64_t rb = (L.ref >= 0 && (L.ref >> 6) == wd) ? (1ull << (L.ref & 63)) : 0ull;
        u64_t U = wave_or64(m | rb);
        while (U) {
            const int bb = __builtin_ctzll(U);
            U &= U - 1;
            const int k = 64 * wd + bb;
            const bool haso = (m >> bb) & 1ull, hasr = (rb >> bb) & 1ull;
            const u64_t mo = __ballot(haso), mr = __ballot(hasr);
            if (haso) {
                const int slot = kseg[k] + pre[2 * k] + __popcll(mo & lt);
                int o = -1;
#pragma unroll
                for (int i = 0; i < 8; i++) o = (L.okf[i] == k) ? L.ob0 + i : o;
                if (o < 0) { o = L.ob0 + 8; while (B.obs_kf[d.obs0 + o] != k) o++; }   // tracks longer than eight observations
                T.slot_perm[d.obs0 + o] = slot;
                T.slot_obs[d.obs0 + slot] = L.p;   // the landmark of the record in this slot
                T.slot_o[d.obs0 + slot] = o;
                T.slot_q[d.obs0 + slot] = L.q;
                T.slot_ref[d.obs0 + slot] = L.ref;
                for (int v = 0; v < mw; v++) T.slot_mask[(size_t)(d.obs0 + slot) * T.smw + v] = M[v];
                const int r = below + __popcll(m & ((1ull << bb) - 1ull));
                T.tslot[d.obs0 + L.ob0 + r] = slot;   // the landmark's slots in keyframe order (st_slot_of)
                if (r < 8) T.tsq[8 * (size_t)(d.pt0 + L.q) + r] = slot;
            }
            if (hasr) {
                const int r = rseg[k] + pre[2 * k + 1] + __popcll(mr & lt);
                T.pt_perm[d.pt0 + L.p] = r;
                T.pt_inv[d.pt0 + r] = L.p;
                T.rec_q[d.pt0 + r] = L.q;
            }
        }
        below += __popcll(m);
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
        const int p = (FILL && valid) ? T.slot_obs[d.obs0 + slot] : 0;
        const int r = (valid && idp) ? T.slot_ref[d.obs0 + slot] : -1;
        int4 sr_lo = make_int4(0, 0, 0, 0), sr_hi = make_int4(0, 0, 0, 0);
        if (FILL) slot_row_load(T, d, valid ? T.slot_q[d.obs0 + slot] : 0, valid, sr_lo, sr_hi);
        const u64_t lmw0 = valid ? T.slot_mask[(size_t)(d.obs0 + slot) * T.smw] : 0ull;   // word 0 of the landmark's mask
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
                    if (h0) items[c0[b] + __popcll(m0 & lt)] = make_int2(slot, slot_row_at(sr_lo, sr_hi, T, d, LM, ob, p, b, lmw0));
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
            const int p = (FILL && valid) ? T.pt_inv[d.pt0 + rec] : 0;
            const int q = valid ? T.rec_q[d.pt0 + rec] : 0;
            const u64_t* MQ = T.mask_q + d.mask0 + (size_t)q * mw;   // the landmark's mask, stored in lm_order
            int4 sr_lo = make_int4(0, 0, 0, 0), sr_hi = make_int4(0, 0, 0, 0);
            if (FILL) slot_row_load(T, d, q, valid, sr_lo, sr_hi);
            const u64_t lmw0 = valid ? MQ[0] : 0ull;
            for (int wd = a >> 6; wd < mw; wd++) {
                const u64_t rg = range(wd);
                if (!rg) continue;
                const u64_t Mr = valid ? (MQ[wd] & rg) : 0ull;
                u64_t U = wave_or64(Mr);
                while (U) {
                    const int bb = __builtin_ctzll(U);
                    U &= U - 1;
                    const int b = 64 * wd + bb;
                    const bool h1 = (Mr >> bb) & 1ull;
                    const u64_t m1 = __ballot(h1);
                    if (FILL && h1) items[c1[b] + __popcll(m1 & lt)] = make_int2(d.n_obs + rec, slot_row_at(sr_lo, sr_hi, T, d, LM, ob, p, b, lmw0));
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
// The same walk for windows of at most 64 keyframes (one mask word): the running counts of the pairs (a, b) live in LANE b of two
// registers instead of LDS -- the count of the partner in hand is a v_readlane, its update a conditional add in lane b -- so the
// inner loop has no barrier and no LDS round trip (a single C3 window: k_st_count 35 -> 24 us, k_st_fill 75 -> 58 us).
// Same ballots, same ranks: the item lists are identical to those of st_row_body.
template <bool FILL>
DEVI void st_row_body1(const Batch& B, const StBuild& T) {
    const int w = blockIdx.y, a = blockIdx.x;
    const WinDesc& d = B.desc[w];
    const int nf = d.n_free;
    if (a >= nf) return;
    const int lane = threadIdx.x;
    const bool idp = d.variant == 2;
    const int rowbase = a * nf - a * (a - 1) / 2 - a;   // pair index of (a, b) = rowbase + b
    int* ib = T.item_begin + d.pair0 + d.win;
    int* im = T.item_mid + d.pair0 + d.win;
    const bool mine = lane > a && lane < nf;             // lane b keeps the counts of pair (a, b)
    int r0 = (FILL && mine) ? ib[rowbase + lane] : 0;
    int r1 = (FILL && mine) ? im[rowbase + lane] : 0;
    const int* kseg = T.kf_seg + d.kf0 + d.win;
    const int* rseg = T.ref_seg + d.kf0 + d.win;
    const int* ob = B.pt_obs_begin + d.pt0 + d.win;
    const u64_t* LM = B.lmask + d.mask0;
    const u64_t lt = lanes_below();
    int2* items = reinterpret_cast<int2*>(T.items) + d.item0;
    u64_t rg = ~0ull;                                    // bits that name a keyframe b with a < b < nf
    rg &= (a + 1 >= 64) ? 0ull : (~0ull << (a + 1));
    if (nf < 64) rg &= ~0ull >> (64 - nf);
    // 1. observation records of a
    for (int c = kseg[a]; c < kseg[a + 1]; c += 64) {
        const int slot = c + lane;
        const bool valid = slot < kseg[a + 1];
        const int p = (FILL && valid) ? T.slot_obs[d.obs0 + slot] : 0;
        const int r = (valid && idp) ? T.slot_ref[d.obs0 + slot] : -1;
        int4 sr_lo = make_int4(0, 0, 0, 0), sr_hi = make_int4(0, 0, 0, 0);
        if (FILL) slot_row_load(T, d, valid ? T.slot_q[d.obs0 + slot] : 0, valid, sr_lo, sr_hi);
        const u64_t lmw0 = valid ? T.slot_mask[(size_t)(d.obs0 + slot) * T.smw] : 0ull;
        const u64_t Mr = lmw0 & rg;
        const u64_t rb = (r > a && r < nf) ? (1ull << r) : 0ull;
        u64_t U = wave_or64(Mr | rb);
        U = ((u64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(U >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)U);   // (uniform: scalar loop)
        while (U) {
            const int b = __builtin_ctzll(U);
            U &= U - 1;
            const bool h0 = (Mr >> b) & 1ull, h1 = (rb >> b) & 1ull;
            const u64_t m0 = __ballot(h0), m1 = __ballot(h1);
            if (FILL) {
                const int base0 = __builtin_amdgcn_readlane(r0, b), base1 = __builtin_amdgcn_readlane(r1, b);
                if (h0) items[base0 + __popcll(m0 & lt)] = make_int2(slot, slot_row_at(sr_lo, sr_hi, T, d, LM, ob, p, b, lmw0));
                if (h1) items[base1 + __popcll(m1 & lt)] = make_int2(slot, d.n_obs + T.pt_perm[d.pt0 + p]);
            }
            if (lane == b) { r0 += __popcll(m0); r1 += __popcll(m1); }
        }
    }
    // 2. landmark records a is the reference keyframe of: reference items only
    if (idp)
        for (int c = rseg[a]; c < rseg[a + 1]; c += 64) {
            const int rec = c + lane;
            const bool valid = rec < rseg[a + 1];
            const int p = (FILL && valid) ? T.pt_inv[d.pt0 + rec] : 0;
            const int q = valid ? T.rec_q[d.pt0 + rec] : 0;
            int4 sr_lo = make_int4(0, 0, 0, 0), sr_hi = make_int4(0, 0, 0, 0);
            if (FILL) slot_row_load(T, d, q, valid, sr_lo, sr_hi);
            const u64_t lmw0 = valid ? T.mask_q[d.mask0 + (size_t)q] : 0ull;   // the landmark's mask, stored in lm_order (one word)
            const u64_t Mr = lmw0 & rg;
            u64_t U = wave_or64(Mr);
            U = ((u64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(U >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)U);
            while (U) {
                const int b = __builtin_ctzll(U);
                U &= U - 1;
                const bool h1 = (Mr >> b) & 1ull;
                const u64_t m1 = __ballot(h1);
                if (FILL) {
                    const int base1 = __builtin_amdgcn_readlane(r1, b);
                    if (h1) items[base1 + __popcll(m1 & lt)] = make_int2(d.n_obs + rec, slot_row_at(sr_lo, sr_hi, T, d, LM, ob, p, b, lmw0));
                }
                if (lane == b) r1 += __popcll(m1);
            }
        }
    if (!FILL) {
        if (mine) { ib[rowbase + lane] = r0 + r1; im[rowbase + lane] = r0; }
        else if (lane == a) { ib[rowbase + lane] = 0; im[rowbase + lane] = 0; }   // the diagonal pair has no list
    }
}
__global__ void __launch_bounds__(64) k_st_count(Batch B, StBuild T, int max_free) {
    extern __shared__ int shc[];
    if (B.desc[blockIdx.y].mwords == 1 && !T.row_lds) st_row_body1<false>(B, T);
    else st_row_body<false>(B, T, shc, shc + max_free);
}
__global__ void __launch_bounds__(64) k_st_fill(Batch B, StBuild T, int max_free) {
    extern __shared__ int shc[];
    if (B.desc[blockIdx.y].mwords == 1 && !T.row_lds) st_row_body1<true>(B, T);
    else st_row_body<true>(B, T, shc, shc + max_free);
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
