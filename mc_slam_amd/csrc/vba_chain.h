// vba_chain.h -- the chain columns of the reduced-system factorisation in ONE launch.
//
// Under the V/Bias-first elimination order (vba_host_structure.h) the leading block columns of the tile L D L^T hold the V/Bias
// blocks of the IMU chain: block-tridiagonal among themselves, so that row J of L has no tile left of (J, J-1) for J < nc ("chain
// columns", 13 of the 23 block columns of a 50-keyframe window).  A chain column is a short dependent step -- its diagonal tile
// needs one product, its panel tiles one product each plus the solve against the diagonal tile -- and the per-column kernels paid a
// dependent launch (few windows: 8 us each, 13 of 23 per factorisation) or two half-empty launches (many windows) for each of them.
// Here one workgroup per window walks the chain:
//   wave 0   eliminates [C_JJ | S_{J+1,J}] with the DPP elimination of k_chol_step4 (diagonal rows in lanes 0..31, the rows of the
//            sub-diagonal tile riding along in lanes 32..63, the right-hand side as one more column): L_JJ, D_J, z_J, L_{J+1,J}
//   wave 1   eliminates [C_JJ | I] on its own SIMD at the same time: the identity comes out as W_J = L_JJ^-T D_J^-1 (as in ll_diag2)
//   waves 2+ the other tiles (I,J) of the column (the PR rows the chain fills): C_IJ = S_IJ - L_{I,J-1} D_{J-1} L_{J,J-1}^T while the
//            elimination runs, L_IJ = C_IJ W_J behind it (two MFMA products per tile, transposed accumulators as in k_chol_panel_ll)
//   then waves 0 / 1 form C_{J+1,J+1} = S_{J+1,J+1} - L_{J+1,J} D_J L_{J+1,J}^T (MFMA, through LDS) for the next column.
// Two barriers per column; everything that passes from wave to wave goes through LDS.  The factor leaves in the layout of the regime
// (PACKED: tile by tile in MFMA operand order for the left-looking kernels; otherwise row-major for k_chol_step4 / k_trsv_p).
// Few-window regime: the tiles behind the chain that collect its updates are brought up to date by ONE more launch
// (k_chol_chain_upd, accumulating over the chain columns in MFMA registers), then the per-column steps start at column nc.
// Restates LinearSolverEigen::solve (Thirdparty/g2o/g2o/solvers/linear_solver_eigen.h:94-124) like the kernels it replaces; same
// elimination order, sums in another fixed order (results agree with the per-column kernels to rounding).
#pragma once
#include "vba_kernels.h"

#define CHAIN_MAXT 2   // tiles per panel wave whose accumulators wait in registers for W_J

// a 32x32 tile held as rows in LDS (pitch 34) -> the factor
template <bool PACKED>
DEVI void chain_store_tile(const double* rows, double* Lf, const WinDesc& d, int I, int J, int lane) {
    const int l15 = lane & 15, l4 = lane >> 4;
    if constexpr (PACKED) {
        double2* dst = reinterpret_cast<double2*>(Lf + ll_tile(d, I, J)) + lane;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int row = 16 * (q >> 2) + l15, c0 = 8 * (q & 3) + l4;
            dst[64 * q] = make_double2(rows[row * 34 + c0], rows[row * 34 + c0 + 4]);
        }
    } else {
        const int n = d.nS;
#pragma unroll
        for (int it = 0; it < 8; it++) {   // sixteen lanes per row, four rows per trip
            const int row = 4 * it + l4, c = 2 * l15;
            *reinterpret_cast<double2*>(Lf + ((size_t)I * 32 + row) * n + (size_t)J * 32 + c) = make_double2(rows[row * 34 + c], rows[row * 34 + c + 1]);
        }
    }
}
// tile (I,k) of the factor as an MFMA operand: piece p of a lane = elements (16 (p >> 2) + l15, 8 (p & 3) + l4) and (.., + 4)
template <bool PACKED>
DEVI void chain_load_op(const double* Lf, const WinDesc& d, int I, int k, int lane, double2 (&x)[8]) {
    if constexpr (PACKED) {
        const double2* t = reinterpret_cast<const double2*>(Lf + ll_tile(d, I, k)) + lane;
#pragma unroll
        for (int q = 0; q < 8; q++) x[q] = t[64 * q];
    } else {
        const int l15 = lane & 15, l4 = lane >> 4, n = d.nS;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const double* e = Lf + ((size_t)I * 32 + 16 * (q >> 2) + l15) * n + (size_t)k * 32 + 8 * (q & 3) + l4;
            x[q] = make_double2(e[0], e[4]);
        }
    }
}
// L_IJ^T = W_J^T C_IJ^T with W_J^T read from LDS (WT[r][c] = W_J[r][c]); acc = the transposed accumulators of ll_panel_init.
// The tile goes to the factor and comes back in xo as the operand pieces of the next column's product.
template <bool PACKED>
DEVI void chain_panel_finish(const double* WT, const d4_t (&acc)[2][2], double* Lf, const WinDesc& d, int I, int J, int lane, double2 (&xo)[8]) {
    const int l15 = lane & 15, l4 = lane >> 4;
    double2* out = reinterpret_cast<double2*>(Lf + ll_tile(d, I, J)) + lane;
#pragma unroll
    for (int tk = 0; tk < 2; tk++)
#pragma unroll
        for (int ti = 0; ti < 2; ti++) {
            d4_t o = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < 8; ks++) {
                const double av = WT[(4 * ks + l4) * 34 + 16 * tk + l15];
                const double bv = acc[ks >> 2][ti][ks & 3];
                o = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, o, 0, 0, 0);
            }
            // o[i] = L[16 ti + l15][16 tk + 4 i + l4]: the result registers are the operand pieces of the tile (ll_pk order)
            xo[ti * 4 + 2 * tk] = make_double2(o[0], o[1]);
            xo[ti * 4 + 2 * tk + 1] = make_double2(o[2], o[3]);
            if constexpr (PACKED) {
                out[64 * (ti * 4 + 2 * tk)] = make_double2(o[0], o[1]);
                out[64 * (ti * 4 + 2 * tk + 1)] = make_double2(o[2], o[3]);
            } else {
                double* e = Lf + ((size_t)I * 32 + 16 * ti + l15) * d.nS + (size_t)J * 32 + 16 * tk + l4;
#pragma unroll
                for (int i = 0; i < 4; i++) e[4 * i] = o[i];
            }
        }
}

// ------------------------------------------------------------------------------------------------
// MANY windows (left-looking regime: tile-packed factor, S pristine), where the sum of the waves' lifetimes counts: two launches.
//   k_chol_chain_diag   one WAVE per window walks the chain's own tiles: per column the elimination of [C_JJ | I] (ll_diag2: the
//                       identity comes out as W_J), then L_{J+1,J} = S_{J+1,J} W_J and C_{J+1,J+1} = S_{J+1,J+1} - L_{J+1,J} D_J
//                       L_{J+1,J}^T as MFMA products whose operands never leave the wave (result registers of the first are the
//                       operand pieces of the second), the right-hand side carried from column to column.  W_J of every column
//                       stays in memory for the second launch.
//   k_chol_chain_panel  one wave per (window, tile row I >= nc) walks the columns: C_IJ = S_IJ - L_{I,J-1} D_{J-1} L_{J,J-1}^T,
//                       L_IJ = C_IJ W_J -- k_chol_panel_ll for all chain columns, the wave's tile staying in registers as the
//                       operand of its next column.
// Against one diagonal + one panel launch per column: 2 instead of 26 launches per factorisation for a 50-keyframe window, and no
// launch whose 4096 waves live for a single short column.
// ------------------------------------------------------------------------------------------------
DEVI double* chain_w_tile(const Batch& B, const WinDesc& d, int J) {   // W_J of window d.win (behind the per-window tiles of the per-column kernels)
    return B.winv + 1024 * ((size_t)B.w_total + (size_t)d.win * B.w_stride + J);
}
__global__ void __launch_bounds__(64, 2) k_chol_chain_diag(Batch B) {
    __shared__ double CT[32 * 34];
    __shared__ double WT[32 * 34];
    __shared__ double dz[64];           // D_J, z_J
    const int w = blockIdx.x;
    if (w >= B.n_win) return;
    const WinDesc& d = B.desc[w];
    WinCtrl& c = B.ctrl[w];
    if (!win_on(d, c)) return;
    const int nc = d.nc;
    if (nc <= 0) return;
    const int n = d.nS;
    const int lane = threadIdx.x, r = lane & 31, hi = lane >> 5, l15 = lane & 15, l4 = lane >> 4;
    const double* S = B.S + d.S0;
    double* Lf = B.Lf + d.S0;
    const double* vec = B.vec + d.vec0;
    double* yv = B.yv + d.vec0;
    double* dvec = B.dvec + d.vec0;
    const int4* ct = reinterpret_cast<const int4*>(B.tl_ct) + d.ct0;
    for (int q = lane; q < 1024; q += 64) CT[(q >> 5) * 34 + (q & 31)] = S[(size_t)(q >> 5) * n + (q & 31)];
    double rhs = vec[r];                // the right-hand side rows of the current diagonal tile (lanes 0..31)
    bool bad = false;
    lds_barrier();
    for (int J = 0; J < nc; J++) {
        const size_t dk = (size_t)J * 32;
        const bool ride = ct[J].z != 0, upd = J + 1 < nc;
        // the tiles of S the products below start from, requested in front of the elimination
        d4_t accp[2][2], accd[2][2];
        if (ride) ll_panel_init(S, n, J + 1, J, l15, l4, accp);
        if (upd) {
#pragma unroll
            for (int ti = 0; ti < 2; ti++)
#pragma unroll
                for (int tj = 0; tj < 2; tj++) {
                    const double* C = S + ((size_t)(J + 1) * 32 + 16 * ti) * n + (size_t)(J + 1) * 32 + 16 * tj;
#pragma unroll
                    for (int i = 0; i < 4; i++) accd[ti][tj][i] = C[(size_t)(l4 + 4 * i) * n + l15];
                }
        }
        const double rhs_next = upd ? vec[dk + 32 + r] : 0.0;
        double t[32];
#pragma unroll
        for (int q = 0; q < 32; q++) {
            const double cv = CT[r * 34 + q];
            t[q] = hi ? ((q == r) ? 1.0 : 0.0) : cv;
        }
        double rr = hi ? 0.0 : rhs, dout = 1.0, zout = 0.0;
        lds_barrier();                  // every lane has its row: CT is free
        elim_tile<0>(t, rr, dout, zout, 4 * l15, 4 * (16 + l15), nullptr);
        bad = bad || (!hi && (dout == 0.0 || !isfinite(dout)));
        if (!hi) {
            dvec[dk + r] = dout;
            yv[dk + r] = zout;
            dz[r] = dout;
            dz[32 + r] = zout;
        }
        double* rowp = (hi ? WT : CT) + r * 34;     // CT[r][c] = L_JJ[r][c] (D on the diagonal, zeros above), WT[r][c] = W_J[r][c]
#pragma unroll
        for (int q = 0; q < 32; q++) rowp[q] = (hi || q < r) ? t[q] : ((q == r) ? dout : 0.0);
        lds_barrier();
        {   // packed order (ll_pk): L_JJ into the factor, W_J^T for the second launch
            double2* Ljj = reinterpret_cast<double2*>(Lf + ll_tile(d, J, J)) + lane;
            double2* Wd = reinterpret_cast<double2*>(chain_w_tile(B, d, J)) + lane;
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int row = 16 * (q >> 2) + l15, c0 = 8 * (q & 3) + l4;
                Ljj[64 * q] = make_double2(CT[row * 34 + c0], CT[row * 34 + c0 + 4]);
                Wd[64 * q] = make_double2(WT[c0 * 34 + row], WT[(c0 + 4) * 34 + row]);
            }
        }
        double sdot = 0.0;
        if (ride) {
            double2 x[8];
            chain_panel_finish<true>(WT, accp, Lf, d, J + 1, J, lane, x);   // L_{J+1,J} = S_{J+1,J} W_J: into the factor, and x = its operand pieces
            double dv[8], zk[8];
#pragma unroll
            for (int ks = 0; ks < 8; ks++) { dv[ks] = dz[4 * ks + l4]; zk[ks] = dz[32 + 4 * ks + l4]; }
            double p0 = 0.0, p1 = 0.0;  // this lane's share of L_{J+1,J} z_J for rows l15 and 16 + l15
#pragma unroll
            for (int ks = 0; ks < 8; ks++) {
                const double2 a0 = x[ks >> 1], a1 = x[4 + (ks >> 1)];
                p0 += ((ks & 1) ? a0.y : a0.x) * zk[ks];
                p1 += ((ks & 1) ? a1.y : a1.x) * zk[ks];
            }
#pragma unroll
            for (int ks = 0; ks < 8; ks++)
#pragma unroll
                for (int ti = 0; ti < 2; ti++)
#pragma unroll
                    for (int tj = 0; tj <= ti; tj++) {      // (what lies above the diagonal of C_{J+1,J+1} is never used)
                        const double2 pi = x[ti * 4 + (ks >> 1)], pj = x[tj * 4 + (ks >> 1)];
                        const double av = -((ks & 1) ? pi.y : pi.x);
                        const double bv = ((ks & 1) ? pj.y : pj.x) * dv[ks];
                        accd[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, accd[ti][tj], 0, 0, 0);
                    }
            p0 += __shfl_xor(p0, 16, 64); p0 += __shfl_xor(p0, 32, 64);
            p1 += __shfl_xor(p1, 16, 64); p1 += __shfl_xor(p1, 32, 64);
            sdot = (lane & 16) ? p1 : p0;           // lane r < 32 holds the sum of row r
        }
        if (upd) {
            lds_barrier();              // the packing above has read CT
#pragma unroll
            for (int ti = 0; ti < 2; ti++)
#pragma unroll
                for (int tj = 0; tj < 2; tj++)
#pragma unroll
                    for (int i = 0; i < 4; i++) CT[(16 * ti + l4 + 4 * i) * 34 + 16 * tj + l15] = accd[ti][tj][i];
            rhs = rhs_next - __shfl(sdot, r, 64);   // b_{J+1} - L_{J+1,J} z_J
            lds_barrier();
        }
    }
    if (__ballot(bad) != 0ull && lane == 0) c.chol_fail = 1;
}

__global__ void __launch_bounds__(64) k_chol_chain_panel(Batch B, int per_win) {
    int w, qrow;
    if (!schur_map(B, per_win, w, qrow)) return;      // the waves of one window sit on one XCD
    const WinDesc& d = B.desc[w];
    if (!win_on(d, B.ctrl[w])) return;
    const int nc = d.nc;
    if (nc <= 0) return;
    const int I = nc + qrow;
    if (I >= d.nb || qrow >= 64) return;
    const int n = d.nS;
    const int lane = threadIdx.x, l15 = lane & 15, l4 = lane >> 4;
    const double* S = B.S + d.S0;
    double* Lf = B.Lf + d.S0;
    const int4* ct = reinterpret_cast<const int4*>(B.tl_ct) + d.ct0;
    double2 xl[8];
    bool had = false, ride_prev = false;
    for (int J = 0; J < nc; J++) {
        const int4 e = ct[J];
        const unsigned long long mask = ((unsigned long long)(unsigned)e.y << 32) | (unsigned)e.x;
        const bool pres = (mask >> qrow) & 1ull;
        if (pres) {
            d4_t acc[2][2];
            ll_panel_init(S, n, I, J, l15, l4, acc);
            const double2* tw = reinterpret_cast<const double2*>(chain_w_tile(B, d, J)) + lane;
            double2 xw[8];
#pragma unroll
            for (int q = 0; q < 8; q++) xw[q] = tw[64 * q];
            if (had && ride_prev) {     // tiles (I, J-1) and (J, J-1) are both in the factor
                const double2* tj_ = reinterpret_cast<const double2*>(Lf + ll_tile(d, J, J - 1)) + lane;
                const double* sd = B.dvec + d.vec0 + (size_t)(J - 1) * 32 + l4;
                double2 xj[8];
                double dv[8];
#pragma unroll
                for (int q = 0; q < 8; q++) { xj[q] = tj_[64 * q]; dv[q] = sd[4 * q]; }
                double av[2][8];
#pragma unroll
                for (int tj = 0; tj < 2; tj++)
#pragma unroll
                    for (int ks = 0; ks < 8; ks++) {
                        const double2 pj = xj[tj * 4 + (ks >> 1)];
                        av[tj][ks] = ((ks & 1) ? pj.y : pj.x) * dv[ks];
                    }
                ll_panel_mfma(av, xl, acc);
            }
            ll_panel_finish_keep(xw, acc, reinterpret_cast<double2*>(Lf + ll_tile(d, I, J)) + lane, xl);
        }
        had = pres;
        ride_prev = e.z != 0;
    }
}

// ------------------------------------------------------------------------------------------------
// The chain for FEW windows (row-major factor, S updated in place behind the chain), where the latency of one window is what counts:
// the MFMA products of a window's chain columns (two per panel tile, ~180 for a 50-keyframe window) on ONE compute unit take longer
// than the launches they replace and slow the eliminating waves down (in-kernel stamps: 25 k cycles per column).  So here ONE
// WORKGROUP PER TILE ROW I >= nc walks the chain on a compute unit of its own, redoing the chain's diagonal work (as every tile-pair
// workgroup of k_chol_step4 redoes the diagonal tile of its step) and carrying ITS rows through it:
//   wave 0  eliminates [C_JJ | S_{J+1,J}]  -> L_{J+1,J}, D_J (workgroup 0 also writes L_JJ, z_J, D_J to memory)
//   wave 1  eliminates [C_JJ | C_IJ] on its own SIMD at the same time -> L_IJ: the rows of the workgroup's tile ride along, no W_J
//   then the four waves form C_{J+1,J+1} = S_{J+1,J+1} - L_{J+1,J} D_J L_{J+1,J}^T and C_{I,J+1} = S_{I,J+1} - L_IJ D_J L_{J+1,J}^T,
//   seven 16x16 quadrant products through LDS.
// No workgroup waits for another; two LDS-only barriers per column.
// ------------------------------------------------------------------------------------------------
#define CHAIN_MAX_NC 256
// a tile held as rows with permuted columns (ElimStep, WRITE_X = 2) -> the row-major factor
DEVI void chain_store_rows_perm(const double* rows, double* Lf, int n, int I, int J, int lane) {
    const int l15 = lane & 15, l4 = lane >> 4;
    const int c = 2 * l15, p0 = (c & 3) * 8 + (c >> 2), p1 = ((c + 1) & 3) * 8 + ((c + 1) >> 2);
#pragma unroll
    for (int it = 0; it < 8; it++) {   // sixteen lanes per row, four rows per trip
        const int row = 4 * it + l4;
        *reinterpret_cast<double2*>(Lf + ((size_t)I * 32 + row) * n + (size_t)J * 32 + c) = make_double2(rows[row * 34 + p0], rows[row * 34 + p1]);
    }
}
__global__ void __launch_bounds__(512) k_chol_chain_rows(Batch B) {
    // one set per chain (the two-sided order, WinDesc::nc_split: the workgroup's two halves walk the two chains side by side)
    __shared__ double CTs[2][32 * 34];      // C_JJ rows
    __shared__ double CIs[2][32 * 34];      // C_IJ rows (this workgroup's tile row)
    __shared__ double XLs[2][2 * 32 * 34];  // rows of L_{J+1,J}, behind them the same rows times D_J (permuted columns)
    __shared__ double XIs[2][2 * 32 * 34];  // rows of L_IJ (and times D_J: not used)
    __shared__ double XD[2 * 32 * 34];      // where the diagonal lanes' per-pivot stores go (never read)
    __shared__ double rcars[2][32];         // the rhs of the current diagonal tile's rows (wave 0 -> wave 1)
    __shared__ double rpart[32];            // the second chain's share of b_I - sum_J L_IJ z_J
    __shared__ short lride[CHAIN_MAX_NC], lpres[CHAIN_MAX_NC];
    const int w = blockIdx.y;
    const WinDesc& d = B.desc[w];
    WinCtrl& c = B.ctrl[w];
    if (!win_on(d, c)) return;
    const int nc = d.nc;
    if (nc <= 0) return;
    const int qrow = blockIdx.x, I = nc + qrow;
    if (I >= d.nb && qrow > 0) return;
    const bool owner = qrow == 0;                   // workgroup 0 writes the chain's own tiles (and carries row nc, if the chain fills it)
    const int n = d.nS;
    // 512 threads: hardware waves 0..3 are the four eliminating waves (chain 0: waves 0 and 1, chain 1: waves 0 and 1), one per SIMD;
    // hardware waves 4..7 the waves 2 and 3 of the two chains
    const int hw = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (wave-uniform: roles and LDS bases in scalar registers)
    const bool two = blockDim.x == 512;
    const int half = two ? ((hw >> 1) & 1) : 0, wave = two ? ((hw & 1) | ((hw >> 2) << 1)) : hw;
    const int lane = threadIdx.x & 63, r = lane & 31, hi = lane >> 5, l15 = lane & 15, l4 = lane >> 4, tid = 64 * wave + lane;
    const int split = (two && d.nc_split > 0) ? d.nc_split : (two ? nc : 0);   // (256 threads: one walk over all chain columns)
    const int J0 = half ? split : 0, J1 = (two && !half) ? split : nc;
    const int rounds = two ? max(split, nc - split) : nc;
    double* CT = CTs[half];
    double* CI = CIs[half];
    double* XL = XLs[half];
    double* XI = XIs[half];
    double* rcar = rcars[half];
    double* S = B.S + d.S0;
    double* Lf = B.Lf + d.S0;
    const double* vec = B.vec + d.vec0;
    double* yv = B.yv + d.vec0;
    double* dvec = B.dvec + d.vec0;
#ifdef VBA_STAMPS
#define RSTAMP(i) { if (w == 0 && qrow == 1 && lane == 0 && wave < 2 && half == 0 && J < 20) B.dbg[64 + (wave ? 192 : 0) + 8 * J + (i)] = (double)__builtin_amdgcn_s_memtime(); }
    if (w == 0 && qrow == 1 && threadIdx.x == 0) B.dbg[60] = (double)__builtin_amdgcn_s_memtime();
#else
#define RSTAMP(i)
#endif
    // the first tiles are requested before anything is known about the row (a row the chain never fills costs one wasted fetch)
    const bool live = J0 < J1;
    double c0[4], ci0[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int q = tid + 256 * u;
        c0[u] = live ? S[((size_t)J0 * 32 + (q >> 5)) * n + (size_t)J0 * 32 + (q & 31)] : 0.0;
        ci0[u] = (live && I < d.nb) ? S[((size_t)I * 32 + (q >> 5)) * n + (size_t)J0 * 32 + (q & 31)] : 0.0;
    }
    int any = 0;
    {
        const int4* ct = reinterpret_cast<const int4*>(B.tl_ct) + d.ct0;
        for (int J = threadIdx.x; J < nc; J += blockDim.x) {
            const int4 e = ct[J];
            const unsigned long long mask = ((unsigned long long)(unsigned)e.y << 32) | (unsigned)e.x;
            const int pres = (qrow < 64 && I < d.nb) ? (int)((mask >> qrow) & 1ull) : 0;
            lride[J] = (short)e.z;
            lpres[J] = (short)pres;
            any |= pres;
        }
    }
    any = __syncthreads_or(any);
    if (!any && !owner) return;                     // the chain never fills this tile row
    if (live) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int q = tid + 256 * u;
            CT[(q >> 5) * 34 + (q & 31)] = c0[u];
            CI[(q >> 5) * 34 + (q & 31)] = lpres[J0] ? ci0[u] : 0.0;
        }
    }
    double t[32], rr = 0.0, rr_carry = 0.0;
    // wave 1: the right-hand side rows of tile I ride through every column (b_I -= L_IJ z_J); the second chain collects its share from 0
    double rr_I = (wave == 1 && hi && I < d.nb && !half) ? vec[(size_t)I * 32 + r] : 0.0;
    if (wave == 0 && !hi && live) rcar[r] = vec[(size_t)J0 * 32 + r];
    // wave 0: the rows of tile (J+1, J) and the rhs rows of tile J+1 are requested a phase early (the chain never modifies S)
    const double* ride_row = S + ((size_t)32 * (J0 + 1) + r) * n + (size_t)32 * J0;      // row r of tile (J0+1, J0); + 32 n + 32 per column
    const double* ride_rhs = vec + (size_t)32 * (J0 + 1) + r;
    auto prefetch_ride = [&](int J) {
        const bool rd = hi && lride[J];
        const double4* row = reinterpret_cast<const double4*>(ride_row);
#pragma unroll
        for (int q = 0; q < 8; q++) {
            double4 v = make_double4(0.0, 0.0, 0.0, 0.0);
            if (rd) v = row[q];
            t[4 * q] = v.x; t[4 * q + 1] = v.y; t[4 * q + 2] = v.z; t[4 * q + 3] = v.w;
        }
        rr = (hi && J + 1 < J1) ? *ride_rhs : 0.0;
        ride_row += (size_t)32 * n + 32;
        ride_rhs += 32;
    };
    if (wave == 0 && live) prefetch_ride(J0);
    // the quadrant products of phase F: wave 0: diagonal (0,0), (1,0); wave 1: diagonal (1,1), row (0,0); wave 2: row (0,1), (1,0);
    // wave 3: row (1,1).  Per wave and product: source quadrant of S (a pointer that moves one tile per column), operand rows, target.
    bool q_on[2], q_row[2];
    const double* q_src[2];
    int q_a[2], q_b[2], q_dst[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int id0 = (wave == 0) ? (u ? 2 : 0) : (wave == 1) ? (u ? 4 : 3) : (wave == 2) ? (u ? 6 : 5) : (u ? -1 : 7);
        q_on[u] = id0 >= 0;
        const int id = id0 < 0 ? 0 : id0;
        q_row[u] = id >= 4;
        const int ti = q_row[u] ? ((id - 4) >> 1) : (id >> 1), tj = q_row[u] ? ((id - 4) & 1) : (id & 1);
        q_src[u] = S + ((size_t)(q_row[u] ? I : J0 + 1) * 32 + 16 * ti + l4) * n + (size_t)32 * (J0 + 1) + 16 * tj + l15;   // tile (J0+1,J0+1) / (I,J0+1)
        q_a[u] = (16 * ti + l15) * 34 + l4 * 8;
        q_b[u] = (16 * tj + l15) * 34 + l4 * 8 + ELIM_U_OFF;
        q_dst[u] = (16 * ti + l4) * 34 + 16 * tj + l15;
    }
    __syncthreads();
#ifdef VBA_STAMPS
    if (w == 0 && qrow == 1 && threadIdx.x == 0) B.dbg[61] = (double)__builtin_amdgcn_s_memtime();
#endif
    // the quadrants of S a wave updates in phase F are requested a column ahead (at the end of the previous column's phase F)
    d4_t ca[2];
    auto fetch_quadrants = [&](int J) {   // for the products of column J: tiles (J+1, J+1) and (I, J+1)
        const bool up = J + 1 < J1, pn = up && lpres[J + 1] != 0;
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const bool on = q_on[u] && (q_row[u] ? pn : up);
#pragma unroll
            for (int i = 0; i < 4; i++) ca[u][i] = on ? q_src[u][(size_t)(4 * i) * n] : 0.0;
            q_src[u] += q_row[u] ? 32 : (size_t)32 * n + 32;
        }
    };
    if (live) fetch_quadrants(J0);
    for (int it = 0; it < rounds; it++) {
        const int J = J0 + it;
        const bool act = J < J1;                    // (the shorter chain keeps the barriers of the longer one company)
        const bool ride = act && lride[J] != 0, pres = act && lpres[J] != 0;
        const bool upd = J + 1 < J1, presn = upd && lpres[J + 1] != 0;
        const size_t dk = (size_t)J * 32;
        // ---------------------------------------------------------------- phase E: the two eliminations
        RSTAMP(0)
        if (act && (wave == 0 || (wave == 1 && pres))) {
            if (!hi) {
#pragma unroll
                for (int q = 0; q < 32; q++) t[q] = CT[r * 34 + q];
                rr = rcar[r];
            } else if (wave == 1) {
#pragma unroll
                for (int q = 0; q < 32; q++) t[q] = CI[r * 34 + q];
                rr = rr_I;
            }
            double dout = 1.0, zout = 0.0;
            RSTAMP(1)
            elim_tile<2>(t, rr, dout, zout, 4 * l15, 4 * (16 + l15), (hi ? (wave ? XI : XL) : XD) + r * 34);
            RSTAMP(2)
            if (wave == 0) {
                if (owner) {
                    if (!hi) {                      // the diagonal tile of the factor: unit L below the diagonal, D on it, zeros above
                        double4* lrow = reinterpret_cast<double4*>(Lf + (dk + r) * n + dk);
#pragma unroll
                        for (int q = 0; q < 8; q++) {
                            double4 v;
                            v.x = (4 * q == r) ? dout : ((4 * q < r) ? t[4 * q] : 0.0);
                            v.y = (4 * q + 1 == r) ? dout : ((4 * q + 1 < r) ? t[4 * q + 1] : 0.0);
                            v.z = (4 * q + 2 == r) ? dout : ((4 * q + 2 < r) ? t[4 * q + 2] : 0.0);
                            v.w = (4 * q + 3 == r) ? dout : ((4 * q + 3 < r) ? t[4 * q + 3] : 0.0);
                            lrow[q] = v;
                        }
                        yv[dk + r] = zout;
                        dvec[dk + r] = dout;
                    }
                    const bool badl = !hi && (dout == 0.0 || !isfinite(dout));
                    if (__ballot(badl) != 0ull && lane == 0) c.chol_fail = 1;
                }
                rr_carry = __shfl(rr, 32 + r, 64);
            } else if (hi) rr_I = rr;
        }
        RSTAMP(3)
        lds_barrier();   // A: L_{J+1,J} (and times D_J) and L_IJ are in LDS
        RSTAMP(4)
        // ---------------------------------------------------------------- phase F: the tiles of the next column
        if (act) {
        if (wave == 0 && !hi) rcar[r] = rr_carry;   // (wave 1 read the old values in front of its elimination)
        {
            double2 xa[2][4], xb[2][4];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const double* XA = (q_row[u] ? XI : XL) + q_a[u];
                const double* XB = XL + q_b[u];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    xa[u][q] = *reinterpret_cast<const double2*>(XA + 2 * q);
                    xb[u][q] = *reinterpret_cast<const double2*>(XB + 2 * q);
                }
            }
#ifdef VBA_STAMPS
#define FSTAMP(i) { if (w == 0 && qrow == 1 && lane == 0 && wave == 1 && half == 0 && J < 15) B.dbg[448 + 4 * J + (i)] = (double)__builtin_amdgcn_s_memtime(); }
            { double sink = xa[0][0].x + xa[1][3].y + xb[0][0].x + xb[1][3].y; asm volatile("" :: "v"(sink)); }
            FSTAMP(0)
            { double sink = ca[0][0] + ca[1][3]; asm volatile("" :: "v"(sink)); }
            FSTAMP(1)
#else
#define FSTAMP(i)
#endif
            // A chain of dependent FP64 MFMAs runs at a third of the issue rate (~180 cycles per instruction measured with one
            // wave per SIMD): both products are cut into four independent accumulators of two k-steps each, all eight chains in
            // one basic block (products that do not apply are formed from whatever the LDS holds and not used).
            d4_t pa[2][4];
#pragma unroll
            for (int u = 0; u < 2; u++)
#pragma unroll
                for (int g = 0; g < 4; g++) pa[u][g] = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int h2 = 0; h2 < 2; h2++)
#pragma unroll
                for (int g = 0; g < 4; g++)
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const int ks = 2 * g + h2;
                        const double x = (ks & 1) ? xa[u][ks >> 1].y : xa[u][ks >> 1].x;
                        const double y = (ks & 1) ? xb[u][ks >> 1].y : xb[u][ks >> 1].x;
                        pa[u][g] = __builtin_amdgcn_mfma_f64_16x16x4f64(-x, y, pa[u][g], 0, 0, 0);
                    }
#ifdef VBA_STAMPS
            { double sink = pa[0][0][0] + pa[1][3][3] + pa[0][3][0] + pa[1][0][1]; asm volatile("" :: "v"(sink)); }
            FSTAMP(2)
#endif
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const bool on = q_on[u] && (q_row[u] ? presn : upd);
                const bool prod = ride && (!q_row[u] || pres);
                double* Cd = (q_row[u] ? CI : CT) + q_dst[u];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const double v = ca[u][i] + ((pa[u][0][i] + pa[u][1][i]) + (pa[u][2][i] + pa[u][3][i]));
                    if (on) Cd[4 * i * 34] = prod ? v : ca[u][i];
                }
            }
        }
        RSTAMP(7)
        if (wave == 3) {
            if (pres) chain_store_rows_perm(XI, Lf, n, I, J, lane);
            if (owner && ride) chain_store_rows_perm(XL, Lf, n, J + 1, J, lane);
        }
        if (upd) fetch_quadrants(J + 1);
        }
        if (act && wave == 0 && upd) prefetch_ride(J + 1);
        else {   // (defined on every path: the 64 registers of t are free during phase F)
#pragma unroll
            for (int q = 0; q < 32; q++) t[q] = 0.0;
            rr = 0.0;
        }
        RSTAMP(5)
        lds_barrier();   // B: C_{J+1,J+1} and C_{I,J+1} are in LDS (and every read of this column's L rows is done)
        RSTAMP(6)
    }
    if (!any) return;
    if (two) {   // b_I - sum_J L_IJ z_J: the forward substitution of the row, the shares of the two chains added up
        if (half && wave == 1 && hi) rpart[r] = rr_I;
        __syncthreads();
        if (!half && wave == 1 && hi) (B.vec + d.vec0)[(size_t)I * 32 + r] = rr_I + rpart[r];
    } else if (wave == 1 && hi)
        (B.vec + d.vec0)[(size_t)I * 32 + r] = rr_I;
}

// Few-window regime: S_IJ -= sum_{k < nc} L_Ik D_k L_Jk^T for the tiles (I,J), J >= nc, that collect updates from chain columns (the
// rows b_I of the right-hand side have ridden through the chain with their tile row: k_chol_chain_rows).  One workgroup of 512
// threads per tile.  The operands are rows of the row-major factor -- 256 contiguous bytes per tile row and chain column -- so a
// thread fetches ONE 32-byte piece per chain column (64 rows x 8 pieces), all columns requested at once, and hands them to the
// MFMA waves through a double-buffered LDS tile pair: eight waves = four 16x16 quadrants x two halves of the k-steps.  (First
// form: every wave gathering its operand elements itself, 8-byte loads 5.9 KB apart: 28 us per launch.)
#define CHAIN_UPD_MAXK 16
__global__ void __launch_bounds__(512) k_chol_chain_upd(Batch B) {
    __shared__ double T[2][64 * 36];       // rows 0..31: tile (I,k), rows 32..63: tile (J,k); pitch 36
    __shared__ double dk_[2][32];
    __shared__ double red[4][4 * 64];
    const int w = blockIdx.y;
    const WinDesc& d = B.desc[w];
    if (!win_on(d, B.ctrl[w])) return;
    if (d.nc <= 0 || (int)blockIdx.x >= d.n_cu) return;
    const int4 cu = reinterpret_cast<const int4*>(B.tl_cu)[(size_t)d.cu0 + blockIdx.x];   // (I << 16 | J, first, end of the chain part of its list)
    const int I = cu.x >> 16, J = cu.x & 0xffff, kb = cu.y, ke = cu.z;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, l15 = lane & 15, l4 = lane >> 4;
    const int quad = wave & 3, half = wave >> 2, ti = quad >> 1, tj = quad & 1;
    const bool live = !(I == J && tj > ti);
    const int n = d.nS;
    double* S = B.S + d.S0;
    const double* Lf = B.Lf + d.S0;
    const double* dv = B.dvec + d.vec0;
    const int* kl = B.tl_kl + d.tl_k0;
    double* C = S + ((size_t)I * 32 + 16 * ti) * n + (size_t)J * 32 + 16 * tj;
    d4_t acc = {0.0, 0.0, 0.0, 0.0};
    if (half == 0 && live) {
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i] = C[(size_t)(l4 + 4 * i) * n + l15];
    }
    // this thread's piece: row (t >> 3) of the tile pair, doubles 4 (t & 7) .. + 3
    const int prow = t >> 3, pcol = 4 * (t & 7);
    const double* src = Lf + ((size_t)(prow < 32 ? I : J) * 32 + (prow & 31)) * n + pcol;
    d4_t pacc[2] = {d4_t{0.0, 0.0, 0.0, 0.0}, d4_t{0.0, 0.0, 0.0, 0.0}};
    for (int e0 = kb; e0 < ke; e0 += CHAIN_UPD_MAXK) {
        const int ne = min(CHAIN_UPD_MAXK, ke - e0);
        double4 pc[CHAIN_UPD_MAXK];
        double dd[CHAIN_UPD_MAXK];
#pragma unroll
        for (int u = 0; u < CHAIN_UPD_MAXK; u++) {
            const bool on = u < ne;
            const size_t k32 = on ? (size_t)kl[e0 + u] * 32 : 0;
            pc[u] = on ? *reinterpret_cast<const double4*>(src + k32) : make_double4(0.0, 0.0, 0.0, 0.0);
            dd[u] = (on && t < 32) ? dv[k32 + t] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < CHAIN_UPD_MAXK; u++) {
            if (u >= ne) continue;                  // (uniform)
            const int bf = u & 1;
            *reinterpret_cast<double4*>(&T[bf][prow * 36 + pcol]) = pc[u];
            if (t < 32) dk_[bf][t] = dd[u];
            lds_barrier();
            if (live) {
                // this wave's k-steps: 4 half .. 4 half + 3 ; two accumulators (a chain of dependent FP64 MFMAs runs slower)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int kk = 4 * (4 * half + q) + l4;
                    const double av = -T[bf][(16 * ti + l15) * 36 + kk];
                    const double bv = T[bf][(32 + 16 * tj + l15) * 36 + kk] * dk_[bf][kk];
                    pacc[q & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, pacc[q & 1], 0, 0, 0);
                }
            }
            // (the buffer written two columns from now is this one: every wave has passed the barrier of the column in between)
        }
        lds_barrier();
    }
#pragma unroll
    for (int i = 0; i < 4; i++) acc[i] += pacc[0][i] + pacc[1][i];
    if (half == 1) {
#pragma unroll
        for (int i = 0; i < 4; i++) red[quad][64 * i + lane] = acc[i];
    }
    __syncthreads();
    if (half == 0 && live) {
#pragma unroll
        for (int i = 0; i < 4; i++) C[(size_t)(l4 + 4 * i) * n + l15] = acc[i] + red[quad][64 * i + lane];
    }
}
