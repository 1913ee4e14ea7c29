// vba_pcg.h -- preconditioned conjugate gradients on the reduced camera system (BASELINE north_star "LM / PCG", configs[3]).
//
// Not in the reference: its g2o only carries the direct solvers (Thirdparty/g2o/g2o/solvers/linear_solver_{eigen,cholmod,
// dense}.h) and every optimiser of src/Optimizer.cpp uses LinearSolverEigen.  Selected with vba_problem.solver =
// VBA_SOLVER_PCG; the LDL^T path stays the default.  The algorithm is the textbook one g2o's own linear_solver_pcg.h (absent
// from the reference's copy) implements, CG on S x = b with a block preconditioner -- block-tridiagonal over the keyframe chain
// (round 4, below; VBA_PCG_JACOBI=1: the block-Jacobi of rounds 1-3, the pdim x pdim diagonal block of every keyframe), stop at sqrt(r'M^-1 r / r0'M^-1 r0) <= 1e-10 or after 20 n_p iterations.
//
// S is the matrix the Schur kernels assemble (lower triangle, block-sparse by keyframe pair).  k_pcg_init mirrors its non-zero
// blocks above the diagonal once per solve, so that every product walks rows, and inverts the diagonal blocks; one CG iteration
// is two launches for the whole batch: k_pcg_matvec (q = S p, a lane per row over the row's non-zero blocks, every 64-row block
// of every window its own workgroup, partial p.q per workgroup) and k_pcg_step (one workgroup per window: alpha, x, r,
// z = M^-1 r, beta, p, the stop test; all sums in a fixed order).  Windows that have converged exit at once.  The host enqueues
// iterations in batches and looks at one pinned word per batch, two batches behind the device (vislam_ba.hip).
// (A first version iterated inside ONE workgroup per window, without launches: 3 ms per CG iteration at C4 size, a single CU
// walking 8.6 MB of blocks -- 800x slower than the LDL^T path.)
// On the visual-inertial systems of this backend the method is slow by nature -- cond(S) ~ 1e9..1e10 (velocity / bias blocks
// against pose blocks) and block-Jacobi needs ~0.5 n_p iterations -- which is why the reference solves directly; see DESIGN.md
// for the measured comparison.
#pragma once
#include "vba_kernels.h"

#define PCG_TOL 1e-10
#define PCG_MBLK 450   // doubles of preconditioner data per keyframe: D_a^-1 (225), G_a (225)
DEVI double* pcg_vec(const Batch& B, const WinDesc& d, int which) { return B.pcg_v + 5 * (size_t)d.vec0 + (size_t)which * d.nS; }   // x r z p q

// Cholesky-based inverse of the SPD pdim x pdim diagonal blocks (pdim <= 15) of one window: sixteen lanes per block -- a lane per
// row of L, then per column of L^-1, then per row of the inverse -- with the block, L (in place) and L^-1 in LDS.  (First version: one
// thread per block with the three 15 x 15 arrays as locals: 5.4 KB of scratch per lane.)  The sums run in the order a scalar
// left-looking Cholesky takes them.  Sets *bad when a block is not positive definite.
#define PCG_BI_STRIDE 16
DEVI void pcg_block_inverses(const Batch& B, const WinDesc& d, double (*sA)[15 * PCG_BI_STRIDE], double (*sLi)[15 * PCG_BI_STRIDE], int* bad) {
    const int t = threadIdx.x, g = t >> 4, i = t & 15, P = d.pdim, nf = d.n_free, n = d.nS;
    const double* S = B.S + d.S0;
    double* Mi = B.pcg_m + PCG_MBLK * (size_t)d.kf0;
    for (int a0 = 0; a0 < nf; a0 += 16) {
        const int a = a0 + g;
        const bool act = a < nf && i < P;
        double* A = sA[g];
        double* Li = sLi[g];
        if (act)
            for (int j = 0; j < P; j++) A[i * PCG_BI_STRIDE + j] = S[(size_t)vpos(d, a, i) * n + vpos(d, a, j)];
        __syncthreads();
        for (int j = 0; j < P; j++) {   // column j of L
            if (act && i == j) {
                double s = A[j * PCG_BI_STRIDE + j];
                for (int k = 0; k < j; k++) s -= A[j * PCG_BI_STRIDE + k] * A[j * PCG_BI_STRIDE + k];
                if (!(s > 0.0)) *bad = 1;
                A[j * PCG_BI_STRIDE + j] = sqrt(s);
            }
            __syncthreads();
            if (act && i > j) {
                double s = A[i * PCG_BI_STRIDE + j];
                for (int k = 0; k < j; k++) s -= A[i * PCG_BI_STRIDE + k] * A[j * PCG_BI_STRIDE + k];
                A[i * PCG_BI_STRIDE + j] = s / A[j * PCG_BI_STRIDE + j];
            }
            __syncthreads();
        }
        if (act) {   // column i of L^-1 (lower)
            for (int r = i; r < P; r++) {
                double s = (r == i) ? 1.0 : 0.0;
                for (int k = i; k < r; k++) s -= A[r * PCG_BI_STRIDE + k] * Li[k * PCG_BI_STRIDE + i];
                Li[r * PCG_BI_STRIDE + i] = s / A[r * PCG_BI_STRIDE + r];
            }
        }
        __syncthreads();
        if (act) {   // row i of A^-1 = L^-T L^-1
            double* out = Mi + PCG_MBLK * (size_t)a + i * P;
            for (int j = 0; j < P; j++) {
                double s = 0.0;
                for (int k = (i > j ? i : j); k < P; k++) s += Li[k * PCG_BI_STRIDE + i] * Li[k * PCG_BI_STRIDE + j];
                out[j] = s;
            }
        }
        __syncthreads();
    }
}

// per-window CG state lives in WinCtrl-independent scratch: pcg_s[8 * win + ..] = rz, rz0, done, iterations, bad, preconditioner
#define PCG_RZ 0
#define PCG_RZ0 1
#define PCG_DONE 2
#define PCG_ITS 3
#define PCG_BAD 4
#define PCG_MODE 5    // 1: block-tridiagonal (keyframe chain) preconditioner, 0: block-Jacobi (fallback)
#define PCG_STATE 8

// ---- block-tridiagonal preconditioner (round 4) -------------------------------------------------------------------------------
// Block-Jacobi ignores exactly the coupling that makes cond(S) ~ 5e9: the IMU chain ties the velocity / bias blocks of consecutive
// keyframes together with information up to 2.5e9 (src/IMU/imudata.cpp:25, src/Optimizer.cpp:244-249).  M = the block-tridiagonal
// part of S in keyframe order -- the 15x15 diagonal blocks and the blocks (a, a-1), IMU and vision terms alike -- factored once per
// solve as M = L D L^T (block Thomas: D_a = S_aa - C_a D_{a-1}^-1 C_a^T, L_{a,a-1} = G_a = C_a D_{a-1}^-1) and applied per CG
// iteration with one forward and one backward sweep over the keyframes (one wave per window, the next keyframe's blocks in
// flight while the current one is multiplied).  The truncation of an SPD matrix need not be SPD: if a D_a is not positive definite
// the window falls back to block-Jacobi.  pcg_m holds per keyframe D_a^-1 (225) and G_a (225).
// the inverse of one SPD P x P block (P <= 15) in LDS: threads 0..15 work, EVERY thread of the workgroup passes the barriers
DEVI void pcg_inv_block(double* A, double* Li, double* out, int P, int* bad) {
    const int i = threadIdx.x;
    const bool act = i < P;
    for (int j = 0; j < P; j++) {   // column j of L (in place)
        if (act && i == j) {
            double s = A[j * PCG_BI_STRIDE + j];
            for (int k = 0; k < j; k++) s -= A[j * PCG_BI_STRIDE + k] * A[j * PCG_BI_STRIDE + k];
            if (!(s > 0.0)) *bad = 1;
            A[j * PCG_BI_STRIDE + j] = sqrt(s);
        }
        __syncthreads();
        if (act && i > j) {
            double s = A[i * PCG_BI_STRIDE + j];
            for (int k = 0; k < j; k++) s -= A[i * PCG_BI_STRIDE + k] * A[j * PCG_BI_STRIDE + k];
            A[i * PCG_BI_STRIDE + j] = s / A[j * PCG_BI_STRIDE + j];
        }
        __syncthreads();
    }
    if (act)
        for (int r = i; r < P; r++) {   // column i of L^-1
            double s = (r == i) ? 1.0 : 0.0;
            for (int k = i; k < r; k++) s -= A[r * PCG_BI_STRIDE + k] * Li[k * PCG_BI_STRIDE + i];
            Li[r * PCG_BI_STRIDE + i] = s / A[r * PCG_BI_STRIDE + r];
        }
    __syncthreads();
    if (act)
        for (int j = 0; j < P; j++) {   // row i of A^-1 = L^-T L^-1
            double s = 0.0;
            for (int k = (i > j ? i : j); k < P; k++) s += Li[k * PCG_BI_STRIDE + i] * Li[k * PCG_BI_STRIDE + j];
            out[i * PCG_BI_STRIDE + j] = s;
        }
    __syncthreads();
}
// block Thomas factorisation of the tridiagonal part (256 threads, S already mirrored); returns through *bad
DEVI void pcg_tridiag_factor(const Batch& B, const WinDesc& d, double* sA, double* sLi, double* sDi, double* sC, double* sG, int* bad) {
    const int t = threadIdx.x, P = d.pdim, nf = d.n_free, n = d.nS;
    const double* S = B.S + d.S0;
    double* M = B.pcg_m + PCG_MBLK * (size_t)d.kf0;
    const int r = t / 15, q = t % 15;   // one entry of a P x P block per thread (t < 225)
    const bool ent = t < 225 && r < P && q < P;
    for (int a = 0; a < nf; a++) {
        double v = 0.0;
        if (ent) v = S[(size_t)vpos(d, a, r) * n + vpos(d, a, q)];
        if (a > 0) {
            if (ent) sC[r * PCG_BI_STRIDE + q] = S[(size_t)vpos(d, a, r) * n + vpos(d, a - 1, q)];
            __syncthreads();
            if (ent) {   // G_a = C_a D_{a-1}^-1
                double g = 0.0;
                for (int k = 0; k < P; k++) g += sC[r * PCG_BI_STRIDE + k] * sDi[k * PCG_BI_STRIDE + q];
                sG[r * PCG_BI_STRIDE + q] = g;
                M[PCG_MBLK * (size_t)a + 225 + r * P + q] = g;
            }
            __syncthreads();
            if (ent)     // D_a = S_aa - G_a C_a^T
                for (int k = 0; k < P; k++) v -= sG[r * PCG_BI_STRIDE + k] * sC[q * PCG_BI_STRIDE + k];
        }
        if (ent) sA[r * PCG_BI_STRIDE + q] = v;
        __syncthreads();
        pcg_inv_block(sA, sLi, sDi, P, bad);
        if (ent) M[PCG_MBLK * (size_t)a + r * P + q] = sDi[r * PCG_BI_STRIDE + q];
        __syncthreads();
        if (*bad) return;   // (uniform: read behind a barrier)
    }
}
// z = M^-1 r with the factors above: ONE wave (lanes: row = lane & 15, four lanes share a row's dot product); y lives in LDS
DEVI void pcg_precond_tri(const Batch& B, const WinDesc& d, double* y) {
    const int lane = threadIdx.x & 63, r = lane & 15, part = lane >> 4, P = d.pdim, nf = d.n_free;
    const double* M = B.pcg_m + PCG_MBLK * (size_t)d.kf0;
    const double* rs = pcg_vec(B, d, 1);
    double* zs = pcg_vec(B, d, 2);
    const bool row = r < P;
    auto dot4 = [&](const double (&m)[4], const double* x) {   // this lane's share: columns part, part + 4, part + 8, part + 12
        double s = 0.0;
#pragma unroll
        for (int u = 0; u < 4; u++) { const int c = part + 4 * u; if (c < P) s += m[u] * x[c]; }
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        return s;
    };
    auto load4 = [&](const double* blk, bool transposed, double (&m)[4]) {
#pragma unroll
        for (int u = 0; u < 4; u++) { const int c = part + 4 * u; m[u] = (row && c < P) ? (transposed ? blk[c * P + r] : blk[r * P + c]) : 0.0; }
    };
    // The sweeps are chains of nf dependent 15x15 products; what they must never wait for is memory: the blocks of the next PF
    // keyframes are always in flight (a block comes from L2 in ~1-2 us, a step takes ~0.1 us; with one step of look-ahead a
    // C4 sweep cost 0.7 ms per CG iteration).
    constexpr int PF = 8;
    // forward: y_a = r_a - G_a y_{a-1}
    {
        double g[PF][4], rv[PF];
        // (every fetch issues the same loads whatever a is -- out-of-range keyframes are clamped and their values never used: a load
        // under a branch makes the compiler wait for ALL outstanding loads, vmcnt(0), at the next use)
        auto fetch = [&](int a, int slot) {
            const int ac = a < 1 ? 1 : (a >= nf ? nf - 1 : a);
            load4(M + PCG_MBLK * (size_t)(nf > 1 ? ac : 0) + 225, false, g[slot]);
            rv[slot] = rs[vpos(d, a < nf ? a : nf - 1, row ? r : 0)];
        };
#pragma unroll
        for (int u = 0; u < PF; u++) fetch(u, u);
        int a0 = 0;
        for (; a0 + PF <= nf; a0 += PF) {            // whole groups: straight-line code, the loads counted exactly
#pragma unroll
            for (int u = 0; u < PF; u++) {
                const int a = a0 + u;
                double v = rv[u];
                const double dp = dot4(g[u], y + 16 * (a > 0 ? a - 1 : 0));
                v = (a > 0) ? v - dp : v;
                fetch(a + PF, u);                     // (the slot is free: its block has just been used)
                if (part == 0 && row) y[16 * a + r] = v;
                wave_lds_sync();
            }
        }
#pragma unroll
        for (int u = 0; u < PF; u++) {                // the last nf % PF keyframes: their blocks are in the slots already
            const int a = a0 + u;
            if (a < nf) {
                double v = rv[u];
                if (a > 0) v -= dot4(g[u], y + 16 * (a - 1));
                if (part == 0 && row) y[16 * a + r] = v;
                wave_lds_sync();
            }
        }
    }
    // backward: z_a = D_a^-1 y_a - G_{a+1}^T z_{a+1}   (z overwrites y in LDS)
    {
        double di[PF][4], gt[PF][4];
        auto fetch = [&](int a, int slot) {   // what keyframe a needs: D_a^-1 and G_{a+1}^T (clamped like above)
            const int ac = a < 0 ? 0 : a;
            load4(M + PCG_MBLK * (size_t)ac, false, di[slot]);
            load4(M + PCG_MBLK * (size_t)(ac + 1 < nf ? ac + 1 : ac) + 225, true, gt[slot]);
        };
#pragma unroll
        for (int u = 0; u < PF; u++) fetch(nf - 1 - u, u);
        int a0 = nf - 1;
        for (; a0 - PF + 1 >= 0; a0 -= PF) {
#pragma unroll
            for (int u = 0; u < PF; u++) {
                const int a = a0 - u;
                double v = dot4(di[u], y + 16 * a);
                const double dp = dot4(gt[u], y + 16 * (a + 1 < nf ? a + 1 : a));
                v = (a + 1 < nf) ? v - dp : v;
                fetch(a - PF, u);
                wave_lds_sync();
                if (part == 0 && row) { y[16 * a + r] = v; zs[vpos(d, a, r)] = v; }
                wave_lds_sync();
            }
        }
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const int a = a0 - u;
            if (a >= 0) {
                double v = dot4(di[u], y + 16 * a);
                if (a + 1 < nf) v -= dot4(gt[u], y + 16 * (a + 1));
                wave_lds_sync();
                if (part == 0 && row) { y[16 * a + r] = v; zs[vpos(d, a, r)] = v; }
                wave_lds_sync();
            }
        }
    }
}
#define PCG_ROWS 64   // rows per matvec workgroup


// z = M^-1 r for rows [i0, i1) of the window, one thread per row
DEVI void pcg_precond(const Batch& B, const WinDesc& d, int t, int nt) {
    const int P = d.pdim;
    const double* Mi = B.pcg_m + PCG_MBLK * (size_t)d.kf0;
    const double* rs = pcg_vec(B, d, 1);
    double* zs = pcg_vec(B, d, 2);
    for (int i = t; i < P * d.n_free; i += nt) {
        const int a = i / P, r = i % P;
        const double* m = Mi + PCG_MBLK * (size_t)a + r * P;
        double s = 0.0;
        for (int q = 0; q < P; q++) s += m[q] * rs[vpos(d, a, q)];
        zs[vpos(d, a, r)] = s;
    }
}

// once per solve: mirror S, invert the diagonal blocks, x = 0, r = b, z = M^-1 r, p = z, rz
__global__ void __launch_bounds__(256) k_pcg_init(Batch B) {
    __shared__ double red[4];
    __shared__ int sh_bad;
    __shared__ double sA[16][15 * PCG_BI_STRIDE], sLi[16][15 * PCG_BI_STRIDE];
    extern __shared__ double pcg_y[];   // 16 doubles per keyframe: the sweeps of the tridiagonal preconditioner
    const int w = blockIdx.x;
    const WinDesc& d = B.desc[w];
    const WinCtrl& c = B.ctrl[w];
    double* st = B.pcg_s + PCG_STATE * (size_t)d.win;
    const int t = threadIdx.x, n = d.nS, P = d.pdim, nf = d.n_free, np = d.np;
    if (!win_on(d, c)) {
        if (t == 0) st[PCG_DONE] = 1.0;   // not part of this solve
        return;
    }
    double* S = B.S + d.S0;
    const int* ab = B.adj_begin + d.kf0 + d.win;
    const int* adj = B.adj + d.adj0;
    if (t == 0) sh_bad = 0;
    __syncthreads();
    for (int a = 0; a < nf; a++) {   // the Schur kernels write gr >= gc only
        for (int e = ab[a]; e <= ab[a + 1]; e++) {
            const int b = (e < ab[a + 1]) ? adj[e] : a;   // the diagonal block last
            for (int q = t; q < P * P; q += 256) {
                const int gr = vpos(d, a, q / P), gc = vpos(d, b, q % P);
                if (gr < gc) S[(size_t)gr * n + gc] = S[(size_t)gc * n + gr];
            }
        }
    }
    __syncthreads();
    int mode = B.pcg_tri ? 1 : 0;
    if (mode) {   // block Thomas on the tridiagonal part; not positive definite -> block-Jacobi
        pcg_tridiag_factor(B, d, sA[0], sLi[0], sA[1], sA[2], sA[3], &sh_bad);
        __syncthreads();
        if (sh_bad) { mode = 0; __syncthreads(); if (t == 0) sh_bad = 0; __syncthreads(); }
    }
    if (!mode) pcg_block_inverses(B, d, sA, sLi, &sh_bad);
    double *xs = pcg_vec(B, d, 0), *rs = pcg_vec(B, d, 1), *zs = pcg_vec(B, d, 2), *ps = pcg_vec(B, d, 3);
    const double* rhs = B.vec + d.vec0;
    for (int i = t; i < np; i += 256) { xs[i] = 0.0; rs[i] = rhs[i]; }
    __syncthreads();
    if (sh_bad) {   // a diagonal block is not positive definite: the verdict the LDL^T path reaches (linear_solver_eigen.h:105-111)
        if (t == 0) { st[PCG_DONE] = 1.0; st[PCG_BAD] = 1.0; st[PCG_ITS] = 0.0; }
        return;
    }
    if (mode) { if (t < 64) pcg_precond_tri(B, d, pcg_y); }
    else pcg_precond(B, d, t, 256);
    __syncthreads();
    double loc = 0.0;
    for (int i = t; i < np; i += 256) { ps[i] = zs[i]; loc += rs[i] * zs[i]; }
    const double rz = block_sum256(loc, red);
    if (t == 0) {
        st[PCG_MODE] = (double)mode;
        st[PCG_RZ] = rz; st[PCG_RZ0] = rz; st[PCG_ITS] = 0.0; st[PCG_BAD] = 0.0;
        st[PCG_DONE] = (rz > 0.0) ? 0.0 : 1.0;   // zero right-hand side: x = 0
    }
}

// q = S p for PCG_ROWS rows of one window: 16 lanes per row share the row's non-zero blocks (the diagonal block and every 16th
// neighbour each), a fixed butterfly adds them up; partial p.q of the workgroup's rows
__global__ void __launch_bounds__(256) k_pcg_matvec(Batch B) {
    __shared__ double red[4];
    const int w = blockIdx.y;
    const WinDesc& d = B.desc[w];
    const double* st = B.pcg_s + PCG_STATE * (size_t)d.win;
    if (st[PCG_DONE] != 0.0) return;
    const int P = d.pdim, n = d.nS;
    if ((int)blockIdx.x * PCG_ROWS >= P * d.n_free) return;
    const int t = threadIdx.x, sub = t & 15;
    const double* S = B.S + d.S0;
    const double* ps = pcg_vec(B, d, 3);
    double* qs = pcg_vec(B, d, 4);
    const int* ab = B.adj_begin + d.kf0 + d.win;
    const int* adj = B.adj + d.adj0;
    double pq = 0.0;
#pragma unroll
    for (int rr = 0; rr < PCG_ROWS / 16; rr++) {
        const int i = blockIdx.x * PCG_ROWS + rr * 16 + (t >> 4);
        double s = 0.0;
        int gr = 0;
        if (i < P * d.n_free) {
            const int a = i / P, r = i % P;
            gr = vpos(d, a, r);
            const double* row = S + (size_t)gr * n;
            const int e0 = ab[a], ne = ab[a + 1] - e0;
            for (int e = sub; e <= ne; e += 16) {   // e == ne: the diagonal block
                const int b = (e < ne) ? adj[e0 + e] : a;
                for (int q = 0; q < P; q++) { const int gc = vpos(d, b, q); s += row[gc] * ps[gc]; }
            }
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (sub == 0 && i < P * d.n_free) {
            qs[gr] = s;
            pq += ps[gr] * s;
        }
    }
    pq = block_sum256(pq, red);
    if (t == 0) B.part[d.part0 + blockIdx.x] = pq;   // (the chi2 partial array is free during the linear solve)
}

// the rest of the iteration, one workgroup per window
__global__ void __launch_bounds__(256) k_pcg_step(Batch B, int* alive) {
    __shared__ double red[4];
    extern __shared__ double pcg_y[];
    const int w = blockIdx.x;
    const WinDesc& d = B.desc[w];
    double* st = B.pcg_s + PCG_STATE * (size_t)d.win;
    if (st[PCG_DONE] != 0.0) return;
    const int t = threadIdx.x, np = d.np;
    const int nblk = (d.pdim * d.n_free + PCG_ROWS - 1) / PCG_ROWS;
    double loc = 0.0;
    for (int k = t; k < nblk; k += 256) loc += B.part[d.part0 + k];
    const double pq = block_sum256(loc, red);
    const double rz = st[PCG_RZ], rz0 = st[PCG_RZ0];
    const int it = (int)st[PCG_ITS] + 1;
    __syncthreads();
    if (!(pq > 0.0)) {   // not positive definite along p
        if (t == 0) { st[PCG_DONE] = 1.0; st[PCG_BAD] = 1.0; st[PCG_ITS] = it; }
        return;
    }
    const double alpha = rz / pq;
    double *xs = pcg_vec(B, d, 0), *rs = pcg_vec(B, d, 1), *zs = pcg_vec(B, d, 2), *ps = pcg_vec(B, d, 3), *qs = pcg_vec(B, d, 4);
    for (int i = t; i < np; i += 256) { xs[i] += alpha * ps[i]; rs[i] -= alpha * qs[i]; }
    __syncthreads();
    if (st[PCG_MODE] != 0.0) { if (t < 64) pcg_precond_tri(B, d, pcg_y); }
    else pcg_precond(B, d, t, 256);
    __syncthreads();
    loc = 0.0;
    for (int i = t; i < np; i += 256) loc += rs[i] * zs[i];
    const double rz2 = block_sum256(loc, red);
    const bool conv = rz2 <= PCG_TOL * PCG_TOL * rz0;
    const bool out_of_its = it >= 20 * np + 50;
    if (!conv && !out_of_its) {
        const double beta = rz2 / rz;
        for (int i = t; i < np; i += 256) ps[i] = zs[i] + beta * ps[i];
    }
    if (t == 0) {
        st[PCG_RZ] = rz2;
        st[PCG_ITS] = it;
        if (conv) st[PCG_DONE] = 1.0;
        else if (out_of_its) { st[PCG_DONE] = 1.0; st[PCG_BAD] = 1.0; }
        else __hip_atomic_store(alive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// solution -> B.vec, verdict -> WinCtrl
__global__ void __launch_bounds__(256) k_pcg_finish(Batch B) {
    const int w = blockIdx.x;
    const WinDesc& d = B.desc[w];
    WinCtrl& c = B.ctrl[w];
    if (!win_on(d, c)) return;
    const double* st = B.pcg_s + PCG_STATE * (size_t)d.win;
    const int t = threadIdx.x;
    if (st[PCG_BAD] != 0.0 || st[PCG_DONE] == 0.0) {   // breakdown, or the host gave up enqueuing
        if (t == 0) c.chol_fail = 1;
        return;
    }
    const double* xs = pcg_vec(B, d, 0);
    double* out = B.vec + d.vec0;
    for (int i = t; i < d.np; i += 256) out[i] = xs[i];
    if (t == 0) c.lin_its += (int)st[PCG_ITS];
}
