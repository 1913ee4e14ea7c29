// vba_pcg.h -- preconditioned conjugate gradients on the reduced camera system (BASELINE north_star "LM / PCG", configs[3]).
//
// Not in the reference: its g2o only carries the direct solvers (Thirdparty/g2o/g2o/solvers/linear_solver_{eigen,cholmod,
// dense}.h) and every optimiser of src/Optimizer.cpp uses LinearSolverEigen.  Selected with vba_problem.solver =
// VBA_SOLVER_PCG; the LDL^T path stays the default.  The algorithm is the textbook one g2o's own linear_solver_pcg.h (absent
// from the reference's copy) implements: block-Jacobi preconditioner (the pdim x pdim diagonal block of every keyframe,
// inverted once per solve), CG on S x = b, stop at sqrt(r'M^-1 r / r0'M^-1 r0) <= 1e-10 or after 20 n_p iterations.
//
// S is the matrix the Schur kernels assemble (lower triangle, block-sparse by keyframe pair); one workgroup per window
// mirrors its non-zero blocks above the diagonal once, so that every product walks rows, then iterates alone: no launch and no
// host round trip per CG iteration, all sums in a fixed order.  On the visual-inertial systems of this backend the method is
// slow by nature -- cond(S) ~ 1e9..1e10 (velocity / bias blocks against pose blocks) and block-Jacobi needs ~0.6 n_p
// iterations -- which is why the reference solves directly; see DESIGN.md for the measured comparison.
#pragma once
#include "vba_kernels.h"

#define PCG_TOL 1e-10

// Cholesky-based inverse of one SPD pdim x pdim block (pdim <= 15), by one thread; returns false if not positive definite
DEVI bool pcg_block_inverse(int P, const double* A, int lda, double* Ai) {
    double L[15 * 15], Li[15 * 15];
    for (int i = 0; i < P; i++)
        for (int j = 0; j <= i; j++) {
            double s = A[i * lda + j];
            for (int k = 0; k < j; k++) s -= L[i * 15 + k] * L[j * 15 + k];
            if (i == j) {
                if (!(s > 0.0)) return false;
                L[i * 15 + i] = sqrt(s);
            } else
                L[i * 15 + j] = s / L[j * 15 + j];
        }
    for (int j = 0; j < P; j++)   // Li = L^-1 (lower)
        for (int i = j; i < P; i++) {
            double s = (i == j) ? 1.0 : 0.0;
            for (int k = j; k < i; k++) s -= L[i * 15 + k] * Li[k * 15 + j];
            Li[i * 15 + j] = s / L[i * 15 + i];
        }
    for (int i = 0; i < P; i++)   // A^-1 = Li^T Li
        for (int j = 0; j < P; j++) {
            double s = 0.0;
            for (int k = (i > j ? i : j); k < P; k++) s += Li[k * 15 + i] * Li[k * 15 + j];
            Ai[i * P + j] = s;
        }
    return true;
}

__global__ void __launch_bounds__(256) k_pcg(Batch B) {
    __shared__ double red[4];
    __shared__ int sh_bad;
    const int w = blockIdx.x;
    const WinDesc& d = B.desc[w];
    WinCtrl& c = B.ctrl[w];
    if (!win_on(d, c)) return;
    const int t = threadIdx.x, n = d.nS, P = d.pdim, nf = d.n_free;
    double* S = B.S + d.S0;
    double* xs = B.pcg_v + 5 * (size_t)d.vec0;   // x, r, z, p, q  (nS each)
    double *rs = xs + n, *zs = rs + n, *ps = zs + n, *qs = ps + n;
    double* Mi = B.pcg_m + 225 * (size_t)d.kf0;  // inverted diagonal blocks
    const double* rhs = B.vec + d.vec0;
    const int* ab = B.adj_begin + d.kf0 + d.win;
    const int* adj = B.adj + d.adj0;
    if (t == 0) sh_bad = 0;
    __syncthreads();
    // mirror the non-zero off-diagonal blocks above the diagonal (the Schur kernels write gr >= gc only), invert the diagonal ones
    for (int a = 0; a < nf; a++)
        for (int e = ab[a]; e < ab[a + 1]; e++) {
            const int b = adj[e];
            for (int q = t; q < P * P; q += 256) {
                const int gr = vpos(d, a, q / P), gc = vpos(d, b, q % P);
                if (gr < gc) S[(size_t)gr * n + gc] = S[(size_t)gc * n + gr];
            }
        }
    for (int a = 0; a < nf; a++)
        for (int q = t; q < P * P; q += 256) {
            const int gr = vpos(d, a, q / P), gc = vpos(d, a, q % P);
            if (gr < gc) S[(size_t)gr * n + gc] = S[(size_t)gc * n + gr];
        }
    __syncthreads();
    for (int a = t; a < nf; a += 256) {
        double blk[225];
        for (int i = 0; i < P; i++)
            for (int j = 0; j < P; j++) blk[i * P + j] = S[(size_t)vpos(d, a, i) * n + vpos(d, a, j)];
        if (!pcg_block_inverse(P, blk, P, Mi + 225 * (size_t)a)) sh_bad = 1;
    }
    __syncthreads();
    if (sh_bad) {   // a diagonal block is not positive definite: the same verdict the LDL^T path reaches (linear_solver_eigen.h:105-111)
        if (t == 0) c.chol_fail = 1;
        return;
    }
    const int np = d.np;   // rows vpos(a, r) cover [0, np) exactly once
    auto precond = [&]() {   // z = M^-1 r, one thread per row
        for (int i = t; i < P * nf; i += 256) {
            const int a = i / P, r = i % P;
            const double* m = Mi + 225 * (size_t)a + r * P;
            double s = 0.0;
            for (int q = 0; q < P; q++) s += m[q] * rs[vpos(d, a, q)];
            zs[vpos(d, a, r)] = s;
        }
    };
    auto matvec = [&]() {    // q = S p over the non-zero blocks of every row
        for (int i = t; i < P * nf; i += 256) {
            const int a = i / P, r = i % P;
            const int gr = vpos(d, a, r);
            const double* row = S + (size_t)gr * n;
            double s = 0.0;
            for (int q = 0; q < P; q++) { const int gc = vpos(d, a, q); s += row[gc] * ps[gc]; }
            for (int e = ab[a]; e < ab[a + 1]; e++) {
                const int b = adj[e];
                for (int q = 0; q < P; q++) { const int gc = vpos(d, b, q); s += row[gc] * ps[gc]; }
            }
            qs[gr] = s;
        }
    };
    for (int i = t; i < np; i += 256) { xs[i] = 0.0; rs[i] = rhs[i]; }
    __syncthreads();
    precond();
    __syncthreads();
    double loc = 0.0;
    for (int i = t; i < np; i += 256) { ps[i] = zs[i]; loc += rs[i] * zs[i]; }
    double rz = block_sum256(loc, red);
    const double rz0 = rz;
    int it = 0;
    const int max_it = 20 * np + 50;
    bool bad = false;
    if (rz0 > 0.0)
        for (; it < max_it; it++) {
            __syncthreads();
            matvec();
            __syncthreads();
            loc = 0.0;
            for (int i = t; i < np; i += 256) loc += ps[i] * qs[i];
            const double pq = block_sum256(loc, red);
            if (!(pq > 0.0)) { bad = true; break; }   // not positive definite along p
            const double alpha = rz / pq;
            for (int i = t; i < np; i += 256) { xs[i] += alpha * ps[i]; rs[i] -= alpha * qs[i]; }
            __syncthreads();
            precond();
            __syncthreads();
            loc = 0.0;
            for (int i = t; i < np; i += 256) loc += rs[i] * zs[i];
            const double rz2 = block_sum256(loc, red);
            if (rz2 <= PCG_TOL * PCG_TOL * rz0) { it++; break; }
            const double beta = rz2 / rz;
            for (int i = t; i < np; i += 256) ps[i] = zs[i] + beta * ps[i];
            rz = rz2;
        }
    __syncthreads();
    if (bad || it >= max_it) {
        if (t == 0) c.chol_fail = 1;
        return;
    }
    double* out = B.vec + d.vec0;
    for (int i = t; i < np; i += 256) out[i] = xs[i];
    if (t == 0) c.lin_its += it;
}
