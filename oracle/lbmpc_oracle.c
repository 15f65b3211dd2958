/*
 * lbmpc_oracle.c -- TEST INFRASTRUCTURE ONLY.  Scalar CPU restatement (plain C, fp64) of the
 * hot path of gabormarko/learning-based-mpc, used as the parity oracle for the HIP library and
 * as the `cpu_baseline` ("port") leg of bench.py.  Nothing in the product
 * (learning-based-mpc_amd/) may import, link or call this file.
 *
 * The reference is MATLAB on un-vendored toolboxes (MPT3, YALMIP, CasADi+IPOPT; versions not
 * pinned, README.md:25-30); it cannot be built or run here.  This file restates the published
 * algorithms at the reference's call sites and is pinned against the reference's stored outputs
 * (the .npz files under tests/golden, decoded from its .mat files): see tests/test_oracle_golden.py.
 * Parity UNPINNED (no stored reference output exists): non-binding components of v, the
 * (s, alpha) of MRPISet, everything 3x3/4x4, NominalRobustMPC outputs, the RNG streams.
 *
 * Arithmetic contract shared with the HIP kernels (so that flags / active sets agree exactly):
 * built with -ffp-contract=off; every dot product is an fma chain in ascending index order from
 * the stated initial value; reductions by max/min are order-free; argmin ties go to the lowest row.
 *
 * Matrices are column-major (MATLAB layout) exactly as in include/lbmpc.h.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/lbmpc.h"

#define CM(M, rows, i, j) ((M)[(size_t)(i) + (size_t)(rows) * (size_t)(j)])

#define QP_TOL_VIOL 1e-10 /* a row is violated if slack < -QP_TOL_VIOL                 */
#define QP_TOL_DEP 1e-13  /* new normal is dependent if delta <= QP_TOL_DEP * (c.g)    */
#define QP_TOL_R 1e-12    /* r_j counts as positive if > QP_TOL_R * max|r|             */
#define QP_MAX_ITERS 200

/* ------------------------------------------------------------------------------------------
 * RNG: Philox4x32-10 (Salmon et al., SC'11), counter-based so that CPU and GPU streams are
 * bit-identical for any sharding.  The reference uses MATLAB's unseeded global rand
 * (InitialSetComputation.m:51,111; ModelingCar.m:43,57,80): not reproducible, parity unpinned.
 * ---------------------------------------------------------------------------------------- */
static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

/* one 32-bit word -> uniform in (0, 1): (w + 1/2) 2^-32 (exact in fp64) */
static double u01(uint32_t w) { return ((double)w + 0.5) * 0x1.0p-32; }

enum { P_I0_U = 0, P_I0_EV = 1, P_I0_LV = 2, P_ST_LV = 3, P_ST_U = 4, P_ST_EV = 5 };

/* `n` <= 4 uniforms of stream (seed, trial): draw (index, purpose, attempt) is ONE Philox block */
static void draw_uniforms(uint64_t seed, uint64_t trial, uint32_t index, uint32_t purpose,
                          uint32_t attempt, int n, double* out) {
    uint32_t c[4] = {(uint32_t)trial, (uint32_t)(trial >> 32), index, (purpose << 28) | (attempt << 4)};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    for (int k = 0; k < n && k < 4; ++k) out[k] = u01(c[k]);
}

/* Rejection sampling from the bounding box of the vertices against A x <= b:
 * InitialSetComputation.m:101-122, ModelingCar.m:47-91.  Returns 0, or 1 at the attempt cap. */
static int sample_polytope(const double* XA, const double* Xb, int nr, const double* lo,
                           const double* hi, const double* U /* nx x nx col-major or NULL */, int nx, uint64_t seed,
                           uint64_t trial, uint32_t index, uint32_t purpose, int max_attempts, double* x) {
    double u[LBMPC_MAX_NX], t[LBMPC_MAX_NX];
    for (int a = 0; a < max_attempts; ++a) {
        draw_uniforms(seed, trial, index, purpose, (uint32_t)a, nx, u);
        for (int k = 0; k < nx; ++k) t[k] = lo[k] + (hi[k] - lo[k]) * u[k];
        for (int k = 0; k < nx; ++k) {
            if (!U) { x[k] = t[k]; continue; }
            double acc = 0.0;
            for (int i = 0; i < nx; ++i) acc = fma(CM(U, nx, k, i), t[i], acc);
            x[k] = acc;
        }
        int ok = 1;
        for (int r = 0; r < nr; ++r) {
            double acc = 0.0;
            for (int k = 0; k < nx; ++k) acc = fma(CM(XA, nr, r, k), x[k], acc);
            if (!(acc <= Xb[r])) { ok = 0; break; }
        }
        if (ok) return 0;
    }
    return 1;
}

static void sample_u_lv(const lbmpc_sampler* s, uint64_t seed, uint64_t trial, uint32_t index,
                        uint32_t purpose, double* u) {
    double r[LBMPC_MAX_NZ];
    draw_uniforms(seed, trial, index, purpose, 0, s->nu, r);
    for (int j = 0; j < s->nu; ++j) u[j] = (s->max_u_LV[j] - s->min_u_LV[j]) * r[j] + s->min_u_LV[j];
}

static void mat_vec(const double* M, int rows, int cols, const double* x, double* y) {
    for (int i = 0; i < rows; ++i) {
        double acc = 0.0;
        for (int k = 0; k < cols; ++k) acc = fma(CM(M, rows, i, k), x[k], acc);
        y[i] = acc;
    }
}

/* one information-set sample w = xi_EV - xi_LV - B u_LV  (InitialSetComputation.m:51-52) */
static int sample_w_I0(const lbmpc_sampler* s, uint64_t seed, uint64_t set, uint32_t i,
                       int max_attempts, double* w) {
    double u[LBMPC_MAX_NZ], ev[LBMPC_MAX_NX], lv[LBMPC_MAX_NX], Bu[LBMPC_MAX_NX];
    int st = 0;
    sample_u_lv(s, seed, set, i, P_I0_U, u);
    st |= sample_polytope(s->EV_A, s->EV_b, s->nEV, s->EV_lo, s->EV_hi, s->EV_U, s->nx, seed, set, i, P_I0_EV, max_attempts, ev);
    st |= sample_polytope(s->LV_A, s->LV_b, s->nLV, s->LV_lo, s->LV_hi, s->LV_U, s->nx, seed, set, i, P_I0_LV, max_attempts, lv);
    mat_vec(s->B, s->nx, s->nu, u, Bu);
    for (int k = 0; k < s->nx; ++k) w[k] = (ev[k] - lv[k]) - Bu[k];
    return st;
}

/* ------------------------------------------------------------------------------------------
 * Subsystem 1: support functions (MPT3 W.support with W in V-rep; MRPISet.m:27-63)
 * ---------------------------------------------------------------------------------------- */
static double support_V(const double* Vw, int nV, int nx, const double* p) {
    double best = -INFINITY;
    for (int v = 0; v < nV; ++v) {
        double acc = 0.0;
        for (int k = 0; k < nx; ++k) acc = fma(CM(Vw, nV, v, k), p[k], acc);
        if (acc > best) best = acc;
    }
    return best;
}

static void mat_mul(const double* X, const double* Y, int n, double* Z) { /* Z = X Y, n x n */
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double acc = 0.0;
            for (int k = 0; k < n; ++k) acc = fma(CM(X, n, i, k), CM(Y, n, k, j), acc);
            CM(Z, n, i, j) = acc;
        }
}

/* Gt[t] = Vw * P_t' (nV x nx), so that h_W(P_t d) = max_v Gt[t][v,:].d ; P_t = (A^t)' or A^t */
static void stack_G(const double* A, int nx, const double* Vw, int nV, int s, int transpose_flag, double* G) {
    double P[LBMPC_MAX_NX * LBMPC_MAX_NX], Q[LBMPC_MAX_NX * LBMPC_MAX_NX];
    for (int i = 0; i < nx; ++i)
        for (int j = 0; j < nx; ++j) CM(P, nx, i, j) = (i == j) ? 1.0 : 0.0;
    for (int t = 0; t < s; ++t) {
        double* Gt = G + (size_t)t * nV * nx;
        for (int v = 0; v < nV; ++v)
            for (int k = 0; k < nx; ++k) {
                /* (P_t d).j = sum_k P_t[j,k] d_k ; Vw[v,:] . (P_t d) = sum_k (sum_j Vw[v,j] P_t[j,k]) d_k */
                double acc = 0.0;
                for (int j = 0; j < nx; ++j) {
                    double pjk = transpose_flag ? CM(P, nx, k, j) : CM(P, nx, j, k); /* P holds A^t */
                    acc = fma(CM(Vw, nV, v, j), pjk, acc);
                }
                CM(Gt, nV, v, k) = acc;
            }
        mat_mul(P, A, nx, Q);
        memcpy(P, Q, sizeof(double) * nx * nx);
    }
}

int orc_support_batch(const double* A, int nx, const double* Vw, int nV, int s, double alpha,
                      const double* D, int64_t nd, int transpose_flag, double* h_out) {
    if (!A || !Vw || !D || !h_out || nx < 1 || nV < 1 || s < 1 || nd < 0) return LBMPC_ERR_ARG;
    if (nx > LBMPC_MAX_NX || nV > LBMPC_MAX_NV || s > LBMPC_MAX_S || (long)s * nV > 2048) return LBMPC_ERR_DIM;
    double* G = (double*)malloc(sizeof(double) * (size_t)s * nV * nx);
    stack_G(A, nx, Vw, nV, s, transpose_flag, G);
    double scale = 1.0 / (1.0 - alpha);
    for (int64_t i = 0; i < nd; ++i) {
        const double* d = D + (size_t)i * nx;
        double sum = 0.0;
        for (int t = 0; t < s; ++t) sum += support_V(G + (size_t)t * nV * nx, nV, nx, d);
        h_out[i] = sum * scale;
    }
    free(G);
    return 0;
}

int orc_mrpi_s_alpha(const double* A, int nx, const double* Vw, int nV, const double* WA,
                     const double* Wb, int nW, const double* Dirs, int ndirs, double eps, int mode,
                     int max_s, int* s_out, double* alpha_out, double* Ms_out) {
    if (!A || !Vw || !WA || !Wb || !Dirs || !s_out || !alpha_out) return LBMPC_ERR_ARG;
    if (nx > LBMPC_MAX_NX || nV > LBMPC_MAX_NV || nW > 65536 || max_s > LBMPC_MAX_S) return LBMPC_ERR_DIM;
    int tr = (mode == LBMPC_MS_RAKOVIC);
    /* G[t] for t = 0..max_s (A^t or (A^t)') */
    double* G = (double*)malloc(sizeof(double) * (size_t)(max_s + 1) * nV * nx);
    stack_G(A, nx, Vw, nV, max_s + 1, tr, G);
    double* sums = (double*)calloc((size_t)ndirs, sizeof(double));
    double alpha = 1000.0, Ms = 1000.0, Mmax = -INFINITY;
    int s = 0, rc = 0;
    while (alpha > eps / (eps + Ms)) {
        if (s >= max_s) { rc = LBMPC_ST_ITER_CAP; break; }
        s += 1;
        /* alpha = max_j h_W(P_s f_j) / g_j, f_j = row j of W.A   (MRPISet.m:29) */
        alpha = -INFINITY;
        for (int j = 0; j < nW; ++j) {
            double f[LBMPC_MAX_NX];
            for (int k = 0; k < nx; ++k) f[k] = CM(WA, nW, j, k);
            double a = support_V(G + (size_t)s * nV * nx, nV, nx, f) / Wb[j];
            if (a > alpha) alpha = a;
        }
        /* Ms: literal forms use t = 1..s (MRPISet.m:42-46; Functions/MRPISet.m:12-15), Rakovic t = 0..s-1 */
        int tnew = tr ? s - 1 : s;
        for (int d = 0; d < ndirs; ++d) {
            double h = support_V(G + (size_t)tnew * nV * nx, nV, nx, Dirs + (size_t)d * nx);
            sums[d] += h;
            if (h > Mmax) Mmax = h;
        }
        if (mode == LBMPC_MS_MAX_LITERAL) Ms = Mmax;
        else {
            Ms = -INFINITY;
            for (int d = 0; d < ndirs; ++d) if (sums[d] > Ms) Ms = sums[d];
        }
    }
    *s_out = s; *alpha_out = alpha;
    if (Ms_out) *Ms_out = Ms;
    free(G); free(sums);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * Subsystem 2: halfspace maxima + homothety LP
 * ---------------------------------------------------------------------------------------- */
int orc_halfspace_max(const double* WA, int nW, int nx, const double* samples, int64_t N,
                      int64_t batch, int64_t batch_stride, double* m_out) {
    if (!WA || !samples || !m_out || N < 1 || batch < 0) return LBMPC_ERR_ARG;
    if (nx > LBMPC_MAX_NX || nW > LBMPC_MAX_NW) return LBMPC_ERR_DIM;
    for (int64_t b = 0; b < batch; ++b) {
        const double* S = samples + (size_t)b * batch_stride;
        double m[LBMPC_MAX_NW];
        for (int j = 0; j < nW; ++j) m[j] = -INFINITY;
        for (int64_t i = 0; i < N; ++i) {
            const double* w = S + (size_t)i * nx;
            for (int j = 0; j < nW; ++j) {
                double acc = 0.0;
                for (int k = 0; k < nx; ++k) acc = fma(CM(WA, nW, j, k), w[k], acc);
                if (acc > m[j]) m[j] = acc;
            }
        }
        for (int j = 0; j < nW; ++j) m_out[(size_t)b * nW + j] = m[j];
    }
    return 0;
}

int orc_sample_information_sets(const lbmpc_sampler* smp, const double* WA, int nW, int64_t N,
                                int64_t batch, uint64_t seed, uint64_t set0, int max_attempts,
                                double* samples_out, double* m_out, int* status_out) {
    int nx = smp->nx;
    for (int64_t b = 0; b < batch; ++b) {
        double m[LBMPC_MAX_NW];
        int st = 0;
        for (int j = 0; j < nW; ++j) m[j] = -INFINITY;
        for (int64_t i = 0; i < N; ++i) {
            double w[LBMPC_MAX_NX];
            if (sample_w_I0(smp, seed, set0 + (uint64_t)b, (uint32_t)i, max_attempts, w)) st = LBMPC_ST_SAMPLER_CAP;
            if (samples_out) memcpy(samples_out + ((size_t)b * N + i) * nx, w, sizeof(double) * nx);
            for (int j = 0; j < nW; ++j) {
                double acc = 0.0;
                for (int k = 0; k < nx; ++k) acc = fma(CM(WA, nW, j, k), w[k], acc);
                if (acc > m[j]) m[j] = acc;
            }
        }
        if (m_out) for (int j = 0; j < nW; ++j) m_out[(size_t)b * nW + j] = m[j];
        if (status_out) status_out[b] = st;
    }
    return 0;
}

int orc_sample_polytope(const double* XA, const double* Xb, int nr, int nx, const double* lo, const double* hi,
                        int64_t n, uint64_t seed, uint64_t stream, uint32_t index0, int purpose, int max_attempts,
                        double* out, int* status_out) {
    if (nx > LBMPC_MAX_NX || nr > LBMPC_MAX_NXI) return LBMPC_ERR_DIM;
    for (int64_t i = 0; i < n; ++i) {
        int cap = sample_polytope(XA, Xb, nr, lo, hi, NULL, nx, seed, stream, index0 + (uint32_t)i, (uint32_t)purpose, max_attempts, out + (size_t)i * nx);
        if (status_out) status_out[i] = cap ? LBMPC_ST_SAMPLER_CAP : 0;
    }
    return 0;
}

/* Gaussian elimination with partial pivoting, n <= 8. returns 0 if singular */
static int solve_dense(int n, double* M /* n x n row-major */, double* rhs) {
    for (int c = 0; c < n; ++c) {
        int piv = c;
        for (int r = c + 1; r < n; ++r) if (fabs(M[r * n + c]) > fabs(M[piv * n + c])) piv = r;
        if (fabs(M[piv * n + c]) < 1e-13) return 0;
        if (piv != c) {
            for (int k = 0; k < n; ++k) { double t = M[c * n + k]; M[c * n + k] = M[piv * n + k]; M[piv * n + k] = t; }
            double t = rhs[c]; rhs[c] = rhs[piv]; rhs[piv] = t;
        }
        for (int r = c + 1; r < n; ++r) {
            double f = M[r * n + c] / M[c * n + c];
            for (int k = c; k < n; ++k) M[r * n + k] -= f * M[c * n + k];
            rhs[r] -= f * rhs[c];
        }
    }
    for (int c = n - 1; c >= 0; --c) {
        double acc = rhs[c];
        for (int k = c + 1; k < n; ++k) acc -= M[c * n + k] * rhs[k];
        rhs[c] = acc / M[c * n + c];
    }
    return 1;
}

/* The LP of InitialSetComputation.m:69-83 / UQRobustMPC.m:124-139 after the max-reduction, for a
 * GENERAL polytope W, solved exactly by enumerating the vertices of its feasible set in
 * (beta, y) in R^{nx+1} (2 nW + 2 rows): slow, obviously correct, used to pin alpha against the
 * reference's stored Alpha_ini and to validate the closed form below.  Returns the best vertex
 * (lowest-index subset on ties). */
int orc_homothety_lp_enum(const double* WA, const double* Wb, int nW, int nx, const double* m,
                          double* alpha, double* v, double* y_out) {
    int nv = nx + 1, nr = 2 * nW + 2;
    if (nW > LBMPC_MAX_NW || nx > LBMPC_MAX_NX) return LBMPC_ERR_DIM;
    double R[(2 * LBMPC_MAX_NW + 2) * (LBMPC_MAX_NX + 1)], rhs[2 * LBMPC_MAX_NW + 2];
    /* rows: R x <= rhs with x = (beta, y) */
    for (int j = 0; j < nW; ++j) {
        R[j * nv] = -Wb[j];
        for (int k = 0; k < nx; ++k) R[j * nv + 1 + k] = CM(WA, nW, j, k);
        rhs[j] = 0.0;
        R[(nW + j) * nv] = Wb[j];
        for (int k = 0; k < nx; ++k) R[(nW + j) * nv + 1 + k] = -CM(WA, nW, j, k);
        rhs[nW + j] = Wb[j] - m[j];
    }
    for (int k = 0; k < nv; ++k) { R[(2 * nW) * nv + k] = 0.0; R[(2 * nW + 1) * nv + k] = 0.0; }
    R[(2 * nW) * nv] = -1.0; rhs[2 * nW] = 0.0;     /* -beta <= 0 */
    R[(2 * nW + 1) * nv] = 1.0; rhs[2 * nW + 1] = 1.0; /* beta <= 1 */
    int idx[LBMPC_MAX_NX + 1];
    for (int k = 0; k < nv; ++k) idx[k] = k;
    double best = -INFINITY, bx[LBMPC_MAX_NX + 1];
    int found = 0;
    for (;;) {
        double M[(LBMPC_MAX_NX + 1) * (LBMPC_MAX_NX + 1)], x[LBMPC_MAX_NX + 1];
        for (int a = 0; a < nv; ++a) {
            for (int k = 0; k < nv; ++k) M[a * nv + k] = R[idx[a] * nv + k];
            x[a] = rhs[idx[a]];
        }
        if (solve_dense(nv, M, x)) {
            int ok = 1;
            for (int r = 0; r < nr && ok; ++r) {
                double acc = 0.0;
                for (int k = 0; k < nv; ++k) acc += R[r * nv + k] * x[k];
                if (acc > rhs[r] + 1e-11) ok = 0;
            }
            if (ok && x[0] > best) { best = x[0]; memcpy(bx, x, sizeof(double) * nv); found = 1; }
        }
        /* next combination */
        int p = nv - 1;
        while (p >= 0 && idx[p] == nr - nv + p) --p;
        if (p < 0) break;
        ++idx[p];
        for (int k = p + 1; k < nv; ++k) idx[k] = idx[k - 1] + 1;
    }
    if (!found) { *alpha = 1.0; for (int k = 0; k < nx; ++k) v[k] = 0.0; return LBMPC_ST_LP_INFEASIBLE; }
    *alpha = 1.0 - bx[0];
    for (int k = 0; k < nx; ++k) {
        if (y_out) y_out[k] = bx[1 + k];
        v[k] = bx[0] > 0.0 ? bx[1 + k] / bx[0] : 0.0;
    }
    return 0;
}

/* Box structure of W: each row is sigma * c * e_k (off-axis entries <= 1e-12 c), one + and one -
 * row per axis.  pr/qr: row index of the + / - row of axis k; ic: 1/c per row. */
int orc_box_structure(const double* WA, int nW, int nx, int* pr, int* qr, double* ic) {
    if (nW != 2 * nx) return LBMPC_ERR_UNSUPPORTED;
    for (int k = 0; k < nx; ++k) pr[k] = qr[k] = -1;
    for (int j = 0; j < nW; ++j) {
        int ax = 0;
        for (int k = 1; k < nx; ++k) if (fabs(CM(WA, nW, j, k)) > fabs(CM(WA, nW, j, ax))) ax = k;
        double a = CM(WA, nW, j, ax);
        if (a == 0.0) return LBMPC_ERR_UNSUPPORTED;
        for (int k = 0; k < nx; ++k)
            if (k != ax && fabs(CM(WA, nW, j, k)) > 1e-12 * fabs(a)) return LBMPC_ERR_UNSUPPORTED;
        ic[j] = 1.0 / fabs(a);
        if (a > 0) { if (pr[ax] >= 0) return LBMPC_ERR_UNSUPPORTED; pr[ax] = j; }
        else       { if (qr[ax] >= 0) return LBMPC_ERR_UNSUPPORTED; qr[ax] = j; }
    }
    for (int k = 0; k < nx; ++k) if (pr[k] < 0 || qr[k] < 0) return LBMPC_ERR_UNSUPPORTED;
    return 0;
}

/* Closed form of the same LP for an axis-aligned box W, with the canonical
 * tie-break LBMPC_TIE_CENTER: y_k = midpoint of the optimal interval of coordinate k.
 * This is the arithmetic the HIP kernels implement (same operation order). */
int orc_homothety_box(const int* pr, const int* qr, const double* ic, const double* Wb, int nx,
                      const double* m, double* alpha, double* v, uint32_t* active_mask) {
    double Bp[LBMPC_MAX_NX], Bq[LBMPC_MAX_NX], Mp[LBMPC_MAX_NX], Mq[LBMPC_MAX_NX], rho[LBMPC_MAX_NX];
    int infeas = 0;
    double a = 0.0;
    for (int k = 0; k < nx; ++k) {
        int p = pr[k], q = qr[k];
        Bp[k] = Wb[p] * ic[p]; Bq[k] = Wb[q] * ic[q];
        Mp[k] = m[p] * ic[p];  Mq[k] = m[q] * ic[q];
        if (m[p] > Wb[p] || m[q] > Wb[q]) infeas = 1;
        rho[k] = (Mp[k] + Mq[k]) / (Bp[k] + Bq[k]);
        if (rho[k] > a) a = rho[k];
    }
    if (infeas) {
        *alpha = 1.0;
        for (int k = 0; k < nx; ++k) v[k] = 0.0;
        if (active_mask) *active_mask = 0;
        return LBMPC_ST_LP_INFEASIBLE;
    }
    double beta = 1.0 - a;
    uint32_t mask = 0;
    for (int k = 0; k < nx; ++k) {
        double lo1 = Mp[k] - a * Bp[k], lo2 = -(beta * Bq[k]);
        double hi1 = beta * Bp[k], hi2 = a * Bq[k] - Mq[k];
        double lo = lo1 > lo2 ? lo1 : lo2;
        double hi = hi1 < hi2 ? hi1 : hi2;
        double y = 0.5 * (lo + hi);
        v[k] = beta > 0.0 ? y / beta : 0.0;
        if (a > 0.0 && rho[k] >= a) mask |= 1u << k;
    }
    *alpha = a;
    if (active_mask) *active_mask = mask;
    return 0;
}

/* GENERAL polytope W (InitialSetComputation.m:69-99 accepts any W): beta* by the vertex enumeration above, then the
 * canonical y of LBMPC_TIE_CENTER for a general W: the mean of the feasible basic solutions of the optimal face
 * Y* = {y : W_A y <= beta* W_b, -W_A y <= (1-beta*) W_b - m} over all nx-subsets of its 2 nW rows (subsets with a
 * singular matrix skipped; subsets in lexicographic order).  A convex combination of vertices of Y*, hence optimal;
 * for a box it is the midpoint of every coordinate's optimal interval, the rule of the closed form below.
 * active_mask: bit j set when row j of W binds on either side (|slack| <= 1e-9). */
int orc_homothety_general(const double* WA, const double* Wb, int nW, int nx, const double* m,
                          double* alpha, double* v, uint32_t* active_mask) {
    double y1[LBMPC_MAX_NX];
    int st = orc_homothety_lp_enum(WA, Wb, nW, nx, m, alpha, v, y1);
    if (active_mask) *active_mask = 0;
    if (st) return st;
    const double beta = 1.0 - *alpha;
    int nr = 2 * nW;
    double R[2 * LBMPC_MAX_NW * LBMPC_MAX_NX], rhs[2 * LBMPC_MAX_NW];
    for (int j = 0; j < nW; ++j) {
        for (int k = 0; k < nx; ++k) { R[j * nx + k] = CM(WA, nW, j, k); R[(nW + j) * nx + k] = -CM(WA, nW, j, k); }
        rhs[j] = beta * Wb[j];
        rhs[nW + j] = (Wb[j] - m[j]) - beta * Wb[j];
    }
    int idx[LBMPC_MAX_NX];
    for (int k = 0; k < nx; ++k) idx[k] = k;
    double sum[LBMPC_MAX_NX];
    for (int k = 0; k < nx; ++k) sum[k] = 0.0;
    int count = 0;
    for (;;) {
        double M[LBMPC_MAX_NX * LBMPC_MAX_NX], x[LBMPC_MAX_NX];
        for (int a = 0; a < nx; ++a) {
            for (int k = 0; k < nx; ++k) M[a * nx + k] = R[idx[a] * nx + k];
            x[a] = rhs[idx[a]];
        }
        if (solve_dense(nx, M, x)) {
            int ok = 1;
            for (int r = 0; r < nr && ok; ++r) {
                double acc = 0.0;
                for (int k = 0; k < nx; ++k) acc += R[r * nx + k] * x[k];
                if (acc > rhs[r] + 1e-11) ok = 0;
            }
            if (ok) { for (int k = 0; k < nx; ++k) sum[k] += x[k]; count++; }
        }
        int p = nx - 1;
        while (p >= 0 && idx[p] == nr - nx + p) --p;
        if (p < 0) break;
        ++idx[p];
        for (int k = p + 1; k < nx; ++k) idx[k] = idx[k - 1] + 1;
    }
    double y[LBMPC_MAX_NX];
    for (int k = 0; k < nx; ++k) y[k] = count > 0 ? sum[k] / (double)count : y1[k];
    for (int k = 0; k < nx; ++k) v[k] = beta > 0.0 ? y[k] / beta : 0.0;
    if (active_mask) {
        uint32_t mask = 0;
        for (int j = 0; j < nW; ++j) {
            double acc = 0.0;
            for (int k = 0; k < nx; ++k) acc += R[j * nx + k] * y[k];
            if (fabs(rhs[j] - acc) <= 1e-9 || fabs(rhs[nW + j] + acc) <= 1e-9) mask |= 1u << j;
        }
        *active_mask = mask;
    }
    return 0;
}

int orc_homothety_lp(const double* WA, const double* Wb, int nW, int nx, const double* m,
                     int64_t batch, int tiebreak, double* alpha, double* v, uint32_t* active_mask,
                     int* status) {
    int pr[LBMPC_MAX_NX], qr[LBMPC_MAX_NX];
    double ic[LBMPC_MAX_NW];
    if (tiebreak != LBMPC_TIE_CENTER) return LBMPC_ERR_ARG;
    int rc = orc_box_structure(WA, nW, nx, pr, qr, ic);
    if (rc == LBMPC_ERR_UNSUPPORTED) { /* not an axis-aligned box: the general solver */
        if (nW > LBMPC_MAX_NW || nx > LBMPC_MAX_NX) return LBMPC_ERR_DIM;
        for (int64_t b = 0; b < batch; ++b) {
            uint32_t mask;
            int st = orc_homothety_general(WA, Wb, nW, nx, m + (size_t)b * nW, alpha + b, v + (size_t)b * nx, &mask);
            if (active_mask) active_mask[b] = mask;
            if (status) status[b] = st;
        }
        return 0;
    }
    if (rc) return rc;
    for (int64_t b = 0; b < batch; ++b) {
        uint32_t mask;
        int st = orc_homothety_box(pr, qr, ic, Wb, nx, m + (size_t)b * nW, alpha + b, v + (size_t)b * nx, &mask);
        if (active_mask) active_mask[b] = mask;
        if (status) status[b] = st;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Subsystem 3: tube QP (UQRobustMPC.m:141-159, solve :68-91) and the closed loop
 * ---------------------------------------------------------------------------------------- */
typedef struct orc_ctx {
    int nx, nu, N, nc, Nu, nS, nW, nz, nH, m;
    double A[16], B[LBMPC_MAX_NX * LBMPC_MAX_NZ], K[LBMPC_MAX_NZ * LBMPC_MAX_NX];
    double T[16];                          /* (I - Phi)^-1 */
    double FcT[LBMPC_MAX_NC * LBMPC_MAX_NX]; /* (F + G K) T, nc x nx row-major */
    double hs[LBMPC_MAX_NC];
    double Ginv[LBMPC_MAX_NZ * LBMPC_MAX_NZ]; /* (2 blkdiag(Px,Pc))^-1, row-major */
    double* SA;   /* nS x nx row-major */
    double* Sb;   /* nS */
    double* H;    /* nH x nz row-major, nH = nc (Nu+1): rows of F_bar Psi^i */
    double WA[LBMPC_MAX_NW * LBMPC_MAX_NX]; /* row-major */
    double Wb[LBMPC_MAX_NW];
    int pr[LBMPC_MAX_NX], qr[LBMPC_MAX_NX];
    double ic[LBMPC_MAX_NW];
    int isbox;                      /* 0: W is a general polytope, the set update is orc_homothety_general */
    double WAcm[LBMPC_MAX_NW * LBMPC_MAX_NX]; /* W_A column-major (the LP routines take the caller's layout) */
    int hasE;                       /* orc_set_certificate: benchmark variant only (see step_one) */
    double ME[LBMPC_MAX_NX * LBMPC_MAX_NX];
} orc_ctx;

static int invert_spd_like(int n, const double* M /* row-major */, double* Minv) {
    /* Gauss-Jordan with partial pivoting */
    double a[LBMPC_MAX_NZ * 2 * LBMPC_MAX_NZ];
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) { a[i * 2 * n + j] = M[i * n + j]; a[i * 2 * n + n + j] = (i == j); }
    for (int c = 0; c < n; ++c) {
        int piv = c;
        for (int r = c + 1; r < n; ++r) if (fabs(a[r * 2 * n + c]) > fabs(a[piv * 2 * n + c])) piv = r;
        if (fabs(a[piv * 2 * n + c]) < 1e-300) return 0;
        if (piv != c) for (int k = 0; k < 2 * n; ++k) { double t = a[c * 2 * n + k]; a[c * 2 * n + k] = a[piv * 2 * n + k]; a[piv * 2 * n + k] = t; }
        double d = a[c * 2 * n + c];
        for (int k = 0; k < 2 * n; ++k) a[c * 2 * n + k] /= d;
        for (int r = 0; r < n; ++r) if (r != c) {
            double f = a[r * 2 * n + c];
            if (f != 0.0) for (int k = 0; k < 2 * n; ++k) a[r * 2 * n + k] -= f * a[c * 2 * n + k];
        }
    }
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) Minv[i * n + j] = a[i * 2 * n + n + j];
    return 1;
}

int orc_uqmpc_create(const lbmpc_params* p, orc_ctx** out) {
    *out = NULL;
    if (!p) return LBMPC_ERR_ARG;
    int nx = p->nx, nu = p->nu, N = p->N, nc = p->nc, nz = nx + N * nu;
    if (nx < 1 || nx > LBMPC_MAX_NX || nz > LBMPC_MAX_NZ || nc > LBMPC_MAX_NC || p->nW > LBMPC_MAX_NW) return LBMPC_ERR_DIM;
    orc_ctx* c = (orc_ctx*)calloc(1, sizeof(orc_ctx));
    c->nx = nx; c->nu = nu; c->N = N; c->nc = nc; c->Nu = p->Nu; c->nS = p->nS; c->nW = p->nW;
    c->nz = nz; c->nH = nc * (p->Nu + 1); c->m = c->nS + c->nH;
    memcpy(c->A, p->A, sizeof(double) * nx * nx);
    memcpy(c->B, p->B, sizeof(double) * nx * nu);
    memcpy(c->K, p->K, sizeof(double) * nu * nx);
    memcpy(c->hs, p->hs, sizeof(double) * nc);
    /* T = (I - Phi)^-1 */
    double ImP[16], Tr[16];
    for (int i = 0; i < nx; ++i) for (int j = 0; j < nx; ++j) ImP[i * nx + j] = (i == j ? 1.0 : 0.0) - CM(p->Phi, nx, i, j);
    if (!invert_spd_like(nx, ImP, Tr)) { free(c); return LBMPC_ERR_ARG; }
    for (int i = 0; i < nx; ++i) for (int j = 0; j < nx; ++j) CM(c->T, nx, i, j) = Tr[i * nx + j];
    /* FcT = (F + G K) T */
    for (int r = 0; r < nc; ++r) {
        double fc[LBMPC_MAX_NX];
        for (int k = 0; k < nx; ++k) {
            double acc = CM(p->F, nc, r, k);
            for (int j = 0; j < nu; ++j) acc = fma(CM(p->G, nc, r, j), CM(p->K, nu, j, k), acc);
            fc[k] = acc;
        }
        for (int k = 0; k < nx; ++k) {
            double acc = 0.0;
            for (int j = 0; j < nx; ++j) acc = fma(fc[j], CM(c->T, nx, j, k), acc);
            c->FcT[r * nx + k] = acc;
        }
    }
    /* Ginv = (2 blkdiag(Px, Pc))^-1 */
    double P2[LBMPC_MAX_NZ * LBMPC_MAX_NZ];
    memset(P2, 0, sizeof(P2));
    for (int i = 0; i < nx; ++i) for (int j = 0; j < nx; ++j) P2[i * nz + j] = 2.0 * CM(p->Px, nx, i, j);
    int nc_ = N * nu;
    for (int i = 0; i < nc_; ++i) for (int j = 0; j < nc_; ++j) P2[(nx + i) * nz + nx + j] = 2.0 * CM(p->Pc, nc_, i, j);
    if (!invert_spd_like(nz, P2, c->Ginv)) { free(c); return LBMPC_ERR_ARG; }
    /* symmetrise */
    for (int i = 0; i < nz; ++i) for (int j = i + 1; j < nz; ++j) { double s = 0.5 * (c->Ginv[i * nz + j] + c->Ginv[j * nz + i]); c->Ginv[i * nz + j] = c->Ginv[j * nz + i] = s; }
    c->SA = (double*)malloc(sizeof(double) * (size_t)c->nS * nx);
    c->Sb = (double*)malloc(sizeof(double) * (size_t)c->nS);
    for (int j = 0; j < c->nS; ++j) { for (int k = 0; k < nx; ++k) c->SA[j * nx + k] = CM(p->S_A, c->nS, j, k); c->Sb[j] = p->S_b[j]; }
    /* H rows: F_bar Psi^i, i = 0..Nu  (UQRobustMPC.m:153-155) */
    c->H = (double*)malloc(sizeof(double) * (size_t)c->nH * nz);
    double* M = (double*)malloc(sizeof(double) * nc * nz); /* row-major nc x nz, current F_bar Psi^i */
    double* M2 = (double*)malloc(sizeof(double) * nc * nz);
    for (int r = 0; r < nc; ++r) for (int k = 0; k < nz; ++k) M[r * nz + k] = CM(p->F_bar, nc, r, k);
    for (int i = 0; i <= p->Nu; ++i) {
        memcpy(c->H + (size_t)i * nc * nz, M, sizeof(double) * nc * nz);
        for (int r = 0; r < nc; ++r) for (int k = 0; k < nz; ++k) {
            double acc = 0.0;
            for (int j = 0; j < nz; ++j) acc = fma(M[r * nz + j], CM(p->Psi, nz, j, k), acc);
            M2[r * nz + k] = acc;
        }
        memcpy(M, M2, sizeof(double) * nc * nz);
    }
    free(M); free(M2);
    for (int j = 0; j < c->nW; ++j) { for (int k = 0; k < nx; ++k) c->WA[j * nx + k] = CM(p->W_A, c->nW, j, k); c->Wb[j] = p->W_b[j]; }
    memcpy(c->WAcm, p->W_A, sizeof(double) * (size_t)c->nW * nx);
    int rc = orc_box_structure(p->W_A, c->nW, nx, c->pr, c->qr, c->ic);
    c->isbox = rc == 0;
    /* UQRobustMPC.m:124-139 takes any polytope W: vertex enumeration then (nx + 1 <= nW) */
    if (rc == LBMPC_ERR_UNSUPPORTED && c->nW >= nx + 1) rc = 0;
    if (rc) { free(c->SA); free(c->Sb); free(c->H); free(c); return rc; }
    *out = c;
    return 0;
}

/* Benchmark variant of the closed loop (bench.py: cpu_baseline.with_certificate), off unless this is called: ME is
 * the matrix of an ellipsoid {x : x' ME x <= 1} INSIDE S (the caller computes it; the HIP library uses the same
 * kind of certificate, uqmpc.hip inscribed_ellipsoid).  Then q' ME q <= (alpha (1 - 1e-6))^2 proves that no S row is
 * violated at z = 0 (Cauchy-Schwarz), and with min(1 - h_k) >= 0 no H row is, so qp_solve would return z = 0 with an
 * empty active set after one scan of all rows: step_one skips the solve.  The parity tests never set it: the
 * oracle they check against evaluates every row.  ME = NULL switches it off. */
int orc_set_certificate(orc_ctx* c, const double* ME_rowmajor) {
    if (!c) return LBMPC_ERR_ARG;
    c->hasE = ME_rowmajor != NULL;
    for (int i = 0; i < c->nx * c->nx; ++i) c->ME[i] = ME_rowmajor ? ME_rowmajor[i] : 0.0;
    return 0;
}

void orc_destroy(orc_ctx* c) {
    if (!c) return;
    free(c->SA); free(c->Sb); free(c->H); free(c);
}

/* Export the derived constant blocks so that tests can compare them with the product's. */
int orc_ctx_export(const orc_ctx* c, double* H_rowmajor, double* Ginv_rowmajor, double* T_colmajor, double* FcT_rowmajor) {
    if (H_rowmajor) memcpy(H_rowmajor, c->H, sizeof(double) * (size_t)c->nH * c->nz);
    if (Ginv_rowmajor) memcpy(Ginv_rowmajor, c->Ginv, sizeof(double) * c->nz * c->nz);
    if (T_colmajor) memcpy(T_colmajor, c->T, sizeof(double) * c->nx * c->nx);
    if (FcT_rowmajor) memcpy(FcT_rowmajor, c->FcT, sizeof(double) * c->nc * c->nx);
    return 0;
}

/* slack of QP row j at z.  S rows: alpha Sb_j + SA_j.(q + s0), q = (1-alpha) T v - x
 * (i.e. S_hat_b - S_A (x - s0), UQRobustMPC.m:152); H rows: (1 - h_r) - H_row.z (:154) */
static double row_slack(const orc_ctx* c, int j, const double* e /* q + s0 */, double alpha,
                        const double* omh /* 1 - hk */, const double* z) {
    if (j < c->nS) {
        double acc = alpha * c->Sb[j];
        for (int k = 0; k < c->nx; ++k) acc = fma(c->SA[j * c->nx + k], e[k], acc);
        return acc;
    }
    int hrow = j - c->nS;
    double acc = omh[hrow % c->nc];
    const double* h = c->H + (size_t)hrow * c->nz;
    for (int k = 0; k < c->nz; ++k) acc = fma(-h[k], z[k], acc);
    return acc;
}

static void row_coeffs(const orc_ctx* c, int j, double* cj) { /* c_j of c_j.z <= d_j */
    if (j < c->nS) {
        for (int k = 0; k < c->nz; ++k) cj[k] = 0.0;
        for (int k = 0; k < c->nx; ++k) cj[k] = -c->SA[j * c->nx + k];
    } else {
        memcpy(cj, c->H + (size_t)(j - c->nS) * c->nz, sizeof(double) * c->nz);
    }
}

/* Goldfarb-Idnani dual active-set method (Math. Prog. 27, 1983) for
 *   min z' P z  s.t.  rows above,   P = blkdiag(Px, Pc)   (UQRobustMPC.m:149-155)
 * started from the unconstrained minimiser z = 0 with an empty active set.  Replaces IPOPT. */
/* Pivot rule: candidate rows first.  `hint` (nhint rows, may be NULL) is the final active set of the previous
 * closed-loop step of the same trial; each H row of it also proposes its one-step-earlier twin (row - nc: the
 * prediction step index shifts by one per closed-loop step).  While any candidate that is not active is violated
 * the most violated one (ties: lowest row) is the pivot and no full pass over the rows is needed; otherwise the
 * most violated row of all (ties: lowest row).  Any rule that picks violated rows reaches the same, unique
 * minimiser; this one needs fewer full passes along a trajectory. */
#ifdef ORC_STATS
/* diagnostic build only (tools/qp_stats.py): per-solve histograms, single thread */
long long orc_stats_hist[8][64]; /* [0] final nq, [1] max nq, [2] iterations, [3] full scans, [4] drops, [5] final H rows, [6] H rows scanned? */
void orc_stats_reset(void) { memset(orc_stats_hist, 0, sizeof(orc_stats_hist)); }
#define ST_BUMP(h, v) orc_stats_hist[h][(v) < 63 ? (v) : 63]++
#endif
static int qp_solve(const orc_ctx* c, const double* q, double alpha, const double* omh, double* z,
                    int* act, double* lam, int* nact_out, int* iters_out, const int* hint, int nhint) {
    int nz = c->nz, nx = c->nx, m = c->m, nq = 0, iters = 0, status = 0;
#ifdef ORC_STATS
    int st_maxnq = 0, st_scans = 0, st_drops = 0;
#endif
    double CA[LBMPC_MAX_NZ][LBMPC_MAX_NZ]; /* active rows */
    double Wm[LBMPC_MAX_NZ][LBMPC_MAX_NZ]; /* Wm[a] = Ginv c_a */
    double u[LBMPC_MAX_NZ + 1];
    for (int k = 0; k < nz; ++k) z[k] = 0.0;
    for (;;) {
        /* Step 1: most violated row */
        double e[LBMPC_MAX_NX];
        for (int k = 0; k < nx; ++k) e[k] = q[k] + z[k];
        int p = -1;
        double sp = -QP_TOL_VIOL;
        for (int hh = 0; hh < nhint; ++hh)
            for (int tw = 0; tw < 2; ++tw) {
                int j = hint[hh];
                if (tw) { if (j < c->nS + c->nc) continue; j -= c->nc; }
                int isact = 0;
                for (int a = 0; a < nq; ++a) if (act[a] == j) isact = 1;
                if (isact) continue;
                double s = row_slack(c, j, e, alpha, omh, z);
                if (s < sp || (s == sp && p >= 0 && j < p)) { sp = s; p = j; }
            }
#ifdef ORC_STATS
        if (p < 0) st_scans++;
#endif
        if (p < 0)
        for (int j = 0; j < m; ++j) {
            int isact = 0;
            for (int a = 0; a < nq; ++a) if (act[a] == j) isact = 1;
            if (isact) continue;
            double s = row_slack(c, j, e, alpha, omh, z);
            if (s < sp) { sp = s; p = j; }
        }
        if (p < 0) break;
        double cp[LBMPC_MAX_NZ], g[LBMPC_MAX_NZ];
        row_coeffs(c, p, cp);
        for (int i = 0; i < nz; ++i) {
            double acc = 0.0;
            for (int k = 0; k < nz; ++k) acc = fma(c->Ginv[i * nz + k], cp[k], acc);
            g[i] = acc;
        }
        double cg = 0.0;
        for (int k = 0; k < nz; ++k) cg = fma(cp[k], g[k], cg);
        u[nq] = 0.0;
        /* Step 2 */
        for (;;) {
            if (++iters > QP_MAX_ITERS) { status = LBMPC_ST_ITER_CAP; goto done; }
            double r[LBMPC_MAX_NZ], rhs[LBMPC_MAX_NZ], L[LBMPC_MAX_NZ][LBMPC_MAX_NZ];
            double delta = cg;
            if (nq > 0) {
                /* M = C_A Ginv C_A', Cholesky, r = M^-1 C_A g */
                for (int a = 0; a < nq; ++a) {
                    double acc = 0.0;
                    for (int k = 0; k < nz; ++k) acc = fma(CA[a][k], g[k], acc);
                    rhs[a] = acc;
                }
                /* divisions by the diagonal are multiplications by its reciprocal (shared with the kernels) */
                double invd[LBMPC_MAX_NZ];
                for (int a = 0; a < nq; ++a)
                    for (int b = 0; b <= a; ++b) {
                        double acc = 0.0;
                        for (int k = 0; k < nz; ++k) acc = fma(CA[b][k], Wm[a][k], acc); /* M[a][b] = c_b . (Ginv c_a) */
                        for (int k = 0; k < b; ++k) acc = fma(-L[a][k], L[b][k], acc);
                        if (a == b) { L[a][a] = sqrt(acc > 0.0 ? acc : 0.0); invd[a] = 1.0 / L[a][a]; }
                        else L[a][b] = acc * invd[b];
                    }
                for (int a = 0; a < nq; ++a) { /* forward */
                    double acc = rhs[a];
                    for (int k = 0; k < a; ++k) acc = fma(-L[a][k], r[k], acc);
                    r[a] = acc * invd[a];
                }
                for (int a = nq - 1; a >= 0; --a) { /* backward (terms applied in descending k: a column sweep) */
                    double acc = r[a];
                    for (int k = nq - 1; k > a; --k) acc = fma(-L[k][a], r[k], acc);
                    r[a] = acc * invd[a];
                }
                for (int a = 0; a < nq; ++a) delta = fma(-rhs[a], r[a], delta);
            }
            /* primal direction zd = -(g - Wm r) */
            double zd[LBMPC_MAX_NZ];
            for (int k = 0; k < nz; ++k) {
                double acc = g[k];
                for (int a = 0; a < nq; ++a) acc = fma(-Wm[a][k], r[a], acc);
                zd[k] = -acc;
            }
            int dependent = (nq >= nz) || !(delta > QP_TOL_DEP * cg);
            /* t1: largest dual step keeping u >= 0 */
            double t1 = INFINITY, rmax = 0.0;
            int kdrop = -1;
            for (int a = 0; a < nq; ++a) if (fabs(r[a]) > rmax) rmax = fabs(r[a]);
            for (int a = 0; a < nq; ++a)
                if (r[a] > QP_TOL_R * rmax) {
                    double t = u[a] / r[a];
                    if (t < t1) { t1 = t; kdrop = a; }
                }
            double t2 = dependent ? INFINITY : -sp / delta;
            double t = t1 < t2 ? t1 : t2;
#ifdef ORC_TRACE
            fprintf(stderr, "CPU it=%d p=%d sp=%.17g nq=%d cg=%.17g delta=%.17g t1=%.17g t2=%.17g kdrop=%d dep=%d r0=%.17g u0=%.17g\n", iters, p, sp, nq, cg, delta, t1, t2, kdrop, dependent, nq ? r[0] : 0.0, u[0]);
#endif
            if (t == INFINITY) { status = LBMPC_ST_QP_INFEASIBLE; goto done; }
            for (int a = 0; a < nq; ++a) u[a] = fma(-t, r[a], u[a]);
            u[nq] += t;
            if (!dependent) {
                for (int k = 0; k < nz; ++k) z[k] = fma(t, zd[k], z[k]);
                sp = fma(t, delta, sp);
            }
            if (!dependent && t2 <= t1) { /* full step: add p */
                act[nq] = p;
                memcpy(CA[nq], cp, sizeof(double) * nz);
                memcpy(Wm[nq], g, sizeof(double) * nz);
                nq++;
#ifdef ORC_STATS
                if (nq > st_maxnq) st_maxnq = nq;
#endif
                break;
            }
#ifdef ORC_STATS
            st_drops++;
#endif
            /* drop constraint kdrop, stay in step 2 */
            for (int a = kdrop; a < nq - 1; ++a) {
                act[a] = act[a + 1]; u[a] = u[a + 1];
                memcpy(CA[a], CA[a + 1], sizeof(double) * nz);
                memcpy(Wm[a], Wm[a + 1], sizeof(double) * nz);
            }
            u[nq - 1] = u[nq];
            nq--;
        }
    }
done:
#ifdef ORC_STATS
    if (st_scans > 1 || nq > 0) { /* a QP the kernels would solve (not the trivial z = 0 confirmation) */
        int nh = 0;
        for (int a = 0; a < nq; ++a) nh += act[a] >= c->nS;
        ST_BUMP(0, nq); ST_BUMP(1, st_maxnq); ST_BUMP(2, iters); ST_BUMP(3, st_scans); ST_BUMP(4, st_drops); ST_BUMP(5, nh);
    }
#endif
    for (int a = 0; a < nq; ++a) lam[a] = u[a];
    *nact_out = nq;
    *iters_out = iters;
    return status;
}

/* One UQ-RMPC step, UQRobustMPC.m:68-91 */
static int step_one(const orc_ctx* c, int uq_update, const double* x, double alpha_prev,
                    const double* v_prev, const double* w_new, double* s0, double* cc, double* u,
                    double* alpha_out, double* v_out, double* hk, double* Sb_hat, double* Wb_hat,
                    int* n_active, int* active_set, int* qp_iters, int* hint /* in: previous, out: this step's */,
                    int* nhint) {
    int nx = c->nx, nW = c->nW, nz = c->nz, status = 0;
    double alpha = alpha_prev, v[LBMPC_MAX_NX], bh[LBMPC_MAX_NW];
    memcpy(v, v_prev, sizeof(double) * nx);
    /* b_hat_prev = alpha W_b + (1-alpha) W_A v   (H-rep of W_hat, :75) */
    for (int j = 0; j < nW; ++j) {
        double t = 0.0;
        for (int k = 0; k < nx; ++k) t = fma(c->WA[j * nx + k], v[k], t);
        bh[j] = alpha * c->Wb[j] + (1.0 - alpha) * t;
    }
    if (uq_update) { /* LPforUQ :124-139 == homothety LP on m = max(b_hat_prev, W_A w_new) */
        double mm[LBMPC_MAX_NW];
        int outside = 0;
        for (int j = 0; j < nW; ++j) {
            double hw = 0.0;
            for (int k = 0; k < nx; ++k) hw = fma(c->WA[j * nx + k], w_new[k], hw);
            mm[j] = bh[j];
            if (hw > bh[j]) { mm[j] = hw; outside = 1; }
        }
        if (outside) {
            int st = c->isbox ? orc_homothety_box(c->pr, c->qr, c->ic, c->Wb, nx, mm, &alpha, v, NULL)
                              : orc_homothety_general(c->WAcm, c->Wb, nW, nx, mm, &alpha, v, NULL);
            if (st > status) status = st;
            for (int j = 0; j < nW; ++j) {
                double t = 0.0;
                for (int k = 0; k < nx; ++k) t = fma(c->WA[j * nx + k], v[k], t);
                bh[j] = alpha * c->Wb[j] + (1.0 - alpha) * t;
            }
        }
    }
    double beta = 1.0 - alpha;
    double Tv[LBMPC_MAX_NX], q[LBMPC_MAX_NX], omh[LBMPC_MAX_NC], hh[LBMPC_MAX_NC];
    mat_vec(c->T, nx, nx, v, Tv);
    /* h_k = alpha hs + (1-alpha) F_Com T v   (closed form of the nc support LPs :80-84) */
    for (int r = 0; r < c->nc; ++r) {
        double gsum = 0.0;
        for (int k = 0; k < nx; ++k) gsum = fma(c->FcT[r * nx + k], v[k], gsum);
        hh[r] = alpha * c->hs[r] + beta * gsum;
        omh[r] = 1.0 - hh[r];
    }
    for (int k = 0; k < nx; ++k) q[k] = beta * Tv[k] - x[k];
    double z[LBMPC_MAX_NZ], lam[LBMPC_MAX_NZ];
    int act[LBMPC_MAX_NZ], nact = 0, iters = 0;
    int st = 0, certified = 0;
    if (c->hasE) { /* benchmark variant (orc_set_certificate): z = 0 is optimal when no row is violated there */
        double acc = 0.0, mo = omh[0];
        for (int a = 0; a < nx; ++a) for (int b = 0; b < nx; ++b) acc += q[a] * c->ME[a * nx + b] * q[b];
        for (int r = 1; r < c->nc; ++r) if (omh[r] < mo) mo = omh[r];
        const double rad = alpha * (1.0 - 1e-6);
        certified = alpha > 0.0 && acc <= rad * rad && mo >= 0.0;
    }
    if (certified) for (int k = 0; k < nz; ++k) z[k] = 0.0;   /* what qp_solve returns after one scan: nact = iters = 0 */
    else st = qp_solve(c, q, alpha, omh, z, act, lam, &nact, &iters, hint, nhint ? *nhint : 0);
    if (hint && nhint) { for (int a = 0; a < nact; ++a) hint[a] = act[a]; *nhint = nact; }
    if (st > status) status = st;
    if (s0) memcpy(s0, z, sizeof(double) * nx);
    if (cc) memcpy(cc, z + nx, sizeof(double) * (nz - nx));
    if (u) for (int j = 0; j < c->nu; ++j) { /* u = K x + c(1:nu)  :90 */
        double acc = z[nx + j];
        for (int k = 0; k < nx; ++k) acc = fma(CM(c->K, c->nu, j, k), x[k], acc);
        u[j] = acc;
    }
    if (alpha_out) *alpha_out = alpha;
    if (v_out) memcpy(v_out, v, sizeof(double) * nx);
    if (hk) memcpy(hk, hh, sizeof(double) * c->nc);
    if (Wb_hat) memcpy(Wb_hat, bh, sizeof(double) * nW);
    if (Sb_hat) for (int j = 0; j < c->nS; ++j) { /* S_hat = alpha S + (1-alpha) T v  :76 */
        double t = 0.0;
        for (int k = 0; k < nx; ++k) t = fma(c->SA[j * nx + k], Tv[k], t);
        Sb_hat[j] = alpha * c->Sb[j] + beta * t;
    }
    if (n_active) *n_active = nact;
    if (active_set) {
        /* ascending order */
        for (int a = 0; a < LBMPC_MAX_NZ; ++a) active_set[a] = -1;
        for (int a = 0; a < nact; ++a) {
            int pos = a;
            while (pos > 0 && active_set[pos - 1] > act[a]) { active_set[pos] = active_set[pos - 1]; --pos; }
            active_set[pos] = act[a];
        }
    }
    if (qp_iters) *qp_iters = iters;
    return status;
}

int orc_uqmpc_step(orc_ctx* c, int64_t batch, int uq_update, const double* x, const double* alpha_prev,
                   const double* v_prev, const double* w_new, double* s0, double* cc, double* u,
                   double* alpha, double* v, double* hk, double* Sb_hat, double* Wb_hat, int* status,
                   int* n_active, int* active_set, int* qp_iters) {
    if (!c || !x || !alpha_prev || !v_prev || (uq_update && !w_new)) return LBMPC_ERR_ARG;
    int nx = c->nx, ncv = c->nz - c->nx;
    for (int64_t b = 0; b < batch; ++b) {
        int st = step_one(c, uq_update, x + b * nx, alpha_prev[b], v_prev + b * nx, w_new ? w_new + b * nx : NULL,
                          s0 ? s0 + b * nx : NULL, cc ? cc + b * ncv : NULL, u ? u + b * c->nu : NULL,
                          alpha ? alpha + b : NULL, v ? v + b * nx : NULL, hk ? hk + b * c->nc : NULL,
                          Sb_hat ? Sb_hat + b * c->nS : NULL, Wb_hat ? Wb_hat + b * c->nW : NULL,
                          n_active ? n_active + b : NULL, active_set ? active_set + b * LBMPC_MAX_NZ : NULL,
                          qp_iters ? qp_iters + b : NULL, NULL, NULL);
        if (status) status[b] = st;
    }
    return 0;
}

/* Closed-loop Monte-Carlo: Case_3...:57-88, Case_4...:32-66, Case_5...:55-98 in relative
 * coordinates (SURVEY.md hard part 5). */
static void mc_trial(const orc_ctx* c, const lbmpc_sampler* smp, const lbmpc_mc_cfg* cfg, lbmpc_mc_out* out, int64_t t) {
    int nx = c->nx, nu = c->nu, nW = c->nW, K = cfg->steps;
    {
        uint64_t tid = (uint64_t)(cfg->trial0 + t);
        int status = 0, infeasible = 0, iters_total = 0;
        int hint[LBMPC_MAX_NZ], nhint = 0; /* active set of the previous step's QP (pivot candidates) */
        /* initial set learning: InitialSetComputation.solve (:48-59) */
        double m[LBMPC_MAX_NW], alpha, v[LBMPC_MAX_NX];
        for (int j = 0; j < nW; ++j) m[j] = -INFINITY;
        for (int i = 0; i < cfg->card_I0; ++i) {
            double w[LBMPC_MAX_NX];
            if (sample_w_I0(smp, cfg->seed, tid, (uint32_t)i, 100, w)) status = LBMPC_ST_SAMPLER_CAP;
            for (int j = 0; j < nW; ++j) {
                double acc = 0.0;
                for (int k = 0; k < nx; ++k) acc = fma(c->WA[j * nx + k], w[k], acc);
                if (acc > m[j]) m[j] = acc;
            }
        }
        int st = c->isbox ? orc_homothety_box(c->pr, c->qr, c->ic, c->Wb, nx, m, &alpha, v, NULL)
                          : orc_homothety_general(c->WAcm, c->Wb, nW, nx, m, &alpha, v, NULL);
        if (st > status) status = st;
        if (out->alpha0) out->alpha0[t] = alpha;
        double x[LBMPC_MAX_NX], xprev[LBMPC_MAX_NX], uprev[LBMPC_MAX_NZ];
        memcpy(x, cfg->x0, sizeof(double) * nx);
        int rec = t < cfg->traj_trials;
        for (int k = 0; k < nx; ++k) if (x[k] > cfg->distance_error || x[k] < -cfg->distance_error) infeasible = 1;
        if (rec && out->traj_x) memcpy(out->traj_x + ((size_t)t * (K + 1)) * nx, x, sizeof(double) * nx);
        if (rec && out->traj_alpha) out->traj_alpha[(size_t)t * (K + 1)] = alpha;
        for (int k = 0; k < K; ++k) {
            double xi_lv[LBMPC_MAX_NX], xi_ev[LBMPC_MAX_NX], u_lv[LBMPC_MAX_NZ], w_new[LBMPC_MAX_NX];
            /* LVModeling first (Case_3...:58; ModelingCar.m:42-43) */
            if (sample_polytope(smp->LV_A, smp->LV_b, smp->nLV, smp->LV_lo, smp->LV_hi, smp->LV_U, nx, cfg->seed, tid, (uint32_t)k, P_ST_LV, cfg->sampler_max_attempts, xi_lv)) status = status > LBMPC_ST_SAMPLER_CAP ? status : LBMPC_ST_SAMPLER_CAP;
            sample_u_lv(smp, cfg->seed, tid, (uint32_t)k, P_ST_U, u_lv);
            if (k == 0) for (int i = 0; i < nx; ++i) w_new[i] = 0.0; /* Case_3...:62-65 */
            else {
                double Ax[LBMPC_MAX_NX], Bu[LBMPC_MAX_NX];
                mat_vec(c->A, nx, nx, xprev, Ax);
                mat_vec(c->B, nx, nu, uprev, Bu);
                for (int i = 0; i < nx; ++i) w_new[i] = (x[i] - Ax[i]) - Bu[i]; /* :67 */
            }
            double s0[LBMPC_MAX_NX], u[LBMPC_MAX_NZ], a2, v2[LBMPC_MAX_NX];
            int it = 0;
            st = step_one(c, 1, x, alpha, v, w_new, s0, NULL, u, &a2, v2, NULL, NULL, NULL, NULL, NULL, &it, hint, &nhint);
            if (st > status) status = st;
            iters_total += it;
            alpha = a2; memcpy(v, v2, sizeof(double) * nx);
            /* EVModeling (ModelingCar.m:36-37) */
            if (sample_polytope(smp->EV_A, smp->EV_b, smp->nEV, smp->EV_lo, smp->EV_hi, smp->EV_U, nx, cfg->seed, tid, (uint32_t)k, P_ST_EV, cfg->sampler_max_attempts, xi_ev)) status = status > LBMPC_ST_SAMPLER_CAP ? status : LBMPC_ST_SAMPLER_CAP;
            double Ax[LBMPC_MAX_NX], Bu[LBMPC_MAX_NX], Bul[LBMPC_MAX_NX], xn[LBMPC_MAX_NX];
            mat_vec(c->A, nx, nx, x, Ax);
            mat_vec(c->B, nx, nu, u, Bu);
            mat_vec(c->B, nx, nu, u_lv, Bul);
            /* the step disturbance in the form the information set uses (InitialSetComputation.m:51-52):
             * w = (xi_EV - xi_LV) - B u_LV;  relative dynamics x+ = (A x + B u) + w */
            for (int i = 0; i < nx; ++i) { double wd = (xi_ev[i] - xi_lv[i]) - Bul[i]; xn[i] = (Ax[i] + Bu[i]) + wd; }
            memcpy(xprev, x, sizeof(double) * nx);
            memcpy(uprev, u, sizeof(double) * nu);
            memcpy(x, xn, sizeof(double) * nx);
            /* violation test, Case_5...:90-97 */
            for (int j = 0; j < nu; ++j) if (u[j] > cfg->acc_bound || u[j] < -cfg->acc_bound) infeasible = 1;
            for (int i = 0; i < nx; ++i) if (x[i] > cfg->distance_error || x[i] < -cfg->distance_error) infeasible = 1;
            if (rec) {
                if (out->traj_x) memcpy(out->traj_x + ((size_t)t * (K + 1) + k + 1) * nx, x, sizeof(double) * nx);
                if (out->traj_u) memcpy(out->traj_u + ((size_t)t * K + k) * nu, u, sizeof(double) * nu);
                if (out->traj_alpha) out->traj_alpha[(size_t)t * (K + 1) + k + 1] = alpha;
                if (out->traj_s0) memcpy(out->traj_s0 + ((size_t)t * K + k) * nx, s0, sizeof(double) * nx);
            }
        }
        if (out->infeasible) out->infeasible[t] = infeasible;
        if (out->status) out->status[t] = status;
        if (out->qp_iters) out->qp_iters[t] = iters_total;
        if (out->alpha_final) out->alpha_final[t] = alpha;
        if (out->v_final) memcpy(out->v_final + (size_t)t * nx, v, sizeof(double) * nx);
        if (out->x_final) memcpy(out->x_final + (size_t)t * nx, x, sizeof(double) * nx);
    }
}

int orc_mc_run(orc_ctx* c, const lbmpc_sampler* smp, const lbmpc_mc_cfg* cfg, lbmpc_mc_out* out) {
    if (!c || !smp || !cfg || !out || !cfg->x0) return LBMPC_ERR_ARG;
    for (int64_t t = 0; t < cfg->trials; ++t) mc_trial(c, smp, cfg, out, t);
    return 0;
}

/* The same loop with the independent trials spread over `threads` host threads (OpenMP): the all-cores leg of
 * bench.py's cpu_baseline.  Results are those of orc_mc_run (no shared state between trials). */
int orc_mc_run_omp(orc_ctx* c, const lbmpc_sampler* smp, const lbmpc_mc_cfg* cfg, lbmpc_mc_out* out, int threads) {
    if (!c || !smp || !cfg || !out || !cfg->x0 || threads < 1) return LBMPC_ERR_ARG;
#pragma omp parallel for schedule(dynamic, 8) num_threads(threads)
    for (int64_t t = 0; t < cfg->trials; ++t) mc_trial(c, smp, cfg, out, t);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Support function of a polytope in H-rep by an exact LP: the LPs of
 * Functions_general/ComputeFeasibleRegion.m -- Com_hs (:447-503: max F_com e s.t. S_A e <= S_b),
 * Com_z (:505-519: max M_i z s.t. M_before z <= 1 - hs, the nu_s search :66-116) and the support of the
 * nominal feasible region F_Ns = proj_s {z : F_bar Psi^i z <= 1 - h, i = 0..nu} (the set the 5000
 * YALMIP projections of :153-171 sample) -- which the reference hands to IPOPT.
 * Primal active-set method from the interior point z = 0 (needs b > 0): move along the projection p of c
 * onto the null space of the working rows until a row blocks (ratio test, lowest row on ties), add it;
 * when p vanishes the multipliers of c = A_W' lam decide: all >= 0 optimal, else drop the LOWEST ROW with a
 * negative multiplier (Bland).  Rows are normalised to unit 2-norm by the caller.  Dense Cholesky of
 * A_W A_W' (k <= n <= 16) from scratch at every iteration.
 * ---------------------------------------------------------------------------------------- */
#define LP_MAX_ITERS 400


static int lp_support_one(const double* A /* m x n row-major, unit rows */, const double* b, int m, int n,
                          const double* c, double* value, double* zout, int* iters_out, int skip /* row ignored, -1: none */) {
    double z[LBMPC_MAX_NZ], p[LBMPC_MAX_NZ], lam[LBMPC_MAX_NZ], L[LBMPC_MAX_NZ][LBMPC_MAX_NZ];
    int W[LBMPC_MAX_NZ], k = 0, iters = 0, status = 0;
    double cn = 0.0;
    for (int i = 0; i < n; ++i) { z[i] = 0.0; if (fabs(c[i]) > cn) cn = fabs(c[i]); }
    if (cn == 0.0) { *value = 0.0; if (zout) for (int i = 0; i < n; ++i) zout[i] = 0.0; if (iters_out) *iters_out = 0; return 0; }
    for (;;) {
        if (++iters > LP_MAX_ITERS) { status = LBMPC_ST_ITER_CAP; break; }
        /* lam = (A_W A_W')^-1 A_W c ; p = c - A_W' lam */
        for (int a = 0; a < k; ++a) {
            for (int bq = 0; bq <= a; ++bq) {
                double acc = 0.0;
                for (int i = 0; i < n; ++i) acc = fma(A[(size_t)W[a] * n + i], A[(size_t)W[bq] * n + i], acc);
                for (int q = 0; q < bq; ++q) acc = fma(-L[a][q], L[bq][q], acc);
                if (a == bq) L[a][a] = sqrt(acc > 0.0 ? acc : 0.0);
                else L[a][bq] = acc / L[bq][bq];
            }
            double r = 0.0;
            for (int i = 0; i < n; ++i) r = fma(A[(size_t)W[a] * n + i], c[i], r);
            lam[a] = r;
        }
        for (int a = 0; a < k; ++a) { double acc = lam[a]; for (int q = 0; q < a; ++q) acc = fma(-L[a][q], lam[q], acc); lam[a] = acc / L[a][a]; }
        for (int a = k - 1; a >= 0; --a) { double acc = lam[a]; for (int q = k - 1; q > a; --q) acc = fma(-L[q][a], lam[q], acc); lam[a] = acc / L[a][a]; }
        double pn = 0.0;
        for (int i = 0; i < n; ++i) {
            double acc = c[i];
            for (int a = 0; a < k; ++a) acc = fma(-A[(size_t)W[a] * n + i], lam[a], acc);
            p[i] = acc;
            if (fabs(acc) > pn) pn = fabs(acc);
        }
        if (k == n || pn <= 1e-11 * cn) {
            /* stationary on the working set: Bland's rule on the multipliers */
            int drop = -1;
            for (int a = 0; a < k; ++a)
                if (lam[a] < -1e-10 * cn && (drop < 0 || W[a] < W[drop])) drop = a;
            if (drop < 0) break; /* optimal */
            for (int a = drop; a < k - 1; ++a) W[a] = W[a + 1];
            --k;
            continue;
        }
        double best = INFINITY;
        int jb = -1;
        for (int j = 0; j < m; ++j) {
            int inW = 0;
            for (int a = 0; a < k; ++a) if (W[a] == j) inW = 1;
            if (inW || j == skip) continue;
            double ap = 0.0, az = 0.0;
            for (int i = 0; i < n; ++i) { ap = fma(A[(size_t)j * n + i], p[i], ap); az = fma(A[(size_t)j * n + i], z[i], az); }
            if (ap > 1e-12 * pn) {
                double sl = b[j] - az;
                if (sl < 0.0) sl = 0.0;
                double r = sl / ap;
                if (r < best) { best = r; jb = j; }
            }
        }
        if (jb < 0) { status = LBMPC_ST_LP_UNBOUNDED; break; }
        for (int i = 0; i < n; ++i) z[i] = fma(best, p[i], z[i]);
        W[k++] = jb;
    }
    double val = 0.0;
    for (int i = 0; i < n; ++i) val = fma(c[i], z[i], val);
    *value = val;
    if (zout) for (int i = 0; i < n; ++i) zout[i] = z[i];
    if (iters_out) *iters_out = iters;
    return status;
}

/* h_out[i + nd*s] = max { C[:,i]' z : A z <= b[:,s] } for every direction i and right-hand side s.
 * A: m x n column-major (rows need not be normalised; they are here), b: m x nb, C: n x nd. */
int orc_polytope_support(const double* A, int m, int n, const double* b, int64_t nb, const double* Cd, int64_t nd,
                         double* h_out, double* z_out, int* status_out) {
    if (!A || !b || !Cd || !h_out || m < 1 || n < 1 || n > LBMPC_MAX_NZ || nb < 0 || nd < 0) return LBMPC_ERR_ARG;
    double* An = (double*)malloc(sizeof(double) * (size_t)m * n);
    double* sc = (double*)malloc(sizeof(double) * (size_t)m);
    double* bn = (double*)malloc(sizeof(double) * (size_t)m);
    for (int j = 0; j < m; ++j) {
        double s2 = 0.0;
        for (int i = 0; i < n; ++i) s2 = fma(CM(A, m, j, i), CM(A, m, j, i), s2);
        double nr = sqrt(s2);
        sc[j] = nr > 0.0 ? 1.0 / nr : 0.0;
        for (int i = 0; i < n; ++i) An[(size_t)j * n + i] = CM(A, m, j, i) * sc[j];
    }
    int rc = 0;
    for (int64_t s = 0; s < nb && !rc; ++s) {
        for (int j = 0; j < m; ++j) {
            double bj = b[(size_t)s * m + j];
            bn[j] = sc[j] > 0.0 ? bj * sc[j] : INFINITY; /* a zero row never blocks */
            if (!(bj > 0.0)) rc = LBMPC_ERR_ARG;         /* z = 0 must be strictly feasible */
        }
        if (rc) break;
        for (int64_t i = 0; i < nd; ++i) {
            double c[LBMPC_MAX_NZ], val;
            for (int q = 0; q < n; ++q) c[q] = Cd[(size_t)i * n + q];
            int st = lp_support_one(An, bn, m, n, c, &val, z_out ? z_out + ((size_t)s * nd + i) * n : NULL, NULL, -1);
            h_out[(size_t)s * nd + i] = val;
            if (status_out) status_out[(size_t)s * nd + i] = st;
        }
    }
    free(An); free(sc); free(bn);
    return rc;
}

/* Row redundancy (the LPs behind MPT3's minHRep, offline_parameters_computation_nxn.m:221-222,399): LP j maximises
 * the unit normal of row j over all the other rows; slack_out[j] = b_j/|a_j| - max (-inf when unbounded). */
int orc_polytope_redundant(const double* A, int m, int n, const double* b, double* slack_out, int* status_out) {
    if (!A || !b || !slack_out || m < 1 || n < 1 || n > LBMPC_MAX_NZ) return LBMPC_ERR_ARG;
    double* An = (double*)malloc(sizeof(double) * (size_t)m * n);
    double* bn = (double*)malloc(sizeof(double) * (size_t)m);
    int rc = 0;
    for (int j = 0; j < m; ++j) {
        double s2 = 0.0;
        for (int i = 0; i < n; ++i) s2 = fma(CM(A, m, j, i), CM(A, m, j, i), s2);
        double nr = sqrt(s2);
        if (!(nr > 0.0) || !(b[j] > 0.0)) { rc = LBMPC_ERR_ARG; break; }
        double sc = 1.0 / nr;
        for (int i = 0; i < n; ++i) An[(size_t)j * n + i] = CM(A, m, j, i) * sc;
        bn[j] = b[j] * sc;
    }
    for (int j = 0; j < m && !rc; ++j) {
        double val;
        int st = lp_support_one(An, bn, m, n, An + (size_t)j * n, &val, NULL, NULL, j);
        slack_out[j] = st == LBMPC_ST_LP_UNBOUNDED ? -INFINITY : bn[j] - val;
        if (status_out) status_out[j] = st;
    }
    free(An); free(bn);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * Scenario MPC (Functions_general/ScenarioMPC.m, Cases_general/Case_6...:96-150), SURVEY.md 8f-3
 * ---------------------------------------------------------------------------------------- */
enum { P_SC_U = 6, P_SC_EV = 7, P_SC_LV = 8 };
#define SC_MAXN LBMPC_MAX_NZ
#define SC_MAXM (LBMPC_MAX_NZ * LBMPC_MAX_NC)

/* Goldfarb-Idnani for min z'Pz s.t. Cm z <= d from z = 0, Ginv = (2P)^-1: the algorithm of qp_solve above on a
 * dense row set (most violated row, ties to the lowest; no pivot candidates). */
static int gi_dense(int n, int m, const double* Cm /* m x n row-major */, const double* d, const double* Ginv,
                    double* z, int* nact_out, int* iters_out) {
    int nq = 0, iters = 0, status = 0, act[SC_MAXN];
    double CA[SC_MAXN][SC_MAXN], Wm[SC_MAXN][SC_MAXN], u[SC_MAXN + 1];
    for (int k = 0; k < n; ++k) z[k] = 0.0;
    for (;;) {
        int p = -1;
        double sp = -QP_TOL_VIOL;
        for (int j = 0; j < m; ++j) {
            int isact = 0;
            for (int a = 0; a < nq; ++a) if (act[a] == j) isact = 1;
            if (isact) continue;
            double s = d[j];
            for (int k = 0; k < n; ++k) s = fma(-Cm[(size_t)j * n + k], z[k], s);
            if (s < sp) { sp = s; p = j; }
        }
        if (p < 0) break;
        double cp[SC_MAXN], g[SC_MAXN];
        for (int k = 0; k < n; ++k) cp[k] = Cm[(size_t)p * n + k];
        for (int i = 0; i < n; ++i) { double acc = 0.0; for (int k = 0; k < n; ++k) acc = fma(Ginv[i * n + k], cp[k], acc); g[i] = acc; }
        double cg = 0.0;
        for (int k = 0; k < n; ++k) cg = fma(cp[k], g[k], cg);
        u[nq] = 0.0;
        for (;;) {
            if (++iters > QP_MAX_ITERS) { status = LBMPC_ST_ITER_CAP; goto done; }
            double r[SC_MAXN], rhs[SC_MAXN], L[SC_MAXN][SC_MAXN], invd[SC_MAXN], delta = cg;
            for (int a = 0; a < nq; ++a) { double acc = 0.0; for (int k = 0; k < n; ++k) acc = fma(CA[a][k], g[k], acc); rhs[a] = acc; }
            for (int a = 0; a < nq; ++a)
                for (int b = 0; b <= a; ++b) {
                    double acc = 0.0;
                    for (int k = 0; k < n; ++k) acc = fma(CA[b][k], Wm[a][k], acc);
                    for (int k = 0; k < b; ++k) acc = fma(-L[a][k], L[b][k], acc);
                    if (a == b) { L[a][a] = sqrt(acc > 0.0 ? acc : 0.0); invd[a] = 1.0 / L[a][a]; }
                    else L[a][b] = acc * invd[b];
                }
            for (int a = 0; a < nq; ++a) { double acc = rhs[a]; for (int k = 0; k < a; ++k) acc = fma(-L[a][k], r[k], acc); r[a] = acc * invd[a]; }
            for (int a = nq - 1; a >= 0; --a) { double acc = r[a]; for (int k = nq - 1; k > a; --k) acc = fma(-L[k][a], r[k], acc); r[a] = acc * invd[a]; }
            for (int a = 0; a < nq; ++a) delta = fma(-rhs[a], r[a], delta);
            double zd[SC_MAXN];
            for (int k = 0; k < n; ++k) { double acc = g[k]; for (int a = 0; a < nq; ++a) acc = fma(-Wm[a][k], r[a], acc); zd[k] = -acc; }
            int dependent = (nq >= n) || !(delta > QP_TOL_DEP * cg);
            double t1 = INFINITY, rmax = 0.0;
            int kdrop = -1;
            for (int a = 0; a < nq; ++a) if (fabs(r[a]) > rmax) rmax = fabs(r[a]);
            for (int a = 0; a < nq; ++a)
                if (r[a] > QP_TOL_R * rmax) { double t = u[a] / r[a]; if (t < t1) { t1 = t; kdrop = a; } }
            double t2 = dependent ? INFINITY : -sp / delta;
            double t = t1 < t2 ? t1 : t2;
            if (t == INFINITY) { status = LBMPC_ST_QP_INFEASIBLE; goto done; }
            for (int a = 0; a < nq; ++a) u[a] = fma(-t, r[a], u[a]);
            u[nq] += t;
            if (!dependent) { for (int k = 0; k < n; ++k) z[k] = fma(t, zd[k], z[k]); sp = fma(t, delta, sp); }
            if (!dependent && t2 <= t1) {
                act[nq] = p;
                memcpy(CA[nq], cp, sizeof(double) * n);
                memcpy(Wm[nq], g, sizeof(double) * n);
                nq++;
                break;
            }
            for (int a = kdrop; a < nq - 1; ++a) { act[a] = act[a + 1]; u[a] = u[a + 1]; memcpy(CA[a], CA[a + 1], sizeof(double) * n); memcpy(Wm[a], Wm[a + 1], sizeof(double) * n); }
            u[nq - 1] = u[nq];
            nq--;
        }
    }
done:
    *nact_out = nq;
    *iters_out = iters;
    return status;
}

typedef struct sc_const {
    int nx, nu, N, nc, n, m, N_SC;
    double A[16], F[LBMPC_MAX_NC * LBMPC_MAX_NX], B[LBMPC_MAX_NX * LBMPC_MAX_NU];  /* row-major */
    double GQ[SC_MAXN][SC_MAXN][LBMPC_MAX_NX]; /* [i][j][a] = (Gamma_i' Q)[j][a] */
    double Cm[SC_MAXM * SC_MAXN];              /* rows (i, r): F_r Gamma_i + G_r e_i' */
    double Ginv[SC_MAXN * SC_MAXN];            /* (2 P)^-1, P = N_SC (sum_i Gamma_i' Q Gamma_i + blkdiag R) */
} sc_const;

static int sc_setup(const lbmpc_scmpc_params* p, sc_const* c) {
    int nx = p->nx, nu = p->nu, N = p->N, nc = p->nc, n = N * nu;
    if (nx < 1 || nx > LBMPC_MAX_NX || nu < 1 || nu > LBMPC_MAX_NU || n > SC_MAXN || nc < 1 || nc > LBMPC_MAX_NC || p->N_SC < 1) return LBMPC_ERR_DIM;
    memset(c, 0, sizeof(*c));
    c->nx = nx; c->nu = nu; c->N = N; c->nc = nc; c->n = n; c->m = N * nc; c->N_SC = p->N_SC;
    for (int a = 0; a < nx; ++a) for (int b = 0; b < nx; ++b) c->A[a * nx + b] = CM(p->A, nx, a, b);
    for (int a = 0; a < nx; ++a) for (int j = 0; j < nu; ++j) c->B[a * nu + j] = CM(p->B, nx, a, j);
    for (int r = 0; r < nc; ++r) for (int b = 0; b < nx; ++b) c->F[r * nx + b] = CM(p->F, nc, r, b);
    /* Gamma_i (nx x n): column block j <= i is A^(i-j) B */
    static double Gam[SC_MAXN][LBMPC_MAX_NX][SC_MAXN];
    double Apow[SC_MAXN][16];
    for (int a = 0; a < nx; ++a) for (int b = 0; b < nx; ++b) Apow[0][a * nx + b] = (a == b);
    for (int t = 1; t < N; ++t)
        for (int a = 0; a < nx; ++a) for (int b = 0; b < nx; ++b) {
            double acc = 0.0;
            for (int q = 0; q < nx; ++q) acc = fma(Apow[t - 1][a * nx + q], c->A[q * nx + b], acc);
            Apow[t][a * nx + b] = acc;
        }
    memset(Gam, 0, sizeof(Gam));
    for (int i = 0; i < N; ++i)
        for (int j = 0; j <= i; ++j)
            for (int a = 0; a < nx; ++a) for (int l = 0; l < nu; ++l) {
                double acc = 0.0;
                for (int q = 0; q < nx; ++q) acc = fma(Apow[i - j][a * nx + q], c->B[q * nu + l], acc);
                Gam[i][a][j * nu + l] = acc;
            }
    double P2[SC_MAXN * SC_MAXN];
    memset(P2, 0, sizeof(P2));
    for (int i = 0; i < N; ++i) {
        for (int j = 0; j < n; ++j) for (int a = 0; a < nx; ++a) {
            double acc = 0.0;
            for (int q = 0; q < nx; ++q) acc = fma(Gam[i][q][j], CM(p->Q, nx, q, a), acc);
            c->GQ[i][j][a] = acc;
        }
        for (int j = 0; j < n; ++j) for (int l = 0; l < n; ++l) {
            double acc = 0.0;
            for (int a = 0; a < nx; ++a) acc = fma(c->GQ[i][j][a], Gam[i][a][l], acc);
            P2[j * n + l] += acc;
        }
        for (int j = 0; j < nu; ++j) for (int l = 0; l < nu; ++l) P2[(i * nu + j) * n + i * nu + l] += CM(p->R, nu, j, l);
        for (int r = 0; r < nc; ++r) for (int j = 0; j < n; ++j) {
            double acc = 0.0;
            for (int a = 0; a < nx; ++a) acc = fma(c->F[r * nx + a], Gam[i][a][j], acc);
            if (j / nu == i) acc += CM(p->G, nc, r, j % nu);
            c->Cm[(size_t)(i * nc + r) * n + j] = acc;
        }
    }
    for (int j = 0; j < n * n; ++j) P2[j] *= 2.0 * (double)p->N_SC;
    if (!invert_spd_like(n, P2, c->Ginv)) return LBMPC_ERR_ARG;
    for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) { double s = 0.5 * (c->Ginv[i * n + j] + c->Ginv[j * n + i]); c->Ginv[i * n + j] = c->Ginv[j * n + i] = s; }
    return 0;
}

int orc_scmpc_mc_run(const lbmpc_scmpc_params* p, const lbmpc_sampler* smp, const lbmpc_mc_cfg* cfg, lbmpc_scmpc_out* out) {
    if (!p || !smp || !cfg || !out || !cfg->x0) return LBMPC_ERR_ARG;
    sc_const* c = (sc_const*)malloc(sizeof(sc_const));
    int rc = sc_setup(p, c);
    if (rc) { free(c); return rc; }
    int nx = c->nx, nu = c->nu, N = c->N, nc = c->nc, n = c->n, m = c->m, K = cfg->steps;
    for (int64_t t = 0; t < cfg->trials; ++t) {
        uint64_t tid = (uint64_t)(cfg->trial0 + t);
        int status = 0, infeasible = 0, iters_total = 0, rec = t < cfg->traj_trials;
        double x[LBMPC_MAX_NX];
        memcpy(x, cfg->x0, sizeof(double) * nx);
        for (int a = 0; a < nx; ++a) if (x[a] > cfg->distance_error || x[a] < -cfg->distance_error) infeasible = 1;
        if (rec && out->traj_x) memcpy(out->traj_x + ((size_t)t * (K + 1)) * nx, x, sizeof(double) * nx);
        for (int k = 0; k < K; ++k) {
            double maxF[SC_MAXM], sum[SC_MAXN][LBMPC_MAX_NX];
            for (int j = 0; j < m; ++j) maxF[j] = -INFINITY;
            memset(sum, 0, sizeof(sum));
            for (int sc = 0; sc < c->N_SC; ++sc) {
                double phi[LBMPC_MAX_NX];
                memcpy(phi, x, sizeof(double) * nx);
                for (int i = 0; i < N; ++i) {
                    uint32_t idx = (uint32_t)((k * c->N_SC + sc) * N + i);
                    double u[LBMPC_MAX_NZ], ev[LBMPC_MAX_NX], lv[LBMPC_MAX_NX], Bu[LBMPC_MAX_NX], pn[LBMPC_MAX_NX];
                    sample_u_lv(smp, cfg->seed, tid, idx, P_SC_U, u);
                    if (sample_polytope(smp->EV_A, smp->EV_b, smp->nEV, smp->EV_lo, smp->EV_hi, smp->EV_U, nx, cfg->seed, tid, idx, P_SC_EV, cfg->sampler_max_attempts, ev)) status = status > LBMPC_ST_SAMPLER_CAP ? status : LBMPC_ST_SAMPLER_CAP;
                    if (sample_polytope(smp->LV_A, smp->LV_b, smp->nLV, smp->LV_lo, smp->LV_hi, smp->LV_U, nx, cfg->seed, tid, idx, P_SC_LV, cfg->sampler_max_attempts, lv)) status = status > LBMPC_ST_SAMPLER_CAP ? status : LBMPC_ST_SAMPLER_CAP;
                    mat_vec(smp->B, nx, nu, u, Bu);
                    for (int a = 0; a < nx; ++a) {
                        double acc = 0.0;
                        for (int b = 0; b < nx; ++b) acc = fma(c->A[a * nx + b], phi[b], acc);
                        pn[a] = acc + ((ev[a] - lv[a]) - Bu[a]);
                    }
                    memcpy(phi, pn, sizeof(double) * nx);
                    for (int r = 0; r < nc; ++r) {
                        double f = 0.0;
                        for (int b = 0; b < nx; ++b) f = fma(c->F[r * nx + b], phi[b], f);
                        if (f > maxF[i * nc + r]) maxF[i * nc + r] = f;
                    }
                    for (int a = 0; a < nx; ++a) sum[i][a] += phi[a];
                }
            }
            double g[SC_MAXN], Uu[SC_MAXN], d[SC_MAXM], z[SC_MAXN];
            for (int j = 0; j < n; ++j) {
                double acc = 0.0;
                for (int i = 0; i < N; ++i) for (int a = 0; a < nx; ++a) acc = fma(c->GQ[i][j][a], sum[i][a], acc);
                g[j] = acc;
            }
            for (int j = 0; j < n; ++j) { double acc = 0.0; for (int l = 0; l < n; ++l) acc = fma(c->Ginv[j * n + l], g[l], acc); Uu[j] = -2.0 * acc; }
            for (int j = 0; j < m; ++j) {
                double acc = 0.0;
                for (int l = 0; l < n; ++l) acc = fma(c->Cm[(size_t)j * n + l], Uu[l], acc);
                d[j] = (1.0 - maxF[j]) - acc;
            }
            int nact = 0, it = 0;
            int st = gi_dense(n, m, c->Cm, d, c->Ginv, z, &nact, &it);
            if (st > status) status = st;
            iters_total += it;
            double u0[LBMPC_MAX_NU];
            for (int j = 0; j < nu; ++j) u0[j] = z[j] + Uu[j];
            /* plant: the step disturbance of lbmpc_mc_run for this (trial, step) */
            double xi_lv[LBMPC_MAX_NX], xi_ev[LBMPC_MAX_NX], u_lv[LBMPC_MAX_NZ], Bul[LBMPC_MAX_NX], Ax[LBMPC_MAX_NX], Bu2[LBMPC_MAX_NX];
            if (sample_polytope(smp->LV_A, smp->LV_b, smp->nLV, smp->LV_lo, smp->LV_hi, smp->LV_U, nx, cfg->seed, tid, (uint32_t)k, P_ST_LV, cfg->sampler_max_attempts, xi_lv)) status = status > LBMPC_ST_SAMPLER_CAP ? status : LBMPC_ST_SAMPLER_CAP;
            sample_u_lv(smp, cfg->seed, tid, (uint32_t)k, P_ST_U, u_lv);
            if (sample_polytope(smp->EV_A, smp->EV_b, smp->nEV, smp->EV_lo, smp->EV_hi, smp->EV_U, nx, cfg->seed, tid, (uint32_t)k, P_ST_EV, cfg->sampler_max_attempts, xi_ev)) status = status > LBMPC_ST_SAMPLER_CAP ? status : LBMPC_ST_SAMPLER_CAP;
            mat_vec(smp->B, nx, nu, u_lv, Bul);
            for (int a = 0; a < nx; ++a) {
                double acc = 0.0, bcc = 0.0;
                for (int b = 0; b < nx; ++b) acc = fma(c->A[a * nx + b], x[b], acc);
                for (int j = 0; j < nu; ++j) bcc = fma(c->B[a * nu + j], u0[j], bcc);
                Ax[a] = acc; Bu2[a] = bcc;
            }
            for (int a = 0; a < nx; ++a) { double wd = (xi_ev[a] - xi_lv[a]) - Bul[a]; x[a] = (Ax[a] + Bu2[a]) + wd; }
            for (int j = 0; j < nu; ++j) if (u0[j] > cfg->acc_bound || u0[j] < -cfg->acc_bound) infeasible = 1;
            for (int a = 0; a < nx; ++a) if (x[a] > cfg->distance_error || x[a] < -cfg->distance_error) infeasible = 1;
            if (rec) {
                if (out->traj_x) memcpy(out->traj_x + ((size_t)t * (K + 1) + k + 1) * nx, x, sizeof(double) * nx);
                if (out->traj_u) memcpy(out->traj_u + ((size_t)t * K + k) * nu, u0, sizeof(double) * nu);
            }
        }
        if (out->infeasible) out->infeasible[t] = infeasible;
        if (out->status) out->status[t] = status;
        if (out->qp_iters) out->qp_iters[t] = iters_total;
        if (out->x_final) memcpy(out->x_final + (size_t)t * nx, x, sizeof(double) * nx);
    }
    free(c);
    return 0;
}
