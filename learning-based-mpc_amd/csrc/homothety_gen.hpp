// homothety_gen.hpp -- the homothety LP for a general-polytope prior W by vertex enumeration, one LP per lane.
// Used by the set-learning entry point (set_learning.hip: homothety_lp_general_kernel) and by the closed-loop set update
// of contexts whose W is not an axis-aligned box (uqmpc.hip: LPforUQ, UQRobustMPC.m:124-139).
#pragma once

namespace lbmpc {

// Same procedure and arithmetic as the oracle's orc_homothety_lp_enum + orc_homothety_general (plain mul / add,
// partial pivoting, subsets in lexicographic order, first best vertex wins): beta* = max over the feasible basic
// solutions of the (2 nW + 2)-row LP in (beta, y); y = mean of the feasible basic solutions of the optimal face.
// C(2 nW + 2, nx + 1) small solves per LP: a handful of LPs per experiment in set learning; in the closed loop one
// per step in which the new disturbance falls outside the current set (under 1 % of the lane-steps of the reference's
// workloads).
struct GenW {
    int nW, nx;
    double WA[LBMPC_MAX_NW * LBMPC_MAX_NX];   // caller row order, row-major
    double Wb[LBMPC_MAX_NW];
};

__device__ inline bool solve_dense_dev(int n, double* M, double* rhs) {
    for (int c = 0; c < n; ++c) {
        int piv = c;
        for (int r = c + 1; r < n; ++r) if (fabs(M[r * n + c]) > fabs(M[piv * n + c])) piv = r;
        if (fabs(M[piv * n + c]) < 1e-13) return false;
        if (piv != c) {
            for (int k = 0; k < n; ++k) { const double t = M[c * n + k]; M[c * n + k] = M[piv * n + k]; M[piv * n + k] = t; }
            const double t = rhs[c]; rhs[c] = rhs[piv]; rhs[piv] = t;
        }
        for (int r = c + 1; r < n; ++r) {
            const double f = M[r * n + c] / M[c * n + c];
            for (int k = c; k < n; ++k) M[r * n + k] -= f * M[c * n + k];
            rhs[r] -= f * rhs[c];
        }
    }
    for (int c = n - 1; c >= 0; --c) {
        double acc = rhs[c];
        for (int k = c + 1; k < n; ++k) acc -= M[c * n + k] * rhs[k];
        rhs[c] = acc / M[c * n + c];
    }
    return true;
}

// m: nW row maxima (caller row order).  Returns the status; alpha, v[0..nx), mask as lbmpc_homothety_lp documents.
__device__ inline int homothety_general_dev(const GenW& W, const double* m, double& alpha_out, double* v_out, uint32_t& mask_out) {
    const int nW = W.nW, nx = W.nx, nv = nx + 1, nr = 2 * nW + 2;
    double R[(2 * LBMPC_MAX_NW + 2) * (LBMPC_MAX_NX + 1)], rhs[2 * LBMPC_MAX_NW + 2];
    for (int j = 0; j < nW; ++j) {
        R[j * nv] = -W.Wb[j];
        for (int k = 0; k < nx; ++k) R[j * nv + 1 + k] = W.WA[j * nx + k];
        rhs[j] = 0.0;
        R[(nW + j) * nv] = W.Wb[j];
        for (int k = 0; k < nx; ++k) R[(nW + j) * nv + 1 + k] = -W.WA[j * nx + k];
        rhs[nW + j] = W.Wb[j] - m[j];
    }
    for (int k = 0; k < nv; ++k) { R[(2 * nW) * nv + k] = 0.0; R[(2 * nW + 1) * nv + k] = 0.0; }
    R[(2 * nW) * nv] = -1.0; rhs[2 * nW] = 0.0;
    R[(2 * nW + 1) * nv] = 1.0; rhs[2 * nW + 1] = 1.0;
    int idx[LBMPC_MAX_NX + 1];
    for (int k = 0; k < nv; ++k) idx[k] = k;
    double best = -LB_INF, bx[LBMPC_MAX_NX + 1];
    bool found = false;
    for (;;) {
        double M[(LBMPC_MAX_NX + 1) * (LBMPC_MAX_NX + 1)], x[LBMPC_MAX_NX + 1];
        for (int a = 0; a < nv; ++a) {
            for (int k = 0; k < nv; ++k) M[a * nv + k] = R[idx[a] * nv + k];
            x[a] = rhs[idx[a]];
        }
        if (solve_dense_dev(nv, M, x)) {
            bool ok = true;
            for (int r = 0; r < nr && ok; ++r) {
                double acc = 0.0;
                for (int k = 0; k < nv; ++k) acc += R[r * nv + k] * x[k];
                if (acc > rhs[r] + 1e-11) ok = false;
            }
            if (ok && x[0] > best) { best = x[0]; for (int k = 0; k < nv; ++k) bx[k] = x[k]; found = true; }
        }
        int p = nv - 1;
        while (p >= 0 && idx[p] == nr - nv + p) --p;
        if (p < 0) break;
        ++idx[p];
        for (int k = p + 1; k < nv; ++k) idx[k] = idx[k - 1] + 1;
    }
    if (!found) {
        alpha_out = 1.0;
        for (int k = 0; k < nx; ++k) v_out[k] = 0.0;
        mask_out = 0;
        return LBMPC_ST_LP_INFEASIBLE;
    }
    const double a1 = 1.0 - bx[0];
    const double beta = 1.0 - a1;
    // stage 2: mean of the feasible basic solutions of the optimal face (rows reuse R / rhs, nx columns)
    const int nr2 = 2 * nW;
    for (int j = 0; j < nW; ++j) {
        for (int k = 0; k < nx; ++k) { R[j * nx + k] = W.WA[j * nx + k]; R[(nW + j) * nx + k] = -W.WA[j * nx + k]; }
        rhs[j] = beta * W.Wb[j];
        rhs[nW + j] = (W.Wb[j] - m[j]) - beta * W.Wb[j];
    }
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
        if (solve_dense_dev(nx, M, x)) {
            bool ok = true;
            for (int r = 0; r < nr2 && ok; ++r) {
                double acc = 0.0;
                for (int k = 0; k < nx; ++k) acc += R[r * nx + k] * x[k];
                if (acc > rhs[r] + 1e-11) ok = false;
            }
            if (ok) { for (int k = 0; k < nx; ++k) sum[k] += x[k]; ++count; }
        }
        int p = nx - 1;
        while (p >= 0 && idx[p] == nr2 - nx + p) --p;
        if (p < 0) break;
        ++idx[p];
        for (int k = p + 1; k < nx; ++k) idx[k] = idx[k - 1] + 1;
    }
    double y[LBMPC_MAX_NX];
    for (int k = 0; k < nx; ++k) y[k] = count > 0 ? sum[k] / (double)count : bx[1 + k];
    alpha_out = a1;
    for (int k = 0; k < nx; ++k) v_out[k] = beta > 0.0 ? y[k] / beta : 0.0;
    uint32_t mask = 0;
    for (int j = 0; j < nW; ++j) {
        double acc = 0.0;
        for (int k = 0; k < nx; ++k) acc += R[j * nx + k] * y[k];
        if (fabs(rhs[j] - acc) <= 1e-9 || fabs(rhs[nW + j] + acc) <= 1e-9) mask |= 1u << j;
    }
    mask_out = mask;
    return 0;
}

}  // namespace lbmpc
