// polytope.hip -- support function of a polytope in H-rep by an exact LP, batched over directions and right-hand
// sides: h[i, s] = max { C[:,i]' z : A z <= b[:,s] }  (0 strictly feasible: b > 0).
//
// Replaces the LPs Functions_general/ComputeFeasibleRegion.m hands to IPOPT:
//   Com_hs / Com_hs_ini (:447-503)  h_S(F_com) = max F_com e  s.t. S_A e <= S_b
//   Com_z (:505-519)               max M_i z s.t. rows of F_bar Psi^i z <= 1 - hs   (the nu_s search, :66-116)
// and gives the support of the nominal feasible region F_Ns = proj_s {z : F_bar Psi^i z <= 1 - h_k, i = 0..nu}, the set
// the reference samples with 5000 YALMIP projections + convhull (:153-203, :380-435): with it,
// h_{F_N}(d) = h_{F_Ns}(d) + h_{S_hat}(d) is exact and the feasible region is returned as an H-rep template.
//
// Primal active-set method from z = 0 (projection of c onto the null space of the working rows, ratio test with ties
// to the lowest row, Bland's rule on the multipliers), the same procedure and arithmetic as the oracle's
// lp_support_one.  Two mappings of one kernel body:
//   WAVE = false, one LP per lane: the lanes of a wave walk the rows together (every lane reads row j at the same
//     time: one broadcast load per row); pays once there are enough LPs to fill the chip with 64 per wave;
//   WAVE = true, one LP per wave: the ratio test is spread over the lanes (lane l takes rows l, l+64, ...: coalesced
//     row loads, one wave arg-min with ties to the lowest row = the serial scan's choice), the small algebra (z, p,
//     multipliers, the k x k Cholesky factor) is computed redundantly by every lane.  A minHRep of 1 662 rows is
//     1 662 waves instead of 26: the offline scripts' LP batches are thousands, not hundreds of thousands.
// An offline-geometry path, not the closed loop.
#include <cmath>
#include <cstdlib>
#include <vector>

#include "common.hpp"

namespace lbmpc {

#define LP_MAX_ITERS 400

template <bool WAVE>
__global__ __launch_bounds__(256) void polytope_support_kernel(const double* __restrict__ A /* m x n row-major, unit rows */,
                                                              const double* __restrict__ bn /* m x nb, scaled */, int m, int n,
                                                              const double* __restrict__ Cd /* n x nd */, int64_t nd, int64_t nb,
                                                              double* __restrict__ h_out, double* __restrict__ z_out,
                                                              int* __restrict__ status_out,
                                                              const int* __restrict__ skip /* nullable: row LP t ignores */) {
    const int64_t thread = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t t = WAVE ? thread / LB_WAVE : thread;   // LP of this lane
    const int lane = lane_id();
    const bool valid = t < nd * nb;
    const int64_t tt = valid ? t : nd * nb - 1;
    const int64_t s = tt / nd, i0 = tt % nd;
    const double* b = bn + (size_t)s * m;
    const int skipj = skip ? skip[tt] : -1;
    double z[LBMPC_MAX_NZ], p[LBMPC_MAX_NZ], lam[LBMPC_MAX_NZ], c[LBMPC_MAX_NZ], L[LBMPC_MAX_NZ][LBMPC_MAX_NZ];
    int W[LBMPC_MAX_NZ], k = 0, iters = 0, status = 0;
    double cn = 0.0;
    for (int i = 0; i < n; ++i) { z[i] = 0.0; c[i] = Cd[(size_t)i0 * n + i]; if (fabs(c[i]) > cn) cn = fabs(c[i]); }
    bool done = cn == 0.0;
    // lanes stay together through the row scans: the loop runs until every lane of the wave is done
    while (__builtin_amdgcn_ballot_w64(!done) != 0) {
        if (!done && ++iters > LP_MAX_ITERS) { status = LBMPC_ST_ITER_CAP; done = true; }
        bool scan = false;
        double pn = 0.0;
        if (!done) {
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
            for (int i = 0; i < n; ++i) {
                double acc = c[i];
                for (int a = 0; a < k; ++a) acc = fma(-A[(size_t)W[a] * n + i], lam[a], acc);
                p[i] = acc;
                if (fabs(acc) > pn) pn = fabs(acc);
            }
            if (k == n || pn <= 1e-11 * cn) {
                int drop = -1;
                for (int a = 0; a < k; ++a)
                    if (lam[a] < -1e-10 * cn && (drop < 0 || W[a] < W[drop])) drop = a;
                if (drop < 0) done = true;   // optimal
                else { for (int a = drop; a < k - 1; ++a) W[a] = W[a + 1]; --k; }
            } else {
                scan = true;
            }
        }
        if (__builtin_amdgcn_ballot_w64(scan) != 0) {
            double best = LB_INF;
            int jb = -1;
            for (int j = WAVE ? lane : 0; j < m; j += WAVE ? LB_WAVE : 1) {
                const double* row = A + (size_t)j * n;   // WAVE: one row per lane; else a wave-uniform address
                double ap = 0.0, az = 0.0;
                for (int i = 0; i < n; ++i) { const double aji = row[i]; ap = fma(aji, p[i], ap); az = fma(aji, z[i], az); }
                bool inW = false;
                for (int a = 0; a < k; ++a) inW = inW || W[a] == j;
                if (scan && !inW && j != skipj && ap > 1e-12 * pn) {
                    double sl = b[j] - az;
                    if (sl < 0.0) sl = 0.0;
                    const double r = sl / ap;
                    if (r < best) { best = r; jb = j; }
                }
            }
            if (WAVE) {   // smallest ratio over the lanes, ties to the lowest row: what the serial scan keeps
                int jj = jb < 0 ? 0x7fffffff : jb;
                wave_argmin(best, jj);
                jb = jj == 0x7fffffff ? -1 : jj;
            }
            if (scan) {
                if (jb < 0) { status = LBMPC_ST_LP_UNBOUNDED; done = true; }
                else {
                    for (int i = 0; i < n; ++i) z[i] = fma(best, p[i], z[i]);
                    W[k++] = jb;
                }
            }
        }
    }
    if (!valid || (WAVE && lane != 0)) return;
    double val = 0.0;
    for (int i = 0; i < n; ++i) val = fma(c[i], z[i], val);
    h_out[t] = val;
    if (z_out) for (int i = 0; i < n; ++i) z_out[(size_t)t * n + i] = z[i];
    if (status_out) status_out[t] = status;
}

// one LP per wave until one LP per lane would still put two waves on every SIMD (LBMPC_LP_MAP=lane|wave forces one: testing)
static void launch_support(int64_t total, const double* A, const double* bn, int m, int n, const double* Cd, int64_t nd, int64_t nb,
                           double* h, double* z, int* st, const int* skip) {
    bool wave = total < (int64_t)LB_WAVE * 2048;
    if (const char* e = getenv("LBMPC_LP_MAP")) wave = e[0] == 'w' ? true : (e[0] == 'l' ? false : wave);
    if (wave)
        hipLaunchKernelGGL(polytope_support_kernel<true>, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, nullptr, A, bn, m, n, Cd, nd, nb, h, z, st, skip);
    else
        hipLaunchKernelGGL(polytope_support_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, nullptr, A, bn, m, n, Cd, nd, nb, h, z, st, skip);
}

}  // namespace lbmpc

using namespace lbmpc;

extern "C" int lbmpc_polytope_support(const double* A, int m, int n, const double* b, int64_t nb, const double* C,
                                      int64_t nd, double* h_out, double* z_out, int* status_out) {
    if (int rc = check_device()) return rc;
    if (!A || !b || !C || !h_out || m < 1 || n < 1 || nb < 0 || nd < 0) { set_error("lbmpc_polytope_support: bad argument"); return LBMPC_ERR_ARG; }
    if (n > LBMPC_MAX_NZ || m > 262144) { set_error("lbmpc_polytope_support: n <= %d, m <= 262144", LBMPC_MAX_NZ); return LBMPC_ERR_DIM; }
    if (nb == 0 || nd == 0) return 0;
    // unit rows (same operations as the oracle); a zero row never blocks
    std::vector<double> An((size_t)m * n), sc(m), bn((size_t)m * nb);
    for (int j = 0; j < m; ++j) {
        double s2 = 0.0;
        for (int i = 0; i < n; ++i) s2 = std::fma(A[j + (size_t)m * i], A[j + (size_t)m * i], s2);
        const double nr = std::sqrt(s2);
        sc[j] = nr > 0.0 ? 1.0 / nr : 0.0;
        for (int i = 0; i < n; ++i) An[(size_t)j * n + i] = A[j + (size_t)m * i] * sc[j];
    }
    for (int64_t s = 0; s < nb; ++s)
        for (int j = 0; j < m; ++j) {
            const double bj = b[(size_t)s * m + j];
            if (!(bj > 0.0)) { set_error("lbmpc_polytope_support: b[%d, %lld] = %g: z = 0 must be strictly feasible", j, (long long)s, bj); return LBMPC_ERR_ARG; }
            bn[(size_t)s * m + j] = sc[j] > 0.0 ? bj * sc[j] : INFINITY;
        }
    const int64_t total = nd * nb;
    DevBuf dA, db, dC, dh, dz, dst;
    LB_HIP(dA.alloc(An.size() * 8)); LB_HIP(db.alloc(bn.size() * 8)); LB_HIP(dC.alloc((size_t)nd * n * 8));
    LB_HIP(dh.alloc((size_t)total * 8));
    if (z_out) LB_HIP(dz.alloc((size_t)total * n * 8));
    if (status_out) LB_HIP(dst.alloc((size_t)total * 4));
    LB_HIP(hipMemcpy(dA.p, An.data(), An.size() * 8, hipMemcpyHostToDevice));
    LB_HIP(hipMemcpy(db.p, bn.data(), bn.size() * 8, hipMemcpyHostToDevice));
    LB_HIP(hipMemcpy(dC.p, C, (size_t)nd * n * 8, hipMemcpyHostToDevice));
    launch_support(total, dA.as<double>(), db.as<double>(), m, n, dC.as<double>(), nd, nb, dh.as<double>(), z_out ? dz.as<double>() : nullptr,
                   status_out ? dst.as<int>() : nullptr, nullptr);
    LB_HIP(hipGetLastError());
    LB_HIP(hipMemcpy(h_out, dh.p, (size_t)total * 8, hipMemcpyDeviceToHost));
    if (z_out) LB_HIP(hipMemcpy(z_out, dz.p, (size_t)total * n * 8, hipMemcpyDeviceToHost));
    if (status_out) LB_HIP(hipMemcpy(status_out, dst.p, (size_t)total * 4, hipMemcpyDeviceToHost));
    return 0;
}

// Row redundancy of {z : A z <= b} (the LPs behind MPT3's minHRep, offline_parameters_computation_nxn.m:221-222,399,
// ComputeFeasibleRegion.m:112): LP j maximises the unit normal of row j over all the OTHER rows, one lane per row.
// slack_out[j] = b_j/|a_j| - max: >= 0 means the other rows already imply row j; -inf (status LP_UNBOUNDED) means
// row j is what bounds that direction.
extern "C" int lbmpc_polytope_redundant(const double* A, int m, int n, const double* b, double* slack_out, int* status_out) {
    if (int rc = check_device()) return rc;
    if (!A || !b || !slack_out || m < 1 || n < 1) { set_error("lbmpc_polytope_redundant: bad argument"); return LBMPC_ERR_ARG; }
    if (n > LBMPC_MAX_NZ || m > 262144) { set_error("lbmpc_polytope_redundant: n <= %d, m <= 262144", LBMPC_MAX_NZ); return LBMPC_ERR_DIM; }
    std::vector<double> An((size_t)m * n), bn(m), h(m);
    std::vector<int> skip(m), st(m);
    for (int j = 0; j < m; ++j) {
        double s2 = 0.0;
        for (int i = 0; i < n; ++i) s2 = std::fma(A[j + (size_t)m * i], A[j + (size_t)m * i], s2);
        const double nr = std::sqrt(s2);
        if (!(nr > 0.0)) { set_error("lbmpc_polytope_redundant: row %d is zero", j); return LBMPC_ERR_ARG; }
        if (!(b[j] > 0.0)) { set_error("lbmpc_polytope_redundant: b[%d] = %g: z = 0 must be strictly feasible", j, b[j]); return LBMPC_ERR_ARG; }
        const double sc = 1.0 / nr;
        for (int i = 0; i < n; ++i) An[(size_t)j * n + i] = A[j + (size_t)m * i] * sc;
        bn[j] = b[j] * sc;
        skip[j] = j;
    }
    DevBuf dA, db, dh, dst, dsk;
    LB_HIP(dA.alloc(An.size() * 8)); LB_HIP(db.alloc((size_t)m * 8)); LB_HIP(dh.alloc((size_t)m * 8));
    LB_HIP(dst.alloc((size_t)m * 4)); LB_HIP(dsk.alloc((size_t)m * 4));
    LB_HIP(hipMemcpy(dA.p, An.data(), An.size() * 8, hipMemcpyHostToDevice));
    LB_HIP(hipMemcpy(db.p, bn.data(), (size_t)m * 8, hipMemcpyHostToDevice));
    LB_HIP(hipMemcpy(dsk.p, skip.data(), (size_t)m * 4, hipMemcpyHostToDevice));
    // direction j = unit row j: the normalised row-major A doubles as the n x nd direction array
    launch_support(m, dA.as<double>(), db.as<double>(), m, n, dA.as<double>(), (int64_t)m, (int64_t)1, dh.as<double>(), nullptr, dst.as<int>(),
                   dsk.as<int>());
    LB_HIP(hipGetLastError());
    LB_HIP(hipMemcpy(h.data(), dh.p, (size_t)m * 8, hipMemcpyDeviceToHost));
    LB_HIP(hipMemcpy(st.data(), dst.p, (size_t)m * 4, hipMemcpyDeviceToHost));
    for (int j = 0; j < m; ++j) {
        slack_out[j] = st[j] == LBMPC_ST_LP_UNBOUNDED ? -INFINITY : bn[j] - h[j];
        if (status_out) status_out[j] = st[j];
    }
    return 0;
}
