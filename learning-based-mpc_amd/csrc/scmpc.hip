// scmpc.hip -- Scenario MPC, the comparison baseline of Case_6.
//
// Replaces Functions_general/ScenarioMPC.m:52-101 (sample_scenarios + a CasADi/IPOPT QP with N_SC*nx*(N+1) state
// variables and N_SC*N*nc rows per step) and its loop in Cases_general/Case_6...:96-150.  With the states eliminated
// the QP has N*nu variables; its N_SC*N*nc rows share nc*N normals (F Gamma_i + G e_i'), so only the tightest scenario
// per normal can bind: the scenario loop is a running max (rows) and a running sum (linear cost term) over the
// N_SC*N sampled disturbances, then one small dense QP.  One trial per wave: the scenarios of a step are spread over
// the lanes (3 Philox blocks per disturbance, N_SC*N = 400 per step: this is what a step costs), the QP
// (Goldfarb-Idnani on nc*N rows, the oracle's gi_dense) runs in lane 0's private memory.  Same arithmetic as
// oracle/lbmpc_oracle.c (orc_scmpc_mc_run): flags, statuses and iteration counts are compared with ==.
#include <cmath>
#include <cstring>
#include <vector>

#include "common.hpp"

namespace lbmpc {

#define SC_MAXN LBMPC_MAX_NZ
#define SC_MAXM (LBMPC_MAX_NZ * LBMPC_MAX_NC)
#define SC_TOL_VIOL 1e-10
#define SC_TOL_DEP 1e-13
#define SC_TOL_R 1e-12
#define SC_MAX_ITERS 200
enum { P_SC_U = 6, P_SC_EV = 7, P_SC_LV = 8 };

struct ScConst {
    int nx, nu, N, nc, n, m, N_SC;
    double A[16], F[LBMPC_MAX_NC * LBMPC_MAX_NX], B[LBMPC_MAX_NX * LBMPC_MAX_NU];
    double GQ[SC_MAXN][SC_MAXN][LBMPC_MAX_NX];
    double Cm[SC_MAXM * SC_MAXN];
    double Ginv[SC_MAXN * SC_MAXN];
};

__device__ inline int gi_dense_dev(int n, int m, const double* __restrict__ Cm, const double* d, const double* __restrict__ Ginv,
                                   double* z, int* iters_out) {
    int nq = 0, iters = 0, status = 0, act[SC_MAXN];
    double CA[SC_MAXN][SC_MAXN], Wm[SC_MAXN][SC_MAXN], u[SC_MAXN + 1];
    for (int k = 0; k < n; ++k) z[k] = 0.0;
    for (;;) {
        int p = -1;
        double sp = -SC_TOL_VIOL;
        for (int j = 0; j < m; ++j) {
            bool isact = false;
            for (int a = 0; a < nq; ++a) isact = isact || act[a] == j;
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
        bool added = false;
        while (!added) {
            if (++iters > SC_MAX_ITERS) { *iters_out = iters; return LBMPC_ST_ITER_CAP; }
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
            const bool dependent = (nq >= n) || !(delta > SC_TOL_DEP * cg);
            double t1 = LB_INF, rmax = 0.0;
            int kdrop = -1;
            for (int a = 0; a < nq; ++a) if (fabs(r[a]) > rmax) rmax = fabs(r[a]);
            for (int a = 0; a < nq; ++a)
                if (r[a] > SC_TOL_R * rmax) { const double t = u[a] / r[a]; if (t < t1) { t1 = t; kdrop = a; } }
            const double t2 = dependent ? LB_INF : -sp / delta;
            const double t = t1 < t2 ? t1 : t2;
            if (t == LB_INF) { *iters_out = iters; return LBMPC_ST_QP_INFEASIBLE; }
            for (int a = 0; a < nq; ++a) u[a] = fma(-t, r[a], u[a]);
            u[nq] += t;
            if (!dependent) { for (int k = 0; k < n; ++k) z[k] = fma(t, zd[k], z[k]); sp = fma(t, delta, sp); }
            if (!dependent && t2 <= t1) {
                act[nq] = p;
                for (int k = 0; k < n; ++k) { CA[nq][k] = cp[k]; Wm[nq][k] = g[k]; }
                nq++;
                added = true;
            } else {
                for (int a = kdrop; a < nq - 1; ++a) {
                    act[a] = act[a + 1]; u[a] = u[a + 1];
                    for (int k = 0; k < n; ++k) { CA[a][k] = CA[a + 1][k]; Wm[a][k] = Wm[a + 1][k]; }
                }
                u[nq - 1] = u[nq];
                nq--;
            }
        }
    }
    *iters_out = iters;
    return status;
}

struct ScArgs {
    int64_t trials, trial0;
    int steps, traj_trials, max_attempts;
    uint64_t seed;
    double x0[LBMPC_MAX_NX], acc_bound, distance_error;
    lbmpc_scmpc_out out;   // device pointers
};

// sample_polytope() (common.hpp) for a wave in which only some lanes draw: the others count as done from the start
template <int NX>
__device__ __forceinline__ bool sample_polytope_if(bool act, const double* __restrict__ XA, const double* __restrict__ Xb, int nr,
                                                   const double* __restrict__ lo, const double* __restrict__ hi,
                                                   const double* __restrict__ U, uint64_t seed, uint64_t trial, uint32_t index,
                                                   uint32_t purpose, int max_attempts, double* x) {
    bool done = !act;
    for (int a = 0; a < max_attempts; ++a) {
        if (!done) {
            double u[NX], xt[NX];
            draw_uniforms<NX>(seed, trial, index, purpose, (uint32_t)a, u);
            proposal_point<NX>(lo, hi, U, u, xt);
            bool ok = true;
            for (int r = 0; r < nr; ++r) {
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < NX; ++k) acc = fma(XA[r * NX + k], xt[k], acc);
                ok = ok && (acc <= Xb[r]);
            }
#pragma unroll
            for (int k = 0; k < NX; ++k) x[k] = xt[k];
            done = ok;
        }
        if (__builtin_amdgcn_ballot_w64(!done) == 0) break;
    }
    return act && !done;
}

// One trial per WAVE: the N_SC scenarios of a step are spread over the lanes (lane l propagates scenarios l, l+64, ...
// through the N stages, 3 Philox blocks per stage), the per-normal maxima are wave reductions, the per-stage sums are
// accumulated in scenario order (the oracle's order: lane values are fetched one by one), and lane 0 solves the QP.
// With one trial per LANE a step was N_SC*N serial draws per lane and a 4096-trial batch filled 64 of 1024 SIMDs.
template <int NX>
__global__ __launch_bounds__(64) void scmpc_kernel(const ScConst* __restrict__ C, SamplerDev smp, ScArgs a) {
    __shared__ double maxF[SC_MAXM], sum[SC_MAXN][NX], zsh[SC_MAXN];
    __shared__ int qp_st[2];
    const int lane = threadIdx.x;
    const int64_t t = blockIdx.x;
    const uint64_t tid = (uint64_t)(a.trial0 + t);
    const int nu = C->nu, N = C->N, nc = C->nc, n = C->n, m = C->m, K = a.steps, NSC = C->N_SC;
    int status = 0, infeasible = 0, iters_total = 0;
    const bool rec = t < a.traj_trials;
    double x[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) { x[i] = a.x0[i]; if (x[i] > a.distance_error || x[i] < -a.distance_error) infeasible = 1; }
    if (rec && lane == 0 && a.out.traj_x) for (int i = 0; i < NX; ++i) a.out.traj_x[(t * (K + 1)) * NX + i] = x[i];
    const double* EVU = smp.EV_hasU ? smp.EV_U : nullptr;
    const double* LVU = smp.LV_hasU ? smp.LV_U : nullptr;
    for (int k = 0; k < K; ++k) {
        for (int j = lane; j < m; j += 64) maxF[j] = -LB_INF;
        for (int j = lane; j < N * NX; j += 64) sum[j / NX][j % NX] = 0.0;
        __syncthreads();
        for (int sc0 = 0; sc0 < NSC; sc0 += 64) {
            const int sc = sc0 + lane;
            const bool act = sc < NSC;
            const int nact = NSC - sc0 < 64 ? NSC - sc0 : 64;
            double phi[NX];
#pragma unroll
            for (int q = 0; q < NX; ++q) phi[q] = x[q];
            for (int i = 0; i < N; ++i) {
                const uint32_t idx = (uint32_t)((k * NSC + (act ? sc : sc0)) * N + i);
                double ev[NX], lv[NX], Bu[NX], pn[NX];
                sample_Bu_lv<NX>(&smp, a.seed, tid, idx, P_SC_U, Bu);
                bool cap = sample_polytope_if<NX>(act, smp.EV_A, smp.EV_b, smp.nEV, smp.EV_lo, smp.EV_hi, EVU, a.seed, tid, idx, P_SC_EV, a.max_attempts, ev);
                cap |= sample_polytope_if<NX>(act, smp.LV_A, smp.LV_b, smp.nLV, smp.LV_lo, smp.LV_hi, LVU, a.seed, tid, idx, P_SC_LV, a.max_attempts, lv);
                if (__builtin_amdgcn_ballot_w64(cap) != 0) status = status > LBMPC_ST_SAMPLER_CAP ? status : LBMPC_ST_SAMPLER_CAP;
#pragma unroll
                for (int q = 0; q < NX; ++q) {
                    double acc = 0.0;
#pragma unroll
                    for (int b = 0; b < NX; ++b) acc = fma(C->A[q * NX + b], phi[b], acc);
                    pn[q] = act ? acc + ((ev[q] - lv[q]) - Bu[q]) : 0.0;
                }
#pragma unroll
                for (int q = 0; q < NX; ++q) phi[q] = pn[q];
                for (int r = 0; r < nc; ++r) {
                    double f = 0.0;
#pragma unroll
                    for (int b = 0; b < NX; ++b) f = fma(C->F[r * NX + b], phi[b], f);
                    const double fm = wave_max(act ? f : -LB_INF);
                    if (lane == 0 && fm > maxF[i * nc + r]) maxF[i * nc + r] = fm;
                }
                // sum over the scenarios in scenario order (bit-identical to the serial accumulation)
#pragma unroll
                for (int q = 0; q < NX; ++q) {
                    double acc = sum[i][q];
                    for (int l2 = 0; l2 < nact; ++l2) acc += readlane_f64(phi[q], l2);
                    if (lane == 0) sum[i][q] = acc;
                }
            }
        }
        __syncthreads();
        double Uu[SC_MAXN];
        {
            double g[SC_MAXN];
            for (int j = 0; j < n; ++j) {
                double acc = 0.0;
                for (int i = 0; i < N; ++i)
#pragma unroll
                    for (int q = 0; q < NX; ++q) acc = fma(C->GQ[i][j][q], sum[i][q], acc);
                g[j] = acc;
            }
            for (int j = 0; j < n; ++j) { double acc = 0.0; for (int q = 0; q < n; ++q) acc = fma(C->Ginv[j * n + q], g[q], acc); Uu[j] = -2.0 * acc; }
        }
        if (lane == 0) {
            double d[SC_MAXM], z[SC_MAXN];
            for (int j = 0; j < m; ++j) {
                double acc = 0.0;
                for (int q = 0; q < n; ++q) acc = fma(C->Cm[(size_t)j * n + q], Uu[q], acc);
                d[j] = (1.0 - maxF[j]) - acc;
            }
            int it = 0;
            qp_st[0] = gi_dense_dev(n, m, C->Cm, d, C->Ginv, z, &it);
            qp_st[1] = it;
            for (int j = 0; j < n; ++j) zsh[j] = z[j];
        }
        __syncthreads();
        status = qp_st[0] > status ? qp_st[0] : status;
        iters_total += qp_st[1];
        double u0[LBMPC_MAX_NU];
        for (int j = 0; j < nu; ++j) u0[j] = zsh[j] + Uu[j];
        // plant: the step disturbance of lbmpc_mc_run for this (trial, step); every lane computes the same values
        double xl[NX], xe[NX], Bul[NX], Ax[NX], Bu2[NX];
        if (sample_polytope<NX>(smp.LV_A, smp.LV_b, smp.nLV, smp.LV_lo, smp.LV_hi, LVU, a.seed, tid, (uint32_t)k, P_ST_LV, a.max_attempts, xl)) status = status > LBMPC_ST_SAMPLER_CAP ? status : LBMPC_ST_SAMPLER_CAP;
        sample_Bu_lv<NX>(&smp, a.seed, tid, (uint32_t)k, P_ST_U, Bul);
        if (sample_polytope<NX>(smp.EV_A, smp.EV_b, smp.nEV, smp.EV_lo, smp.EV_hi, EVU, a.seed, tid, (uint32_t)k, P_ST_EV, a.max_attempts, xe)) status = status > LBMPC_ST_SAMPLER_CAP ? status : LBMPC_ST_SAMPLER_CAP;
#pragma unroll
        for (int q = 0; q < NX; ++q) {
            double acc = 0.0, bcc = 0.0;
#pragma unroll
            for (int b = 0; b < NX; ++b) acc = fma(C->A[q * NX + b], x[b], acc);
            for (int j = 0; j < nu; ++j) bcc = fma(C->B[q * nu + j], u0[j], bcc);
            Ax[q] = acc; Bu2[q] = bcc;
        }
#pragma unroll
        for (int q = 0; q < NX; ++q) { const double wd = (xe[q] - xl[q]) - Bul[q]; x[q] = (Ax[q] + Bu2[q]) + wd; }
        for (int j = 0; j < nu; ++j) if (u0[j] > a.acc_bound || u0[j] < -a.acc_bound) infeasible = 1;
#pragma unroll
        for (int q = 0; q < NX; ++q) if (x[q] > a.distance_error || x[q] < -a.distance_error) infeasible = 1;
        if (rec && lane == 0) {
            if (a.out.traj_x) for (int q = 0; q < NX; ++q) a.out.traj_x[(t * (K + 1) + k + 1) * NX + q] = x[q];
            if (a.out.traj_u) for (int j = 0; j < nu; ++j) a.out.traj_u[(t * K + k) * nu + j] = u0[j];
        }
        __syncthreads();   // the next step re-initialises maxF / sum
    }
    if (lane != 0) return;
    if (a.out.infeasible) a.out.infeasible[t] = infeasible;
    if (a.out.status) a.out.status[t] = status;
    if (a.out.qp_iters) a.out.qp_iters[t] = iters_total;
    if (a.out.x_final) for (int q = 0; q < NX; ++q) a.out.x_final[t * NX + q] = x[q];
}

static bool invert_gj_sc(int n, const double* M, double* Minv) {
    std::vector<double> a((size_t)n * 2 * n);
    const int w = 2 * n;
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { a[i * w + j] = M[i * n + j]; a[i * w + n + j] = (i == j) ? 1.0 : 0.0; }
    for (int c = 0; c < n; ++c) {
        int piv = c;
        for (int r = c + 1; r < n; ++r) if (std::fabs(a[r * w + c]) > std::fabs(a[piv * w + c])) piv = r;
        if (std::fabs(a[piv * w + c]) < 1e-300) return false;
        if (piv != c) for (int k = 0; k < w; ++k) std::swap(a[c * w + k], a[piv * w + k]);
        const double d = a[c * w + c];
        for (int k = 0; k < w; ++k) a[c * w + k] /= d;
        for (int r = 0; r < n; ++r) {
            if (r == c) continue;
            const double f = a[r * w + c];
            if (f != 0.0) for (int k = 0; k < w; ++k) a[r * w + k] -= f * a[c * w + k];
        }
    }
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) Minv[i * n + j] = a[i * w + n + j];
    return true;
}

#define CMS(M, rows, i, j) ((M)[(size_t)(i) + (size_t)(rows) * (size_t)(j)])

// host: the constant blocks of the condensed scenario QP (same operations as the oracle's sc_setup)
static int sc_setup(const lbmpc_scmpc_params* p, ScConst* c) {
    const int nx = p->nx, nu = p->nu, N = p->N, nc = p->nc, n = N * nu;
    if (nx < 1 || nx > LBMPC_MAX_NX || nu < 1 || nu > LBMPC_MAX_NU || N < 1 || n > SC_MAXN || nc < 1 || nc > LBMPC_MAX_NC || p->N_SC < 1) {
        set_error("lbmpc_scmpc_mc_run: limits nx<=%d nu<=%d N*nu<=%d nc<=%d N_SC>=1", LBMPC_MAX_NX, LBMPC_MAX_NU, SC_MAXN, LBMPC_MAX_NC);
        return LBMPC_ERR_DIM;
    }
    std::memset(c, 0, sizeof(*c));
    c->nx = nx; c->nu = nu; c->N = N; c->nc = nc; c->n = n; c->m = N * nc; c->N_SC = p->N_SC;
    for (int a = 0; a < nx; ++a) for (int b = 0; b < nx; ++b) c->A[a * nx + b] = CMS(p->A, nx, a, b);
    for (int a = 0; a < nx; ++a) for (int j = 0; j < nu; ++j) c->B[a * nu + j] = CMS(p->B, nx, a, j);
    for (int r = 0; r < nc; ++r) for (int b = 0; b < nx; ++b) c->F[r * nx + b] = CMS(p->F, nc, r, b);
    std::vector<double> Gam((size_t)SC_MAXN * LBMPC_MAX_NX * SC_MAXN, 0.0), Apow((size_t)SC_MAXN * 16, 0.0);
    auto G3 = [&](int i, int a, int j) -> double& { return Gam[((size_t)i * LBMPC_MAX_NX + a) * SC_MAXN + j]; };
    for (int a = 0; a < nx; ++a) for (int b = 0; b < nx; ++b) Apow[a * nx + b] = (a == b) ? 1.0 : 0.0;
    for (int t = 1; t < N; ++t)
        for (int a = 0; a < nx; ++a) for (int b = 0; b < nx; ++b) {
            double acc = 0.0;
            for (int q = 0; q < nx; ++q) acc = std::fma(Apow[(size_t)(t - 1) * 16 + a * nx + q], c->A[q * nx + b], acc);
            Apow[(size_t)t * 16 + a * nx + b] = acc;
        }
    for (int i = 0; i < N; ++i)
        for (int j = 0; j <= i; ++j)
            for (int a = 0; a < nx; ++a) for (int l = 0; l < nu; ++l) {
                double acc = 0.0;
                for (int q = 0; q < nx; ++q) acc = std::fma(Apow[(size_t)(i - j) * 16 + a * nx + q], c->B[q * nu + l], acc);
                G3(i, a, j * nu + l) = acc;
            }
    std::vector<double> P2((size_t)n * n, 0.0);
    for (int i = 0; i < N; ++i) {
        for (int j = 0; j < n; ++j) for (int a = 0; a < nx; ++a) {
            double acc = 0.0;
            for (int q = 0; q < nx; ++q) acc = std::fma(G3(i, q, j), CMS(p->Q, nx, q, a), acc);
            c->GQ[i][j][a] = acc;
        }
        for (int j = 0; j < n; ++j) for (int l = 0; l < n; ++l) {
            double acc = 0.0;
            for (int a = 0; a < nx; ++a) acc = std::fma(c->GQ[i][j][a], G3(i, a, l), acc);
            P2[(size_t)j * n + l] += acc;
        }
        for (int j = 0; j < nu; ++j) for (int l = 0; l < nu; ++l) P2[(size_t)(i * nu + j) * n + i * nu + l] += CMS(p->R, nu, j, l);
        for (int r = 0; r < nc; ++r) for (int j = 0; j < n; ++j) {
            double acc = 0.0;
            for (int a = 0; a < nx; ++a) acc = std::fma(c->F[r * nx + a], G3(i, a, j), acc);
            if (j / nu == i) acc += CMS(p->G, nc, r, j % nu);
            c->Cm[(size_t)(i * nc + r) * n + j] = acc;
        }
    }
    for (auto& v : P2) v *= 2.0 * (double)p->N_SC;
    if (!invert_gj_sc(n, P2.data(), c->Ginv)) { set_error("lbmpc_scmpc_mc_run: the scenario cost matrix is singular"); return LBMPC_ERR_ARG; }
    for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) { const double s = 0.5 * (c->Ginv[i * n + j] + c->Ginv[j * n + i]); c->Ginv[i * n + j] = c->Ginv[j * n + i] = s; }
    return 0;
}

}  // namespace lbmpc

using namespace lbmpc;

extern "C" int lbmpc_scmpc_mc_run(const lbmpc_scmpc_params* p, const lbmpc_sampler* smp, const lbmpc_mc_cfg* cfg, lbmpc_scmpc_out* out) {
    if (int rc = check_device()) return rc;
    if (!p || !smp || !cfg || !out || !cfg->x0 || !p->A || !p->B || !p->Q || !p->R || !p->F || !p->G) { set_error("lbmpc_scmpc_mc_run: null argument"); return LBMPC_ERR_ARG; }
    if (cfg->trials < 0 || cfg->steps < 0 || cfg->traj_trials < 0 || cfg->sampler_max_attempts < 1) { set_error("lbmpc_scmpc_mc_run: bad configuration"); return LBMPC_ERR_ARG; }
    SamplerDev sd;
    if (int rc = pack_sampler(smp, &sd)) return rc;
    if (sd.nx != p->nx || sd.nu != p->nu) { set_error("lbmpc_scmpc_mc_run: sampler dims differ from the controller's"); return LBMPC_ERR_ARG; }
    std::vector<ScConst> hc(1);
    if (int rc = sc_setup(p, hc.data())) return rc;
    if (cfg->trials == 0) return 0;
    if ((int64_t)cfg->steps * p->N_SC * p->N > 0x7fffffffLL) { set_error("lbmpc_scmpc_mc_run: steps * N_SC * N exceeds the 32-bit draw index"); return LBMPC_ERR_DIM; }
    const size_t Tn = (size_t)cfg->trials, nx = p->nx, nu = p->nu, K = (size_t)cfg->steps;
    const size_t R = (size_t)(cfg->traj_trials < cfg->trials ? cfg->traj_trials : cfg->trials);
    struct Out { void* host; size_t bytes; DevBuf buf; };
    Out o[6] = {{out->infeasible, Tn * 4, {}}, {out->status, Tn * 4, {}}, {out->qp_iters, Tn * 4, {}}, {out->x_final, Tn * nx * 8, {}},
                {out->traj_x, R * (K + 1) * nx * 8, {}}, {out->traj_u, R * K * nu * 8, {}}};
    for (auto& x : o) if (x.host) LB_HIP(x.buf.alloc(x.bytes));
    DevBuf dc;
    LB_HIP(dc.alloc(sizeof(ScConst)));
    LB_HIP(hipMemcpy(dc.p, hc.data(), sizeof(ScConst), hipMemcpyHostToDevice));
    ScArgs a;
    a.trials = cfg->trials; a.trial0 = cfg->trial0; a.steps = cfg->steps; a.traj_trials = (int)R; a.max_attempts = cfg->sampler_max_attempts;
    a.seed = cfg->seed; a.acc_bound = cfg->acc_bound; a.distance_error = cfg->distance_error;
    for (int k = 0; k < LBMPC_MAX_NX; ++k) a.x0[k] = k < p->nx ? cfg->x0[k] : 0.0;
    a.out.infeasible = o[0].buf.as<int32_t>(); a.out.status = o[1].buf.as<int32_t>(); a.out.qp_iters = o[2].buf.as<int32_t>();
    a.out.x_final = o[3].buf.as<double>(); a.out.traj_x = o[4].buf.as<double>(); a.out.traj_u = o[5].buf.as<double>();
    dim3 grid((unsigned)cfg->trials), blk(64);   // one trial per wave
    switch (p->nx) {
        case 1: hipLaunchKernelGGL(scmpc_kernel<1>, grid, blk, 0, nullptr, dc.as<ScConst>(), sd, a); break;
        case 2: hipLaunchKernelGGL(scmpc_kernel<2>, grid, blk, 0, nullptr, dc.as<ScConst>(), sd, a); break;
        case 3: hipLaunchKernelGGL(scmpc_kernel<3>, grid, blk, 0, nullptr, dc.as<ScConst>(), sd, a); break;
        default: hipLaunchKernelGGL(scmpc_kernel<4>, grid, blk, 0, nullptr, dc.as<ScConst>(), sd, a); break;
    }
    LB_HIP(hipGetLastError());
    LB_HIP(hipDeviceSynchronize());
    for (auto& x : o) if (x.host) LB_HIP(hipMemcpy(x.host, x.buf.p, x.bytes, hipMemcpyDeviceToHost));
    return 0;
}
