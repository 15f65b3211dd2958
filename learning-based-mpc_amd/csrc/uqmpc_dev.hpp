// uqmpc_dev.hpp -- shared by uqmpc.hip (step kernel, transient closed loop, host) and mc_steady.hip (disturbance
// pre-pass, steady-state closed loop): controller constants, lane-per-trial device pieces, closed-loop launch arguments.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.hpp"
#include "homothety_gen.hpp"

namespace lbmpc {

#define QP_TOL_VIOL 1e-10
#define QP_TOL_DEP 1e-13
#define QP_TOL_R 1e-12
#define QP_MAX_ITERS 200
#define NZMAX LBMPC_MAX_NZ

struct CtxDev {
    int nx, nu, N, nc, Nu, nS, nW, nz, nH, RL, nzp, nSp;  // nSp: S row storage padded to a multiple of 128
    int nHs;             // H rows stored (nH, + one row of zeros when gen_ok)
    double A[LBMPC_MAX_NX * LBMPC_MAX_NX];    // row-major
    double B[LBMPC_MAX_NX * LBMPC_MAX_NU];    // row-major nx x nu
    double K[LBMPC_MAX_NU * LBMPC_MAX_NX];    // row-major nu x nx
    double T[LBMPC_MAX_NX * LBMPC_MAX_NX];    // (I-Phi)^-1 row-major
    double FcT[LBMPC_MAX_NC * LBMPC_MAX_NX];  // (F+GK) T row-major
    double hs[LBMPC_MAX_NC];
    BoxW box;
    int genW;            // 1: W is not an axis-aligned box: the set update solves the LP by vertex enumeration (gw; box is unused)
    GenW gw;
    const double* Srow;  // nSp x RL : [S_b[j], S_A[j,0..nx-1]], RL = nx + 1
    const float* S32;    // fp32 copy for the first-pass filter, row PAIRS interleaved: [b_j b_j+1 a_j0 a_j+1,0 ...]
    double amax[LBMPC_MAX_NX], bmax;  // max_j |S_A[j,k]|, max_j |S_b[j]|: rounding margin of the filter
    // inscribed ellipsoid E = {x : x' ME x <= 1} of S (centred at 0): q' ME q <= (alpha (1-1e-6))^2 proves that no S
    // row is violated at z = 0 (Cauchy-Schwarz: |a_j.q| <= sqrt(a_j' ME^-1 a_j) sqrt(q' ME q) <= b_j alpha (1-1e-6))
    double ME[LBMPC_MAX_NX * LBMPC_MAX_NX], bmin;
    double cmH[LBMPC_MAX_NZ];  // max_h |H[h,k]|: certificate that no H row is violated (qp_group.hpp, scan_rows)
    // the same bound per chunk of 64 H rows (the rows of late prediction stages shrink like rho(Phi)^i: most chunks certify
    // even when the bound over all rows does not); used by the full scans of qp_lane_gen.hpp
    double cmHc[32][LBMPC_MAX_NZ];
    double GiX[LBMPC_MAX_NX * LBMPC_MAX_NX], Gd[LBMPC_MAX_NZ];  // (2 Px)^-1 row-major nx x nx, diagonal of (2 Pc)^-1 (qp_lane_gen.hpp)
    int hasE;
    int lane_solver;     // 1: try the one-QP-per-lane solver of the S-row regime first (qp_lane.hpp)
    int gen_ok;          // 1: the one-QP-per-lane solver with H rows (qp_lane_gen.hpp) applies: nz = nx + 8, Ginv block diagonal with a diagonal Pc block
    const double* Hm;    // nH x nzp: rows of F_bar Psi^i, i = 0..Nu
    const double* Ginv;  // nz x nz : (2 blkdiag(Px,Pc))^-1
};

}  // namespace lbmpc

struct lbmpc_ctx {
    lbmpc::CtxDev h;          // host copy (device pointers inside)
    lbmpc::CtxDev* d = nullptr;
    double *dS = nullptr, *dH = nullptr, *dG = nullptr;
    float* dS32 = nullptr;
    int device = 0;
    size_t lds_const_bytes = 0;
    // grow-only scratch of lbmpc_mc_run: pre-drawn step disturbances of one trial chunk (+ their cap flags)
    double* xi = nullptr;
    size_t xi_bytes = 0;
    int* xi_status = nullptr;
    size_t xi_status_n = 0;
    unsigned long long* counters = nullptr;  // MC_STATS_N work counters + histograms of the last mc_run
    int* hint = nullptr;                     // pivot candidates, 17 ints per chunk trial (16 rows + count)
    size_t hint_n = 0;
    double* st_d = nullptr;                  // trial state handed between step segments: (3 nx + 1) doubles ...
    int* st_i = nullptr;                     // ... and 5 ints per chunk trial
    size_t st_n = 0;
    lbmpc::SamplerDev* smp_d = nullptr;      // device copy of the sampler of the running lbmpc_mc_run (mc_sample_kernel stages it into LDS)
    double* m0 = nullptr;                    // initial information sets of one trial chunk: maxima over the card(I0) samples, 2 nx (canonical rows) per trial
    int* i0_status = nullptr;                // ... and their sampler-cap flags
    size_t m0_n = 0;
    lbmpc::SmpPhases* smp_ph = nullptr;      // draw descriptors of the information-set pre-pass (sample_sets_kernel)
    void *stg_d = nullptr, *stg_h = nullptr; // staging of lbmpc_uqmpc_step: one device block, one pinned host block (grow-only)
    size_t stg_bytes = 0;
    size_t step_lds_set = 0;                 // dynamic-LDS attribute already granted to this context's step kernel
    double* park = nullptr;                  // state parked during solves, PARK_SLOTS doubles per grid thread
    size_t park_bytes = 0;
};

namespace lbmpc {

struct LdsView {
    const double* Srow;
    const double* Hm;
    const double* Ginv;
    const float* S32;
};
// LDS layout (doubles): [S rows | H rows | Ginv | pad to even | S32 (floats) | per-wave solver slots]
__host__ __device__ inline int lds_const_doubles(const CtxDev& h) {
    const int d64 = (h.nSp * h.RL + h.nHs * h.nzp + h.nz * h.nz + 1) & ~1;
    return d64 + (h.nSp * h.RL + 1) / 2;
}
__host__ __device__ inline int ws_doubles(int) { return 4 * 48; }  // GWS_PER_WAVE: four row slots per wave
#define MC_NCTR 20        // work counters kept per wave in registers
#define MC_STATS_N 160    // u64 slots of lbmpc_ctx::counters
#define MC_PROG_INTS 8    // static LDS of prio_tick (lb_prog), counted against the LDS limit
#define MC_HIST_INTS 128  // per-block LDS histograms (ints): iterations per QP solve [0, 64), final active-set size [64, 128)

}  // namespace lbmpc

namespace lbmpc {

// ---- lane-per-trial pieces ----------------------------------------------------------------------

// W_A[r,:] . x for canonical row r.  For an exact box (every off-axis coefficient an exact zero: all reference
// configurations) the fma chain of the dense form collapses to its one non-zero term, fma(a, x_k, 0.0), bit for bit
// (the other terms add exact zeros; a zero product gives +0.0 in both forms).
template <int NX>
__device__ __forceinline__ double box_row_dot(const CtxDev* __restrict__ C, int r, const double* x) {
    if (C->box.exact) return fma(C->box.WA[r * NX + (r >> 1)], x[r >> 1], 0.0);
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < NX; ++k) t = fma(C->box.WA[r * NX + k], x[k], t);
    return t;
}

// b_hat = alpha W_b + (1-alpha) W_A v (canonical row order)
template <int NX>
__device__ __forceinline__ void w_hat_b(const CtxDev* __restrict__ C, double alpha, const double* v, double* bh) {
#pragma unroll
    for (int r = 0; r < 2 * NX; ++r) {
        const double t = box_row_dot<NX>(C, r, v);
        bh[r] = alpha * C->box.Wb[r] + (1.0 - alpha) * t;
    }
}

// Online set update LPforUQ (UQRobustMPC.m:124-139): homothety LP on m = max(b_hat_prev, W_A w_new);
// (alpha, v) unchanged when w_new is inside W_hat_prev.
template <int NX>
__device__ __forceinline__ int uq_update(const CtxDev* __restrict__ C, double& alpha, double* v, const double* w, double* bh,
                                         bool* changed = nullptr) {
    double mm[2 * NX];
    bool outside = false;
    int st = 0;
    w_hat_b<NX>(C, alpha, v, bh);
#pragma unroll
    for (int r = 0; r < 2 * NX; ++r) {
        const double hw = box_row_dot<NX>(C, r, w);
        mm[r] = bh[r];
        if (hw > bh[r]) { mm[r] = hw; outside = true; }
    }
    if (outside) {
        uint32_t mask;
        st = homothety_box<NX>(&C->box, mm, alpha, v, mask);
        w_hat_b<NX>(C, alpha, v, bh);
    }
    if (changed) *changed = outside;
    return st;
}

// The same for a general-polytope prior (CtxDev::genW; rows in the caller's order): b_hat = alpha W_b + (1-alpha) W_A v,
// m = max(b_hat, W_A w_new), homothety LP by vertex enumeration when w_new is outside (homothety_gen.hpp; oracle:
// step_one with orc_homothety_general).  Kept out of line and fed by value so that the box kernels' register
// allocation does not see it; bh_out (nW values in global memory, or null) receives b_hat of the returned set.
struct UqRes { double alpha, v[LBMPC_MAX_NX]; int st, changed; };
__device__ __noinline__ __attribute__((cold)) UqRes genw_uq_update(const CtxDev* __restrict__ C, int uq, double alpha, double v0, double v1, double v2,
                                             double v3, double w0, double w1, double w2, double w3, double* bh_out) {
    const int nW = C->gw.nW, nx = C->gw.nx;
    double v[LBMPC_MAX_NX] = {v0, v1, v2, v3}, w[LBMPC_MAX_NX] = {w0, w1, w2, w3}, bh[LBMPC_MAX_NW], mm[LBMPC_MAX_NW];
    bool outside = false;
    int st = 0;
    for (int j = 0; j < nW; ++j) {
        double t = 0.0;
        for (int k = 0; k < nx; ++k) t = fma(C->gw.WA[j * nx + k], v[k], t);
        bh[j] = alpha * C->gw.Wb[j] + (1.0 - alpha) * t;
    }
    if (uq) {
        for (int j = 0; j < nW; ++j) {
            double hw = 0.0;
            for (int k = 0; k < nx; ++k) hw = fma(C->gw.WA[j * nx + k], w[k], hw);
            mm[j] = bh[j];
            if (hw > bh[j]) { mm[j] = hw; outside = true; }
        }
        if (outside) {
            uint32_t mask;
            st = homothety_general_dev(C->gw, mm, alpha, v, mask);
            for (int j = 0; j < nW; ++j) {
                double t = 0.0;
                for (int k = 0; k < nx; ++k) t = fma(C->gw.WA[j * nx + k], v[k], t);
                bh[j] = alpha * C->gw.Wb[j] + (1.0 - alpha) * t;
            }
        }
    }
    if (bh_out) for (int j = 0; j < nW; ++j) bh_out[j] = bh[j];
    UqRes r;
    r.alpha = alpha; r.st = st; r.changed = outside ? 1 : 0;
    for (int k = 0; k < LBMPC_MAX_NX; ++k) r.v[k] = k < nx ? v[k] : 0.0;
    return r;
}
// initial set of a closed-loop trial from the row maxima of its information set (InitialSetComputation.m:48-59)
__device__ __noinline__ __attribute__((cold)) UqRes genw_initial(const CtxDev* __restrict__ C, const double* __restrict__ m0) {
    double m[LBMPC_MAX_NW], v[LBMPC_MAX_NX];
    for (int j = 0; j < C->gw.nW; ++j) m[j] = m0[j];
    UqRes r;
    uint32_t mask;
    r.st = homothety_general_dev(C->gw, m, r.alpha, v, mask);
    r.changed = 1;
    for (int k = 0; k < LBMPC_MAX_NX; ++k) r.v[k] = k < C->gw.nx ? v[k] : 0.0;
    return r;
}

// q = (1-alpha) T v - x ; returns min_r (1 - h_k[r]) and its first index (h_k closed form :80-84)
template <int NX>
__device__ __forceinline__ void tighten(const CtxDev* __restrict__ C, double alpha, const double* v, const double* x,
                                        double* q, double& min_omh, int& arg_omh) {
    const double beta = 1.0 - alpha;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) acc = fma(C->T[i * NX + k], v[k], acc);
        q[i] = beta * acc - x[i];
    }
    min_omh = LB_INF; arg_omh = 0;
    for (int r = 0; r < C->nc; ++r) {
        double g = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) g = fma(C->FcT[r * NX + k], v[k], g);
        double omh = 1.0 - (alpha * C->hs[r] + beta * g);
        if (omh < min_omh) { min_omh = omh; arg_omh = r; }
    }
}

// the part of tighten() that depends on (alpha, v) only: bTv = (1-alpha) T v, min_r (1 - h_k[r]) and its index
template <int NX>
__device__ __forceinline__ void tighten_set_rows(const double* __restrict__ Tm, const double* __restrict__ FcT,
                                                 const double* __restrict__ hs, int nc, double alpha, const double* v,
                                                 double* bTv, double& min_omh, int& arg_omh) {
    const double beta = 1.0 - alpha;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) acc = fma(Tm[i * NX + k], v[k], acc);
        bTv[i] = beta * acc;
    }
    min_omh = LB_INF; arg_omh = 0;
    for (int r = 0; r < nc; ++r) {
        double g = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) g = fma(FcT[r * NX + k], v[k], g);
        double omh = 1.0 - (alpha * hs[r] + beta * g);
        if (omh < min_omh) { min_omh = omh; arg_omh = r; }
    }
}
template <int NX>
__device__ __forceinline__ void tighten_set(const CtxDev* __restrict__ C, double alpha, const double* v, double* bTv,
                                            double& min_omh, int& arg_omh) {
    tighten_set_rows<NX>(C->T, C->FcT, C->hs, C->nc, alpha, v, bTv, min_omh, arg_omh);
}

// Per-step constants of the steady-state loop held in VECTOR registers (exact box W, nu = 1): read once through a
// lane-opaque index, so that the loop body has no scalar loads to wait for (the compiler cannot hoist the
// `C->` loads itself: the kernel's stores may alias them).  Same values, same operation order as the C-> forms.
// The constants of the rarer branches (set moved: W_b, 1/|axis|, T, (F+GK)T, hs) stay in LDS (steady_lds_doubles).
template <int NX>
struct StepConst {
    double wa[2 * NX], ME[NX * NX], A[NX * NX], K[NX], B[NX], bmin;
    int hasE;
};
// LDS layout of mc_steady_kernel behind the fp32 S rows (doubles): T | FcT | hs | W_b | ic
template <int NX> struct SteadyLds {
    static constexpr int T = 0, FcT = NX * NX, hs = FcT + LBMPC_MAX_NC * NX, wb = hs + LBMPC_MAX_NC, ic = wb + 2 * NX,
                         total = ic + 2 * NX;
};
__host__ inline size_t steady_lds_bytes(int nSp, int RL, int nx) {
    const size_t d = (size_t)(nSp * RL + 1) / 2 + (size_t)(nx * nx + LBMPC_MAX_NC * nx + LBMPC_MAX_NC + 4 * nx);
    return (d * 8 + 15) / 16 * 16;
}
template <int NX>
__device__ __forceinline__ void load_step_const(const CtxDev* __restrict__ C, StepConst<NX>& s) {
    int vz;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vz));   // a zero the compiler takes for a per-lane value
#pragma unroll
    for (int r = 0; r < 2 * NX; ++r) s.wa[r] = C->box.WA[r * NX + (r >> 1) + vz];
#pragma unroll
    for (int i = 0; i < NX * NX; ++i) s.A[i] = C->A[i + vz];
    // M_E is stored bitwise symmetric (inscribed_ellipsoid): one register per pair
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int k = i; k < NX; ++k) { const double t = C->ME[i * NX + k + vz]; s.ME[i * NX + k] = t; s.ME[k * NX + i] = t; }
#pragma unroll
    for (int i = 0; i < NX; ++i) { s.K[i] = C->K[i + vz]; s.B[i] = C->B[i + vz]; }   // nu = 1: K row 0, B column 0
    s.bmin = C->bmin;
    s.hasE = C->hasE;
}
// b_hat = alpha W_b + (1-alpha) W_A v for an exact box, W_b from LDS
template <int NX>
__device__ __forceinline__ void w_hat_b_regs(const StepConst<NX>& s, const double* wb, double alpha, const double* v, double* bh) {
#pragma unroll
    for (int r = 0; r < 2 * NX; ++r) {
        const double t = fma(s.wa[r], v[r >> 1], 0.0);
        bh[r] = alpha * wb[r] + (1.0 - alpha) * t;
    }
}
// uq_update with b_hat of the current (alpha, v) KEPT by the caller (bh in, bh out): it only changes with (alpha, v)
template <int NX>
__device__ __forceinline__ int uq_update_regs(const StepConst<NX>& s, const double* tcon, double& alpha, double* v,
                                              const double* w, double* bh, bool* changed) {
    double mm[2 * NX];
    bool outside = false;
    int st = 0;
#pragma unroll
    for (int r = 0; r < 2 * NX; ++r) {
        const double hw = fma(s.wa[r], w[r >> 1], 0.0);
        mm[r] = bh[r];
        if (hw > bh[r]) { mm[r] = hw; outside = true; }
    }
    if (outside) {
        uint32_t mask;
        st = homothety_box_rows<NX>(tcon + SteadyLds<NX>::wb, tcon + SteadyLds<NX>::ic, mm, alpha, v, mask);
        w_hat_b_regs<NX>(s, tcon + SteadyLds<NX>::wb, alpha, v, bh);
    }
    *changed = outside;
    return st;
}
template <int NX>
__device__ __forceinline__ bool ellipsoid_outside_regs(const StepConst<NX>& s, double alpha, const double* q) {
    double e2 = 0.0;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        double ti = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) ti = fma(s.ME[i * NX + k], q[k], ti);
        e2 = fma(q[i], ti, e2);
    }
    const double as = alpha * (1.0 - 1e-6);
    const bool inside = s.hasE && alpha * 1e-6 * s.bmin >= 1e-9 && e2 <= as * as;
    return !inside;
}

template <int NX>
__device__ __forceinline__ double hk_value(const CtxDev* __restrict__ C, double alpha, const double* v, int r) {
    double g = 0.0;
#pragma unroll
    for (int k = 0; k < NX; ++k) g = fma(C->FcT[r * NX + k], v[k], g);
    return alpha * C->hs[r] + (1.0 - alpha) * g;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// One chunk of 2*PAIRS rows (PAIRS interleaved row pairs), packed fp32: per pair 1 pk_mul + NX pk_fma + 1 min3.
template <int NX, int T, int PAIRS = 4>
__device__ __forceinline__ void first_pass_eval(const float4* cur, const float* af, const float (*qf)[NX], float* mn) {
    constexpr int RLc = NX + 1, CH = PAIRS * RLc / 2;  // float4 loads per chunk
    float cf[2 * PAIRS * RLc];
#pragma unroll
    for (int i = 0; i < CH; ++i) { cf[4 * i] = cur[i].x; cf[4 * i + 1] = cur[i].y; cf[4 * i + 2] = cur[i].z; cf[4 * i + 3] = cur[i].w; }
#pragma unroll
    for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int pr = 0; pr < PAIRS; ++pr) {
            const float* r = cf + pr * 2 * RLc;
            f32x2 acc = (f32x2){af[t], af[t]} * (f32x2){r[0], r[1]};
#pragma unroll
            for (int k = 0; k < NX; ++k)
                acc = __builtin_elementwise_fma((f32x2){r[2 + 2 * k], r[3 + 2 * k]}, (f32x2){qf[t][k], qf[t][k]}, acc);
            mn[t] = __builtin_fminf(__builtin_fminf(acc.x, acc.y), mn[t]);
        }
    }
}

// First stage of the first-pass filter, the inscribed ellipsoid of S (CtxDev::ME): outside[t] = the trial is NOT
// proven free of violated S rows.
template <int NX, int T>
__device__ __forceinline__ bool ellipsoid_outside(const CtxDev* __restrict__ C, const double* alpha, const double (*q)[NX],
                                                  bool* outside) {
    bool any = false;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        double e2 = 0.0;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            double ti = 0.0;
#pragma unroll
            for (int k = 0; k < NX; ++k) ti = fma(C->ME[i * NX + k], q[t][k], ti);
            e2 = fma(q[t][i], ti, e2);
        }
        const double as = alpha[t] * (1.0 - 1e-6);
        // slack_j >= alpha 1e-6 b_j must clear the tolerance and the rounding of the exact evaluation
        const bool inside = C->hasE && alpha[t] * 1e-6 * C->bmin >= 1e-9 && e2 <= as * as;
        outside[t] = !inside;
        any = any || !inside;
    }
    return any;
}

// Second stage, the packed-fp32 scan of the S rows at z = 0 for the T trials of this lane: may any S row be violated?  slack_j = alpha S_b[j] + S_A[j,:].q ; the
// H rows contribute exactly 1-h_k[r] (min_omh, exact).  This is only a FILTER in front of the exact solver (whose
// own fp64 scan decides, and returns z = 0 after zero iterations when nothing is violated), so it runs in packed
// fp32 -- every row pair is one LDS broadcast shared by the 64*T trials of the wave -- and flags a trial whenever
// the fp32 minimum is below -tol + margin, margin = 2^-20 (alpha max|S_b| + sum_k max|S_A[:,k]| |q_k|) >= twice the
// worst-case rounding error (gamma_7, u = 2^-24) of any row.  A violated row is therefore never missed; a false
// alarm costs one scan.  NaN / overflow compare as "flagged".
// PAIRS: row pairs per chunk (4: the closed-loop kernels; 2: half the registers, for mc_steady_kernel where the scan is rare)
template <int NX, int T, int PAIRS = 4>
__device__ __forceinline__ void filter_scan(const CtxDev* __restrict__ C, const LdsView& L, const double* alpha,
                                            const double (*q)[NX], bool* need) {
    constexpr int RLc = NX + 1;
    constexpr int CH = PAIRS * RLc / 2;
    static_assert(PAIRS % 2 == 0, "a chunk is a whole number of float4");
    float mn[T], af[T], qf[T][NX];
    double thresh[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        mn[t] = __builtin_inff();
        af[t] = (float)alpha[t];
        double e = fabs(alpha[t]) * C->bmax;
#pragma unroll
        for (int k = 0; k < NX; ++k) { qf[t][k] = (float)q[t][k]; e = fma(C->amax[k], fabs(q[t][k]), e); }
        thresh[t] = -QP_TOL_VIOL + 0x1p-20 * e;
    }
    const int nch = C->nSp / (2 * PAIRS);   // chunks of 2*PAIRS rows; even: nSp is a multiple of 128
    const float4* p = reinterpret_cast<const float4*>(L.S32);
    float4 bufA[CH], bufB[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) bufA[i] = p[i];
    // ping-pong: the next chunk is in flight while the current one is evaluated
    for (int c = 0; c < nch; c += 2) {
        const float4* pb = p + (size_t)(c + 1) * CH;
#pragma unroll
        for (int i = 0; i < CH; ++i) bufB[i] = pb[i];
        __builtin_amdgcn_sched_barrier(0);
        first_pass_eval<NX, T, PAIRS>(bufA, af, qf, mn);
        const float4* pa = p + (size_t)(c + 2 < nch ? c + 2 : 0) * CH;
#pragma unroll
        for (int i = 0; i < CH; ++i) bufA[i] = pa[i];
        __builtin_amdgcn_sched_barrier(0);
        first_pass_eval<NX, T, PAIRS>(bufB, af, qf, mn);
    }
#pragma unroll
    for (int t = 0; t < T; ++t) need[t] = !((double)mn[t] >= thresh[t]);
}

template <int NX, int T>
__device__ __forceinline__ bool first_pass(const CtxDev* __restrict__ C, const LdsView& L, const double* alpha,
                                           const double (*q)[NX], const double* min_omh, bool* need) {
    bool outside[T];
    const bool any = ellipsoid_outside<NX, T>(C, alpha, q, outside);
    bool scanned = false;
#pragma unroll
    for (int t = 0; t < T; ++t) need[t] = false;
    if (__builtin_amdgcn_ballot_w64(any) != 0) {
        filter_scan<NX, T>(C, L, alpha, q, need);
        scanned = true;
    }
#pragma unroll
    for (int t = 0; t < T; ++t) need[t] = need[t] || (min_omh[t] < -QP_TOL_VIOL);
    return scanned;  // the rows were scanned
}

// ---- closed-loop Monte-Carlo kernel ---------------------------------------------------------------
struct McArgs {
    int64_t trials, trial0;
    int steps, card_I0;
    uint64_t seed;
    double x0[LBMPC_MAX_NX];
    double acc_bound, distance_error;
    int traj_trials, sampler_max_attempts;
    int tpw;  // trials per wave (and per t): 4 .. 64
    int gen_lanes;  // lanes of the per-wave LDS region of the general lane solver (power of two <= tpw, 0: none)
    // step segment of this launch: [kseg0, kseg1); resume != 0: every trial continues from its saved state at step
    // st_i[kpos] (trials whose kpos is not inside the segment idle); state is saved at the end when kseg1 < steps
    int kseg0, kseg1, resume;
    double* st_d;  // [3 nx + 1][chunk trial]: x, v, w_new, alpha
    int* st_i;     // [5][chunk trial]: status, infeasible, iters, hashint, kpos
    // this launch covers local trials [chunk0, chunk0 + chunk_n); xi holds their pre-drawn step disturbances
    int64_t chunk0, chunk_n;
    const double* m0;       // [chunk trial][2 nx] maxima of W_A w over the initial information set (canonical rows), from the pre-pass
    const int* i0_status;   // [chunk trial] LBMPC_ST_SAMPLER_CAP if one of its samplers hit the attempt cap
    const double* xi;       // pre-drawn step disturbances, layout xi_index()
    const int* xi_status;   // [chunk trial] LBMPC_ST_SAMPLER_CAP if a step sampler hit the attempt cap
    unsigned long long* counters;  // MC_STATS_N: work counters [0, 32), QP-iteration histogram [32, 96), active-set-size histogram [96, 160)
    int* hint;              // [chunk trial][16] final active set of the trial's previous step (pivot candidates)
    int* nhint;             // [chunk trial] its size
    double* park;           // [slot][thread of the grid]: closed-loop state parked during a solve (mc_kernel)
    lbmpc_mc_out out;
};

// Layout of the pre-drawn step disturbances w = (xi_EV - xi_LV) - B u_LV: [step][chunk trial][nx] with chunk trials
// padded to a multiple of 4.  A closed-loop step reads one contiguous 64*nx*8-byte window per wave; the pre-pass
// writes the same window (one trial per lane).
template <int NX>
__host__ __device__ inline int64_t xi_index(int64_t cl, int k, int64_t chunk_n) {
    const int64_t np = (chunk_n + 3) & ~(int64_t)3;
    return ((int64_t)k * np + cl) * NX;
}
__host__ inline size_t xi_bytes_for(int nx, int K, int64_t chunk_n) {
    return (size_t)K * (size_t)((chunk_n + 3) & ~(int64_t)3) * nx * 8;
}

// launch wrappers of the kernels in mc_steady.hip
int launch_mc_sample(int nx, int lv_hasU, int ev_hasU, dim3 grid, dim3 block, hipStream_t st, const SamplerDev* sd_dev, int64_t chunk_n, int64_t trial_first,
                     int K, uint64_t seed, int max_attempts, double* xi, int* xi_status);
int launch_mc_steady(const lbmpc_ctx* c, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const McArgs& a);

}  // namespace lbmpc
