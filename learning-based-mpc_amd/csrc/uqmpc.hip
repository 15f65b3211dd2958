// uqmpc.hip -- subsystem 3: the per-step tightened tube-MPC QP and the batched closed loop.
//
// Replaces Functions_general/UQRobustMPC.m:68-159 (solve, LPforUQ, LP/Com_hs, MPCFormulation:
// 2+nc IPOPT solves per step), NominalRobustMPC.m:47-69, ModelingCar.m:34-91 and the Monte-Carlo
// loops of Cases_general/Case_{3,4,5}.
//
// Execution model (DESIGN.md "Kernel 3"):
//   * mc_sample_kernel (pre-pass, mc_steady.hip): the step disturbances w = (xi_EV - xi_LV) - B u_LV of a trial chunk, one
//     trial per lane, stored step-major so that both its stores and the step loop's loads are contiguous per wave;
//   * mc_kernel / uqmpc_step_kernel, lane per trial (T trials per lane): online set update (closed-form homothety LP),
//     h_k in closed form, then the first pass at the unconstrained minimiser z = 0 as a conservative filter -- inscribed
//     ellipsoid test, then a packed-fp32 scan of the S rows with a rounding margin -- so that only trials that may
//     have a violated row reach the solver;
//   * solver: Goldfarb-Idnani dual active set.  While only S rows are active or violated the QP lives in s0: one QP
//     per lane in registers, scans of the S rows shared four QPs to a pass (qp_lane.hpp); otherwise one QP per 16-lane
//     DPP row, four QPs per wave, factor and KKT vectors one column per lane (qp_group.hpp); closed-loop state is
//     parked in HBM while the solver needs the registers;
//   * u = K x + c_0, plant update, violation flags, lane per trial again;
//   * long horizons run in step segments: mc_kernel for the transient, mc_steady_kernel (mc_steady.hip: no solver) for
//     the steps where no trial needs a QP, and a resume launch of mc_kernel for the trials that did after all.
// Constants ([S_b | S_A] rows in fp64 and fp32, F_bar Psi^i rows, (2P)^-1) are staged once per block into LDS.
#include "uqmpc_dev.hpp"
#include "qp_group.hpp"
#include "qp_lane.hpp"
#include "qp_lane_gen.hpp"

namespace lbmpc {

// Copies the constants into LDS and returns the view; *ws_base = first per-wave solver slot.
__device__ __forceinline__ LdsView stage_constants(const CtxDev* __restrict__ C, double* lds, double** ws_base) {
    const int nS_RL = C->nSp * C->RL, nH_nzp = C->nHs * C->nzp, nzz = C->nz * C->nz;
    const int d64 = (nS_RL + nH_nzp + nzz + 1) & ~1;
    float* s32 = reinterpret_cast<float*>(lds + d64);
    for (int i = threadIdx.x; i < nS_RL; i += blockDim.x) lds[i] = C->Srow[i];
    for (int i = threadIdx.x; i < nH_nzp; i += blockDim.x) lds[nS_RL + i] = C->Hm[i];
    for (int i = threadIdx.x; i < nzz; i += blockDim.x) lds[nS_RL + nH_nzp + i] = C->Ginv[i];
    for (int i = threadIdx.x; i < nS_RL; i += blockDim.x) s32[i] = C->S32[i];
    __syncthreads();
    LdsView L;
    L.Srow = lds; L.Hm = lds + nS_RL; L.Ginv = L.Hm + nH_nzp; L.S32 = s32;
    *ws_base = lds + lds_const_doubles(*C);
    return L;
}

// Solve the QPs of the lanes flagged in `need`, four at a time (one per 16-lane row of the wave).
// Owner lane receives: s0 = z[0..NX), c0 = z[NX..NX+nu), status, iters (+ optional active set / full c).
template <int NX>
__device__ __forceinline__ void solve_flagged(const CtxDev* __restrict__ C, const LdsView& L, double* wws /* wave slot */,
                                              bool need, double alpha, const double* v, const double* q,
                                              double* s0, double* c0, int& status, int& iters,
                                              int* nact_out, int* act_out /* NZMAX, sorted, or null */,
                                              double* c_full /* nz-NX or null */, unsigned* ctr /* [1] rounds [2] trips [3] scans [4] H passes [7] QPs solved here */,
                                              int* hint_base /* [chunk trial][16] pivot candidates in HBM, or null */,
                                              int* nhint_base /* [chunk trial] */, int cl /* this lane's chunk trial */,
                                              long long* prof = nullptr) {
    const int lane = lane_id(), grp = lane >> 4;
    uint64_t mask = __builtin_amdgcn_ballot_w64(need);
    ctr[7] += (unsigned)__builtin_popcountll(mask);
    while (mask) {
        ctr[1] += 1;
        int src[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            src[g] = mask ? (int)__builtin_ctzll(mask) : -1;
            if (mask) mask &= mask - 1;
        }
        // row g works on the problem of lane src[g]
        double a_g = 0.0, q_g[NX];
#pragma unroll
        for (int k = 0; k < NX; ++k) q_g[k] = 0.0;
        int mygrp = -1, cl_g = 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int sl = src[g] < 0 ? 0 : src[g];
            const double ar = readlane_f64(alpha, sl);
            a_g = grp == g ? ar : a_g;
#pragma unroll
            for (int k = 0; k < NX; ++k) { const double qr = readlane_f64(q[k], sl); q_g[k] = grp == g ? qr : q_g[k]; }
            if (lane == src[g]) mygrp = g;
            const int clr = __builtin_amdgcn_readlane(cl, sl);
            cl_g = grp == g ? clr : cl_g;
        }
        const bool live = (grp == 0 ? src[0] : grp == 1 ? src[1] : grp == 2 ? src[2] : src[3]) >= 0;
        if (mygrp >= 0)
            for (int r = 0; r < C->nc; ++r) wws[mygrp * GWS_DOUBLES + r] = 1.0 - hk_value<NX>(C, alpha, v, r);
        wave_sync();
        group_qp_solve<NX>(C, L, wws + grp * GWS_DOUBLES, live, a_g, q_g, ctr + 2,
                           hint_base ? hint_base + (int64_t)cl_g * 16 : nullptr, nhint_base ? nhint_base + cl_g : nullptr, prof);
        wave_sync();
        if (mygrp >= 0) {
            const double* gw = wws + mygrp * GWS_DOUBLES;
            const int* gi = reinterpret_cast<const int*>(gw + 40);
#pragma unroll
            for (int k = 0; k < NX; ++k) s0[k] = gw[16 + k];
#pragma unroll
            for (int j = 0; j < LBMPC_MAX_NU; ++j) if (j < C->nu) c0[j] = gw[16 + NX + j];
            const int st = gi[0], nact = gi[2];
            status = st > status ? st : status;
            iters += gi[1];
            if (nact_out) *nact_out = nact;
            if (act_out) {
                const int* ga = reinterpret_cast<const int*>(gw + 32);
                for (int a = 0; a < NZMAX; ++a) act_out[a] = -1;
                for (int a = 0; a < nact; ++a) {  // ascending insertion
                    int val = ga[a], pos = a;
                    while (pos > 0 && act_out[pos - 1] > val) { act_out[pos] = act_out[pos - 1]; --pos; }
                    act_out[pos] = val;
                }
            }
            if (c_full) for (int k = NX; k < C->nz; ++k) c_full[k - NX] = gw[16 + k];
        }
        wave_sync();
    }
}

// Solve the QPs of the flagged lanes: the one-QP-per-lane solver of the S-row regime first (qp_lane.hpp), then the
// one-QP-per-lane solver with rows of F_bar Psi^i in play (qp_lane_gen.hpp; lgs = this wave's LDS region for glanes
// lanes, glanes = 0: not available), then the 16-lane-row solver for what is left.  Same outputs as solve_flagged.
// ctr: [0] QPs, [1] rounds, [2] trips, [3] scans, [4] H passes of the row solver, [5] lane-solver trips,
// [6] lane-solver scan passes, [7] QPs that went through the row solver, [8] QPs served by lane-solver scans,
// [9..13] general lane solver: trips, S scan passes, H passes, QPs served by scans, factor rebuilds, [14] its QPs.
template <int NX>
__device__ __forceinline__ void solve_lanes(const CtxDev* __restrict__ C, const LdsView& L, double* wws, double* lgs, int glanes, bool need,
                                            double alpha, const double* v, const double* q, double min_omh, double* s0,
                                            double* c0, int& status, int& iters, int* nact_out, int* act_out,
                                            double* c_full, unsigned* ctr, int* hint_base, int* nhint_base, int cl,
                                            long long* prof = nullptr) {
    ctr[0] += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(need));
    bool fb = need;
#ifdef LBMPC_PROFILE_FULL
    const long long tp0 = clock64();
#endif
    if (C->lane_solver) {
        double zf[NX];
        int st_f = 0, it_f = 0, na_f = 0, act_f[NX];
        lane_qp_solve<NX>(C, L, need, alpha, q, min_omh, hint_base ? hint_base + (int64_t)cl * 16 : nullptr,
                          nhint_base ? nhint_base + cl : nullptr, zf, st_f, it_f, na_f, act_f, fb, ctr + 5, prof);
        if (need && !fb) {
#pragma unroll
            for (int k = 0; k < NX; ++k) s0[k] = zf[k];
#pragma unroll
            for (int j = 0; j < LBMPC_MAX_NU; ++j) c0[j] = 0.0;
            status = st_f > status ? st_f : status;
            iters += it_f;
            if (nact_out) *nact_out = na_f;
            if (act_out) {
                for (int a = 0; a < NZMAX; ++a) act_out[a] = -1;
#pragma unroll
                for (int a = 0; a < NX; ++a)
                    if (a < na_f) {  // ascending insertion
                        int pos = a;
                        while (pos > 0 && act_out[pos - 1] > act_f[a]) { act_out[pos] = act_out[pos - 1]; --pos; }
                        act_out[pos] = act_f[a];
                    }
            }
            if (c_full) for (int k = NX; k < C->nz; ++k) c_full[k - NX] = 0.0;
        }
        fb = need && fb;
    }
#ifdef LBMPC_PROFILE_FULL
    const long long tp1 = clock64();
    if (prof) prof[24] += tp1 - tp0;
#endif
    // instantiated for nx = 2, 3 (the 4x4 constants leave no LDS for its state; lbmpc_uqmpc_create clears gen_ok otherwise)
    if constexpr (NX == 2 || NX == 3)
    if (glanes > 0 && __builtin_amdgcn_ballot_w64(fb) != 0) {
        constexpr int NZ = NX + 8;   // the instantiated shape: N * nu = 8 (every reference configuration); checked on the host
        const int lane = lane_id();
        for (int h0 = 0; h0 < LB_WAVE; h0 += glanes) {   // glanes lanes at a time share the wave's LDS region
            const bool nd = fb && lane >= h0 && lane < h0 + glanes;
            if (__builtin_amdgcn_ballot_w64(nd) == 0) continue;
            ctr[14] += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(nd));
            double zf[NZ];
            int st_f = 0, it_f = 0, na_f = 0, act_f[LG_NQL];
            bool fb2 = false;
            lanegen_qp_solve<NX, NZ>(C, L, lgs, glanes, nd, alpha, v, q, min_omh, hint_base ? hint_base + (int64_t)cl * 16 : nullptr,
                                     nhint_base ? nhint_base + cl : nullptr, zf, st_f, it_f, na_f, act_f, fb2, ctr + 9, prof ? prof + 16 : nullptr);
            if (nd && !fb2) {
#pragma unroll
                for (int k = 0; k < NX; ++k) s0[k] = zf[k];
#pragma unroll
                for (int j = 0; j < LBMPC_MAX_NU; ++j) c0[j] = j < C->nu ? zf[NX + j] : 0.0;
                status = st_f > status ? st_f : status;
                iters += it_f;
                if (nact_out) *nact_out = na_f;
                if (act_out) {
                    for (int a = 0; a < NZMAX; ++a) act_out[a] = -1;
#pragma unroll
                    for (int a = 0; a < LG_NQL; ++a)
                        if (a < na_f) {  // ascending insertion
                            int pos = a;
                            while (pos > 0 && act_out[pos - 1] > act_f[a]) { act_out[pos] = act_out[pos - 1]; --pos; }
                            act_out[pos] = act_f[a];
                        }
                }
                if (c_full) {
#pragma unroll
                    for (int k = NX; k < NZ; ++k) c_full[k - NX] = zf[k];
                }
                fb = false;
            }
        }
    }
#ifdef LBMPC_PROFILE_FULL
    const long long tp2 = clock64();
    if (prof) prof[25] += tp2 - tp1;
#endif
    if (__builtin_amdgcn_ballot_w64(fb) != 0)
        solve_flagged<NX>(C, L, wws, fb, alpha, v, q, s0, c0, status, iters, nact_out, act_out, c_full, ctr, hint_base,
                          nhint_base, cl, prof);
#ifdef LBMPC_PROFILE_FULL
    if (prof) prof[26] += clock64() - tp2;
#endif
}

// ---- single-step kernel (lbmpc_uqmpc_step): lane per problem -------------------------------------
struct StepIO {
    const double *x, *alpha_prev, *v_prev, *w_new;
    double *s0, *c, *u, *alpha, *v, *hk, *Sb_hat, *Wb_hat;
    int *status, *n_active, *active_set, *qp_iters;
};

template <int NX>
__global__ __launch_bounds__(256) void uqmpc_step_kernel(const CtxDev* __restrict__ C, int64_t batch, int uq, int gen_lanes, StepIO io) {
    extern __shared__ double lds[];
    double* ws_base;
    const LdsView L = stage_constants(C, lds, &ws_base);
    const int wave = threadIdx.x / LB_WAVE, nwaves = blockDim.x / LB_WAVE, lane = lane_id();
    const int nz = C->nz, ncv = nz - NX;
    double* ws = ws_base + wave * ws_doubles(nz);
    double* lgs = ws_base + nwaves * ws_doubles(nz) + wave * lanegen_doubles(gen_lanes);
    for (int64_t base = ((int64_t)blockIdx.x * nwaves + wave) * LB_WAVE; base < batch;
         base += (int64_t)gridDim.x * nwaves * LB_WAVE) {
        const int64_t b = base + lane;
        const bool valid = b < batch;
        const int64_t bb = valid ? b : batch - 1;
        double x[NX], v[NX], w[NX], q[1][NX], bh[2 * NX], alpha[1], momh[1];
        int aomh[1], status = 0, iters = 0, nact = 0;
        bool need[1];
#pragma unroll
        for (int k = 0; k < NX; ++k) { x[k] = io.x[bb * NX + k]; v[k] = io.v_prev[bb * NX + k]; w[k] = (uq && io.w_new) ? io.w_new[bb * NX + k] : 0.0; }
        alpha[0] = io.alpha_prev[bb];
        if (C->genW) {   // general-polytope prior: out of line (uqmpc_dev.hpp)
            double v4[LBMPC_MAX_NX] = {0.0, 0.0, 0.0, 0.0}, w4[LBMPC_MAX_NX] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k = 0; k < NX; ++k) { v4[k] = v[k]; w4[k] = w[k]; }
            const UqRes r = genw_uq_update(C, uq, alpha[0], v4[0], v4[1], v4[2], v4[3], w4[0], w4[1], w4[2], w4[3],
                                           (valid && io.Wb_hat) ? io.Wb_hat + b * C->gw.nW : nullptr);
            alpha[0] = r.alpha; status = r.st;
#pragma unroll
            for (int k = 0; k < NX; ++k) v[k] = r.v[k];
        } else if (uq) status = uq_update<NX>(C, alpha[0], v, w, bh);
        else w_hat_b<NX>(C, alpha[0], v, bh);
        tighten<NX>(C, alpha[0], v, x, q[0], momh[0], aomh[0]);
        first_pass<NX, 1>(C, L, alpha, q, momh, need);
        double s0[NX], c0[LBMPC_MAX_NU];
#pragma unroll
        for (int k = 0; k < NX; ++k) s0[k] = 0.0;
#pragma unroll
        for (int j = 0; j < LBMPC_MAX_NU; ++j) c0[j] = 0.0;
        double* cdst = (valid && io.c) ? io.c + b * ncv : nullptr;
        if (cdst) for (int k = 0; k < ncv; ++k) cdst[k] = 0.0;
        if (valid && io.active_set) for (int a = 0; a < NZMAX; ++a) io.active_set[b * NZMAX + a] = -1;
        unsigned ctr[MC_NCTR];
#pragma unroll
        for (int i = 0; i < MC_NCTR; ++i) ctr[i] = 0;
        solve_lanes<NX>(C, L, ws, lgs, gen_lanes, valid && need[0], alpha[0], v, q[0], momh[0], s0, c0, status, iters,
                        &nact, (valid && io.active_set) ? io.active_set + b * NZMAX : nullptr, cdst, ctr, nullptr, nullptr, 0);
        if (!valid) continue;
        if (io.s0) for (int k = 0; k < NX; ++k) io.s0[b * NX + k] = s0[k];
        if (io.u)
            for (int j = 0; j < C->nu; ++j) {  // u = K x + c(1:nu)   (UQRobustMPC.m:90)
                double acc = 0.0;
#pragma unroll
                for (int jj = 0; jj < LBMPC_MAX_NU; ++jj) if (jj == j) acc = c0[jj];
#pragma unroll
                for (int k = 0; k < NX; ++k) acc = fma(C->K[j * NX + k], x[k], acc);
                io.u[b * C->nu + j] = acc;
            }
        if (io.alpha) io.alpha[b] = alpha[0];
        if (io.v) for (int k = 0; k < NX; ++k) io.v[b * NX + k] = v[k];
        if (io.hk) for (int r = 0; r < C->nc; ++r) io.hk[b * C->nc + r] = hk_value<NX>(C, alpha[0], v, r);
        if (io.Wb_hat && !C->genW)
#pragma unroll
            for (int r = 0; r < 2 * NX; ++r) io.Wb_hat[b * 2 * NX + C->box.perm[r]] = bh[r];
        if (io.Sb_hat) {  // S_hat = alpha S + (1-alpha) T v   (UQRobustMPC.m:76)
            double Tv[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < NX; ++k) acc = fma(C->T[i * NX + k], v[k], acc);
                Tv[i] = acc;
            }
            for (int j = 0; j < C->nS; ++j) {
                const double* row = L.Srow + j * C->RL;
                double t = 0.0;
#pragma unroll
                for (int k = 0; k < NX; ++k) t = fma(row[1 + k], Tv[k], t);
                io.Sb_hat[b * C->nS + j] = alpha[0] * row[0] + (1.0 - alpha[0]) * t;
            }
        }
        if (io.status) io.status[b] = status;
        if (io.n_active) io.n_active[b] = nact;
        if (io.qp_iters) io.qp_iters[b] = iters;
    }
}

#ifdef LBMPC_PROFILE
__device__ unsigned long long lbmpc_prof[48];
__device__ unsigned long long lbmpc_prof_wave[64 * 4096];  // per wave: [0..47] its phase counters (prof[6] lifetime incl. the final barrier, prof[37] before it), [48..63] work counters ctr[0..15]
#ifdef LBMPC_PROFILE_FULL
#define PROF_T(var) const long long var = clock64()
#define PROF_ADD(slot, t0, t1) prof[slot] += (t1) - (t0)
#define PROF_PTR (prof + 8)
#else   // LBMPC_PROFILE_LIGHT: only the wave's lifetime and its work counters are recorded (two clock reads per wave)
#define PROF_T(var) const long long var = 0
#define PROF_ADD(slot, t0, t1)
#define PROF_PTR nullptr
#endif
#else
#define PROF_PTR nullptr
#define PROF_T(var)
#define PROF_ADD(slot, t0, t1)
#endif

// GW: the instantiation carries the set update for a general-polytope prior W (CtxDev::genW) beside the box form.
// Box contexts use GW = false except for nx = 4, where the instantiation WITH the out-of-line branch measured 3.5 %
// faster on the C4 transient (4.22 -> 4.07 ms, same-box A/B) and 2 % slower on C1 (nx = 2): register allocation luck
// either way, taken where it helps.
template <int NX, int T, bool GW>
__global__ __launch_bounds__(512) void mc_kernel(const CtxDev* __restrict__ C, McArgs a) {
    extern __shared__ double lds[];
    double* ws_base;
    const LdsView L = stage_constants(C, lds, &ws_base);
    const int wave = threadIdx.x / LB_WAVE, nwaves = blockDim.x / LB_WAVE, lane = lane_id();
    const int nz = C->nz, nu = C->nu, K = a.steps;
    double* ws = ws_base + wave * ws_doubles(nz);
    double* lgs = ws_base + nwaves * ws_doubles(nz) + wave * lanegen_doubles(a.gen_lanes);
    int* hist = reinterpret_cast<int*>(ws_base + nwaves * (ws_doubles(nz) + lanegen_doubles(a.gen_lanes)));   // MC_HIST_INTS
    for (int i = threadIdx.x; i < MC_HIST_INTS; i += blockDim.x) hist[i] = 0;
#if LBMPC_PRIO_MODE == 3
    if (threadIdx.x < 8) lb_prog[threadIdx.x] = 0;
#endif
    __syncthreads();
    unsigned ctr[MC_NCTR];  // wave-uniform work counters (lbmpc_mc_counters)
#pragma unroll
    for (int i = 0; i < MC_NCTR; ++i) ctr[i] = 0;
#ifdef LBMPC_PROFILE
    long long prof[48];
    for (int i = 0; i < 48; ++i) prof[i] = 0;
    const long long pk0 = clock64();
#endif
    // a.tpw trials per wave and per t (64 for large batches; fewer lanes per wave for small ones, so that the solver
    // rounds of a step -- 4 QPs at a time -- spread over more waves)
    const int tpw = a.tpw;
    for (int64_t base = ((int64_t)blockIdx.x * nwaves + wave) * (tpw * T); base < a.chunk_n;
         base += (int64_t)gridDim.x * nwaves * (tpw * T)) {
        int64_t tl[T], cl[T];  // local trial index (output arrays), index inside this chunk (xi)
        uint64_t tid[T];    // global id = RNG stream key
        bool valid[T];
        double x[T][NX], wst[T][NX], v[T][NX], alpha[T];
        int status[T], infeasible[T], iters[T];
        bool hashint[T];   // the trial's candidate list in HBM is non-empty
        bool stale[T];     // bTv / momh_c must be recomputed
        int kpos[T];       // first step this trial runs in this launch (resume) / next step to run (at the end)
        bool live[T];      // valid and inside its step range
        double bTv[T][NX], momh_c[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int64_t l = base + t * tpw + lane;
            valid[t] = lane < tpw && l < a.chunk_n;
            cl[t] = valid[t] ? l : a.chunk_n - 1;
            tl[t] = a.chunk0 + cl[t];
            tid[t] = (uint64_t)(a.trial0 + tl[t]);
            status[t] = a.steps > 0 ? a.xi_status[cl[t]] : 0; infeasible[t] = 0; iters[t] = 0; hashint[t] = false; stale[t] = true; momh_c[t] = 0.0;
            kpos[t] = a.resume ? a.st_i[4 * a.chunk_n + cl[t]] : 0;
            alpha[t] = 1.0;
#pragma unroll
            for (int i = 0; i < NX; ++i) { bTv[t][i] = 0.0; x[t][i] = 0.0; v[t][i] = 0.0; wst[t][i] = 0.0; }
            if (a.resume) continue;   // state arrives at step kpos
            // initial set learning: InitialSetComputation.solve (:48-59).  The maxima over the card_I0 samples come from
            // the pre-pass (sample_sets_kernel: lanes walk their own draws instead of waiting for the wave's slowest)
            { const int s0i = a.i0_status[cl[t]]; status[t] = s0i > status[t] ? s0i : status[t]; }
            int st;
            if (GW && C->genW) {
                const UqRes r = genw_initial(C, a.m0 + cl[t] * C->gw.nW);
                alpha[t] = r.alpha; st = r.st;
#pragma unroll
                for (int k = 0; k < NX; ++k) v[t][k] = r.v[k];
            } else {
                double m[2 * NX];
#pragma unroll
                for (int r = 0; r < 2 * NX; ++r) m[r] = a.m0[cl[t] * (2 * NX) + r];
                uint32_t mask;
                st = homothety_box<NX>(&C->box, m, alpha[t], v[t], mask);
            }
            status[t] = st > status[t] ? st : status[t];
#pragma unroll
            for (int k = 0; k < NX; ++k) {
                x[t][k] = a.x0[k]; wst[t][k] = 0.0;
                if (x[t][k] > a.distance_error || x[t][k] < -a.distance_error) infeasible[t] = 1;
            }
            if (valid[t] && a.out.alpha0) a.out.alpha0[tl[t]] = alpha[t];
            if (valid[t] && tl[t] < a.traj_trials) {
                if (a.out.traj_x) for (int k = 0; k < NX; ++k) a.out.traj_x[(tl[t] * (K + 1)) * NX + k] = x[t][k];
                if (a.out.traj_alpha) a.out.traj_alpha[tl[t] * (K + 1)] = alpha[t];
            }
        }
        {   // nothing to do for this wave in this segment?
            bool work = false;
#pragma unroll
            for (int t = 0; t < T; ++t) work = work || (valid[t] && (!a.resume || kpos[t] < a.kseg1));
            if (__builtin_amdgcn_ballot_w64(work) == 0) continue;
        }
#pragma unroll
        for (int t = 0; t < T; ++t) live[t] = valid[t] && !a.resume;
#ifdef LBMPC_PROFILE_FULL
        long long p_last = clock64();
        prof[35] += p_last - pk0;   // prologue (first chunk of the wave)
#endif
        for (int k = a.kseg0; k < a.kseg1; ++k) {
            prio_tick<(NX == 4 ? LBMPC_PRIO_SHIFT4 : LBMPC_PRIO_SHIFT)>(k);
            if (a.resume) {
#pragma unroll
                for (int t = 0; t < T; ++t)
                    if (valid[t] && k == kpos[t]) {  // this trial's saved state takes over from here
                        const int64_t cn = a.chunk_n, c = cl[t];
#pragma unroll
                        for (int i = 0; i < NX; ++i) {
                            x[t][i] = a.st_d[(0 * NX + i) * cn + c]; v[t][i] = a.st_d[(1 * NX + i) * cn + c];
                            wst[t][i] = a.st_d[(2 * NX + i) * cn + c];
                        }
                        alpha[t] = a.st_d[3 * NX * cn + c];
                        status[t] = a.st_i[0 * cn + c]; infeasible[t] = a.st_i[1 * cn + c]; iters[t] = a.st_i[2 * cn + c];
                        hashint[t] = a.st_i[3 * cn + c] != 0;
                        stale[t] = true; live[t] = true;
                    }
            }
            double q[T][NX], wk[T][NX], momh[T];   // wk: this step's disturbance (xi_EV - xi_LV) - B u_LV from the pre-pass
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const double* xw = a.xi + xi_index<NX>(cl[t], k, a.chunk_n);
#pragma unroll
                for (int i = 0; i < NX; ++i) wk[t][i] = xw[i];
            }
            int aomh[T];
            bool need[T];
            PROF_T(p0);
#ifdef LBMPC_PROFILE_FULL
            prof[36] += p0 - p_last;   // between the steps: loop back, resume test, xi loads issued
#endif
#pragma unroll
            for (int t = 0; t < T; ++t) {
                // LVModeling first (Case_3...:58; ModelingCar.m:42-43): xi_LV, xi_EV come from the pre-pass
                PROF_T(pa);
                double bh[2 * NX];
                bool changed = false;
                int st;
                if (GW && C->genW) {
                    double v4[LBMPC_MAX_NX] = {0.0, 0.0, 0.0, 0.0}, w4[LBMPC_MAX_NX] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int i = 0; i < NX; ++i) { v4[i] = v[t][i]; w4[i] = wst[t][i]; }
                    const UqRes r = genw_uq_update(C, 1, alpha[t], v4[0], v4[1], v4[2], v4[3], w4[0], w4[1], w4[2], w4[3], nullptr);
                    alpha[t] = r.alpha; st = r.st; changed = r.changed != 0;
#pragma unroll
                    for (int i = 0; i < NX; ++i) v[t][i] = r.v[i];
                } else st = uq_update<NX>(C, alpha[t], v[t], wst[t], bh, &changed);  // w_new = 0 at k = 0 (Case_3...:62-65)
                status[t] = st > status[t] ? st : status[t];
                PROF_T(pb);
                // (1-alpha) T v and min(1 - h_k) only move when the set update moved (alpha, v): kept between steps
                if (__builtin_amdgcn_ballot_w64(changed || stale[t]) != 0) {
                    tighten_set<NX>(C, alpha[t], v[t], bTv[t], momh_c[t], aomh[t]);
                    stale[t] = false;
                }
                momh[t] = momh_c[t];
#pragma unroll
                for (int i = 0; i < NX; ++i) q[t][i] = bTv[t][i] - x[t][i];
                PROF_T(pc);
                PROF_ADD(16, p0, pa); PROF_ADD(17, pa, pb); PROF_ADD(18, pb, pc);
            }
            PROF_T(p1);
            // first pass, stage 1 (per step, cheap): ellipsoid test; the H rows contribute min(1 - h_k) exactly
            bool outside[T];
            bool heavy = false;
            (void)ellipsoid_outside<NX, T>(C, alpha, q, outside);
            double s0[T][NX], c0[T][LBMPC_MAX_NU];
#pragma unroll
            for (int t = 0; t < T; ++t) {
                need[t] = momh[t] < -QP_TOL_VIOL;
                outside[t] = outside[t] && live[t];
                heavy = heavy || (live[t] && (outside[t] || need[t]));
#pragma unroll
                for (int i = 0; i < NX; ++i) s0[t][i] = 0.0;
#pragma unroll
                for (int j = 0; j < LBMPC_MAX_NU; ++j) c0[t][j] = 0.0;
            }
            ctr[1] += T;
            PROF_T(p2);
            PROF_ADD(0, p0, p1); PROF_ADD(1, p1, p2);
            PROF_T(p3);
            if (__builtin_amdgcn_ballot_w64(heavy) != 0) {
                // Entered in ~1 % of the steps of a long horizon: row filter and QP solves.  The closed-loop state is
                // parked in HBM for the duration (and read back through a laundered pointer, so that the values
                // are new ones): otherwise the register peak of this region makes the allocator spill that state on
                // the steady-state path of EVERY step (14.7 GB of scratch writes per C4 launch before this).
                double* pk = a.park + ((int64_t)blockIdx.x * blockDim.x + threadIdx.x);
                const int64_t ps = (int64_t)gridDim.x * blockDim.x;
                int slot = 0;
#pragma unroll
                for (int tt = 0; tt < T; ++tt) {
#pragma unroll
                    for (int i = 0; i < NX; ++i) {
                        pk[(slot++) * ps] = x[tt][i]; pk[(slot++) * ps] = v[tt][i];
                        pk[(slot++) * ps] = wk[tt][i];
                    }
                    pk[(slot++) * ps] = alpha[tt];
                    pk[(slot++) * ps] = __longlong_as_double(tl[tt]);
                    pk[(slot++) * ps] = __longlong_as_double(cl[tt]);
                    pk[(slot++) * ps] = __longlong_as_double((long long)tid[tt]);
                }
                double a_c[T], v_c[T][NX], q_c[T][NX], mo_c[T];
                int cl_c[T];
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    a_c[t] = alpha[t]; cl_c[t] = (int)cl[t]; mo_c[t] = momh[t];
#pragma unroll
                    for (int i = 0; i < NX; ++i) { v_c[t][i] = v[t][i]; q_c[t][i] = q[t][i]; }
                }
                // stage 2: packed-fp32 scan of the S rows when some trial is outside the ellipsoid
                bool anyout = false;
#pragma unroll
                for (int t = 0; t < T; ++t) anyout = anyout || outside[t];
                if (__builtin_amdgcn_ballot_w64(anyout) != 0) {
                    bool nf[T];
                    filter_scan<NX, T>(C, L, a_c, q_c, nf);
#pragma unroll
                    for (int t = 0; t < T; ++t) need[t] = need[t] || (nf[t] && outside[t]);
                    ctr[0] += T;
                }
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const bool nd = live[t] && need[t];
#ifdef LBMPC_PROFILE_FULL
                    prof[4] += __builtin_popcountll(__builtin_amdgcn_ballot_w64(nd));
                    prof[5] += (__builtin_popcountll(__builtin_amdgcn_ballot_w64(nd)) + 3) >> 2;
#endif
                    if (__builtin_amdgcn_ballot_w64(nd) != 0) {
                        int nact_t = 0;
                        const int it0 = iters[t];
                        solve_lanes<NX>(C, L, ws, lgs, a.gen_lanes, nd, a_c[t], v_c[t], q_c[t], mo_c[t], s0[t], c0[t], status[t], iters[t],
                                        &nact_t, nullptr, nullptr, ctr + 2, a.hint, a.nhint, cl_c[t], PROF_PTR);
                        if (nd) {
                            hashint[t] = nact_t > 0;
                            const int di = iters[t] - it0;
                            atomicAdd(hist + (di < 63 ? di : 63), 1);
                            atomicAdd(hist + 64 + (nact_t < 63 ? nact_t : 63), 1);
                        }
                    }
                }
                asm volatile("" : "+v"(pk));
                slot = 0;
#pragma unroll
                for (int tt = 0; tt < T; ++tt) {
#pragma unroll
                    for (int i = 0; i < NX; ++i) {
                        x[tt][i] = pk[(slot++) * ps]; v[tt][i] = pk[(slot++) * ps];
                        wk[tt][i] = pk[(slot++) * ps];
                    }
                    alpha[tt] = pk[(slot++) * ps];
                    tl[tt] = __double_as_longlong(pk[(slot++) * ps]);
                    cl[tt] = __double_as_longlong(pk[(slot++) * ps]);
                    tid[tt] = (uint64_t)__double_as_longlong(pk[(slot++) * ps]);
                    stale[tt] = true; momh_c[tt] = 0.0;   // not kept across this region: recomputed at the next step
#pragma unroll
                    for (int i = 0; i < NX; ++i) bTv[tt][i] = 0.0;
                }
            }
            PROF_T(p4);
            PROF_ADD(2, p3, p4);
#pragma unroll
            for (int t = 0; t < T; ++t) {
                double u[LBMPC_MAX_NU];
                // no QP this step = the oracle's trivial solve with an empty active set: no candidates next step
                if (live[t] && !need[t] && hashint[t]) { a.nhint[cl[t]] = 0; hashint[t] = false; }
                // u = K x + c(1:nu); EVModeling (ModelingCar.m:36-37); relative dynamics
                double Ax[NX], Bu[NX], xn[NX];
#pragma unroll
                for (int j = 0; j < LBMPC_MAX_NU; ++j) {
                    u[j] = 0.0;
                    if (j < nu) {
                        double acc = c0[t][j];
#pragma unroll
                        for (int i = 0; i < NX; ++i) acc = fma(C->K[j * NX + i], x[t][i], acc);
                        u[j] = acc;
                        if (acc > a.acc_bound || acc < -a.acc_bound) infeasible[t] = 1;
                    }
                }
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    double acc = 0.0, bcc = 0.0;
#pragma unroll
                    for (int kk = 0; kk < NX; ++kk) acc = fma(C->A[i * NX + kk], x[t][kk], acc);
#pragma unroll
                    for (int j = 0; j < LBMPC_MAX_NU; ++j) if (j < nu) bcc = fma(C->B[i * nu + j], u[j], bcc);
                    Ax[i] = acc; Bu[i] = bcc;
                    xn[i] = (Ax[i] + Bu[i]) + wk[t][i];   // relative dynamics x+ = (A x + B u) + w
                }
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    wst[t][i] = (xn[i] - Ax[i]) - Bu[i];  // w_new of the next step (Case_3...:67)
                    x[t][i] = xn[i];
                    if (xn[i] > a.distance_error || xn[i] < -a.distance_error) infeasible[t] = 1;
                }
                if (live[t] && tl[t] < a.traj_trials) {
                    if (a.out.traj_x) for (int i = 0; i < NX; ++i) a.out.traj_x[(tl[t] * (K + 1) + k + 1) * NX + i] = x[t][i];
                    if (a.out.traj_u) for (int j = 0; j < nu; ++j) {
                        double uj = 0.0;
#pragma unroll
                        for (int jj = 0; jj < LBMPC_MAX_NU; ++jj) if (jj == j) uj = u[jj];
                        a.out.traj_u[(tl[t] * K + k) * nu + j] = uj;
                    }
                    if (a.out.traj_alpha) a.out.traj_alpha[tl[t] * (K + 1) + k + 1] = alpha[t];
                    if (a.out.traj_s0) for (int i = 0; i < NX; ++i) a.out.traj_s0[(tl[t] * K + k) * NX + i] = s0[t][i];
                }
                PROF_T(p5);
                PROF_ADD(3, p4, p5);
#ifdef LBMPC_PROFILE_FULL
                p_last = p5;
#endif
            }
        }
#pragma unroll
        for (int t = 0; t < T; ++t) {
            if (!live[t]) continue;
            if (a.kseg1 < K) {  // hand the trial over to the next segment
                const int64_t cn = a.chunk_n, c = cl[t];
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    a.st_d[(0 * NX + i) * cn + c] = x[t][i]; a.st_d[(1 * NX + i) * cn + c] = v[t][i];
                    a.st_d[(2 * NX + i) * cn + c] = wst[t][i];
                }
                a.st_d[3 * NX * cn + c] = alpha[t];
                a.st_i[0 * cn + c] = status[t]; a.st_i[1 * cn + c] = infeasible[t]; a.st_i[2 * cn + c] = iters[t];
                a.st_i[3 * cn + c] = hashint[t] ? 1 : 0; a.st_i[4 * cn + c] = a.kseg1;
                continue;
            }
            const int64_t l = tl[t];
            if (a.out.infeasible) a.out.infeasible[l] = infeasible[t];
            if (a.out.status) a.out.status[l] = status[t];
            if (a.out.qp_iters) a.out.qp_iters[l] = iters[t];
            if (a.out.alpha_final) a.out.alpha_final[l] = alpha[t];
            if (a.out.v_final) for (int i = 0; i < NX; ++i) a.out.v_final[l * NX + i] = v[t][i];
            if (a.out.x_final) for (int i = 0; i < NX; ++i) a.out.x_final[l * NX + i] = x[t][i];
        }
    }
    if (lane == 0 && a.counters)
        for (int i = 0; i < MC_NCTR; ++i) if (ctr[i]) atomicAdd(a.counters + i, (unsigned long long)ctr[i]);
#ifdef LBMPC_PROFILE
    const long long pkb = clock64();
    prof[37] = pkb - pk0;
#endif
    __syncthreads();
    if (a.counters)
        for (int i = threadIdx.x; i < MC_HIST_INTS; i += blockDim.x) if (hist[i]) atomicAdd(a.counters + 32 + i, (unsigned long long)hist[i]);
#ifdef LBMPC_PROFILE
    const long long pk1 = clock64();
    prof[6] = pk1 - pk0;
    if (lane == 0) {
        for (int i = 0; i < 48; ++i) if (i != 7) atomicAdd(&lbmpc_prof[i], (unsigned long long)prof[i]);
        atomicMax(&lbmpc_prof[7], (unsigned long long)prof[6]);  // slowest wave
        const int gw = blockIdx.x * nwaves + wave;
        if (gw < 4096) {
            for (int i = 0; i < 48; ++i) lbmpc_prof_wave[64 * gw + i] = prof[i];
            for (int i = 0; i < 16; ++i) lbmpc_prof_wave[64 * gw + 48 + i] = i < MC_NCTR ? ctr[i] : 0;
        }
    }
#endif
}
#ifdef LBMPC_PROFILE
extern "C" int lbmpc_prof_read_waves(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(lbmpc_prof_wave), sizeof(unsigned long long) * 64 * 4096) == hipSuccess ? 0 : -1;
}
extern "C" int lbmpc_prof_read(unsigned long long* out, int reset) {
    unsigned long long z[48] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(lbmpc_prof), sizeof(z)) != hipSuccess) return -1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(lbmpc_prof), z, sizeof(z)) != hipSuccess) return -1;
    return 0;
}
#endif

// ---- host: context ----------------------------------------------------------------------------------
static bool invert_gj(int n, const double* M /* row-major */, double* Minv) {
    std::vector<double> a((size_t)n * 2 * n);
    const int w = 2 * n;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) { a[i * w + j] = M[i * n + j]; a[i * w + n + j] = (i == j) ? 1.0 : 0.0; }
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

}  // namespace lbmpc

using namespace lbmpc;

#define CMX(M, rows, i, j) ((M)[(size_t)(i) + (size_t)(rows) * (size_t)(j)])

// Inscribed ellipsoid of S about the origin (used only to skip provably empty row scans; see CtxDev::ME).
// E = polar of the minimum-volume enclosing ellipsoid of the points +-a_j/b_j (Khachiyan's algorithm, stopped early:
// tightness is all that depends on convergence), then rescaled so that sqrt(a_j' ME^-1 a_j) <= b_j holds for the
// matrix actually stored.  hasE = 0 when 0 is not interior or the points do not span R^nx.
static void inscribed_ellipsoid(int nx, int nS, const double* S_A /* col-major */, const double* S_b, lbmpc::CtxDev* h) {
    h->hasE = 0; h->bmin = 0.0;
    for (int i = 0; i < LBMPC_MAX_NX * LBMPC_MAX_NX; ++i) h->ME[i] = 0.0;
    double bmin = 1e300;
    for (int j = 0; j < nS; ++j) bmin = std::min(bmin, S_b[j]);
    if (!(bmin > 0.0) || nS < nx) return;
    std::vector<double> P((size_t)nS * nx), u((size_t)nS, 1.0 / nS), g((size_t)nS);
    for (int j = 0; j < nS; ++j) for (int k = 0; k < nx; ++k) P[(size_t)j * nx + k] = CMX(S_A, nS, j, k) / S_b[j];
    double X[16], Xi[16];
    auto moment = [&]() {
        for (int i = 0; i < nx * nx; ++i) X[i] = 0.0;
        for (int j = 0; j < nS; ++j) for (int a = 0; a < nx; ++a) for (int b = 0; b < nx; ++b) X[a * nx + b] += u[j] * P[(size_t)j * nx + a] * P[(size_t)j * nx + b];
        return invert_gj(nx, X, Xi);
    };
    for (int it = 0; it < 3000; ++it) {
        if (!moment()) return;
        int jm = 0; double gm = -1.0;
        for (int j = 0; j < nS; ++j) {
            double acc = 0.0;
            for (int a = 0; a < nx; ++a) { double t = 0.0; for (int b = 0; b < nx; ++b) t += Xi[a * nx + b] * P[(size_t)j * nx + b]; acc += P[(size_t)j * nx + a] * t; }
            if (acc > gm) { gm = acc; jm = j; }
        }
        if (gm - nx < 1e-3 * nx) break;
        const double step = (gm - nx) / (nx * (gm - 1.0));
        for (int j = 0; j < nS; ++j) u[j] *= (1.0 - step);
        u[jm] += step;
    }
    if (!moment()) return;
    // enclosing ellipsoid of the points: y' (Xi / nx) y <= (about) 1;  E: x' (nx X) x <= 1, with ME^-1 = Xi / nx
    double ME[16], MEi[16];
    for (int i = 0; i < nx * nx; ++i) ME[i] = nx * X[i];
    for (int a = 0; a < nx; ++a) for (int b = a + 1; b < nx; ++b) { const double sy = 0.5 * (ME[a * nx + b] + ME[b * nx + a]); ME[a * nx + b] = ME[b * nx + a] = sy; }
    if (!invert_gj(nx, ME, MEi)) return;
    double r2 = 0.0;  // max_j a_j' ME^-1 a_j / b_j^2
    for (int j = 0; j < nS; ++j) {
        double acc = 0.0;
        for (int a = 0; a < nx; ++a) { double t = 0.0; for (int b = 0; b < nx; ++b) t += MEi[a * nx + b] * P[(size_t)j * nx + b]; acc += P[(size_t)j * nx + a] * t; }
        r2 = std::max(r2, acc);
    }
    if (!(r2 > 0.0) || !std::isfinite(r2)) return;
    for (int i = 0; i < nx * nx; ++i) h->ME[i] = ME[i] * r2 * (1.0 + 1e-9);  // shrink E until it is inside S
    h->bmin = bmin;
    h->hasE = 1;
}

extern "C" void lbmpc_destroy(lbmpc_ctx* c) {
    if (!c) return;
    DeviceGuard guard(c->device);
    if (c->d) (void)hipFree(c->d);
    if (c->dS) (void)hipFree(c->dS);
    if (c->dH) (void)hipFree(c->dH);
    if (c->dG) (void)hipFree(c->dG);
    if (c->dS32) (void)hipFree(c->dS32);
    if (c->xi) (void)hipFree(c->xi);
    if (c->xi_status) (void)hipFree(c->xi_status);
    if (c->counters) (void)hipFree(c->counters);
    if (c->park) (void)hipFree(c->park);
    if (c->stg_d) (void)hipFree(c->stg_d);
    if (c->stg_h) (void)hipHostFree(c->stg_h);
    if (c->m0) (void)hipFree(c->m0);
    if (c->i0_status) (void)hipFree(c->i0_status);
    if (c->smp_ph) (void)hipFree(c->smp_ph);
    if (c->smp_d) (void)hipFree(c->smp_d);
    if (c->hint) (void)hipFree(c->hint);
    if (c->st_d) (void)hipFree(c->st_d);
    if (c->st_i) (void)hipFree(c->st_i);
    delete c;
}

extern "C" int lbmpc_uqmpc_create(const lbmpc_params* p, lbmpc_ctx** out) {
    if (out) *out = nullptr;
    if (int rc = check_device()) return rc;
    if (!p || !out) { set_error("lbmpc_uqmpc_create: null argument"); return LBMPC_ERR_ARG; }
    const double* ptrs[] = {p->A, p->B, p->K, p->Phi, p->Px, p->Pc, p->F, p->G, p->F_bar, p->Psi, p->S_A, p->S_b, p->W_A, p->W_b, p->hs};
    for (const double* q : ptrs) if (!q) { set_error("lbmpc_uqmpc_create: null field in lbmpc_params"); return LBMPC_ERR_ARG; }
    const int nx = p->nx, nu = p->nu, N = p->N, nc = p->nc, nz = nx + N * nu, nS = p->nS;
    if (nx < 1 || nu < 1 || N < 1 || nc < 1 || p->Nu < 0 || nS < 1) { set_error("lbmpc_uqmpc_create: bad size"); return LBMPC_ERR_ARG; }
    const int nH = nc * (p->Nu + 1);
    if (nx > LBMPC_MAX_NX || nu > LBMPC_MAX_NU || nz > LBMPC_MAX_NZ || nc > LBMPC_MAX_NC || nS > 2048 || nH > 2048) {
        set_error("lbmpc_uqmpc_create: limits nx<=%d nu<=%d nx+N*nu<=%d nc<=%d nS<=2048 nc*(Nu+1)<=2048", LBMPC_MAX_NX, LBMPC_MAX_NU, LBMPC_MAX_NZ, LBMPC_MAX_NC);
        return LBMPC_ERR_DIM;
    }
    lbmpc_ctx* c = new lbmpc_ctx();
    CtxDev& h = c->h;
    std::memset(&h, 0, sizeof(h));
    if (int rc = pack_box(p->W_A, p->W_b, p->nW, nx, &h.box)) {
        // not an axis-aligned box: the set update solves LPforUQ by vertex enumeration (UQRobustMPC.m:124-139 takes any W)
        if (rc != LBMPC_ERR_UNSUPPORTED) { delete c; return rc; }
        if (p->nW < nx + 1 || p->nW > LBMPC_MAX_NW) { set_error("lbmpc_uqmpc_create: general W needs nx+1 <= nW <= %d", LBMPC_MAX_NW); delete c; return LBMPC_ERR_DIM; }
        std::memset(&h.box, 0, sizeof(h.box));
        h.genW = 1;
        h.gw.nW = p->nW; h.gw.nx = nx;
        for (int j = 0; j < p->nW; ++j) { for (int k = 0; k < nx; ++k) h.gw.WA[j * nx + k] = CMX(p->W_A, p->nW, j, k); h.gw.Wb[j] = p->W_b[j]; }
    }
    h.nx = nx; h.nu = nu; h.N = N; h.nc = nc; h.Nu = p->Nu; h.nS = nS; h.nW = p->nW; h.nz = nz; h.nH = nH;
    h.RL = nx + 1;              // [S_b, S_A row]; chunks of 4 rows are read as double2 broadcasts
    h.nSp = (nS + 127) & ~127;  // pad rows can never be violated
    h.nzp = ((nz + 3) & ~3) | 1;  // odd pitch: conflict-free per-lane row reads
    for (int i = 0; i < nx; ++i) for (int j = 0; j < nx; ++j) h.A[i * nx + j] = CMX(p->A, nx, i, j);
    for (int i = 0; i < nx; ++i) for (int j = 0; j < nu; ++j) h.B[i * nu + j] = CMX(p->B, nx, i, j);
    for (int j = 0; j < nu; ++j) for (int k = 0; k < nx; ++k) h.K[j * nx + k] = CMX(p->K, nu, j, k);
    for (int r = 0; r < nc; ++r) h.hs[r] = p->hs[r];
    // T = (I - Phi)^-1
    double ImP[16], T[16];
    for (int i = 0; i < nx; ++i) for (int j = 0; j < nx; ++j) ImP[i * nx + j] = ((i == j) ? 1.0 : 0.0) - CMX(p->Phi, nx, i, j);
    if (!invert_gj(nx, ImP, T)) { set_error("lbmpc_uqmpc_create: I - Phi is singular"); delete c; return LBMPC_ERR_ARG; }
    for (int i = 0; i < nx * nx; ++i) h.T[i] = T[i];
    // FcT = (F + G K) T
    for (int r = 0; r < nc; ++r) {
        double fc[LBMPC_MAX_NX];
        for (int k = 0; k < nx; ++k) {
            double acc = CMX(p->F, nc, r, k);
            for (int j = 0; j < nu; ++j) acc = std::fma(CMX(p->G, nc, r, j), CMX(p->K, nu, j, k), acc);
            fc[k] = acc;
        }
        for (int k = 0; k < nx; ++k) {
            double acc = 0.0;
            for (int j = 0; j < nx; ++j) acc = std::fma(fc[j], T[j * nx + k], acc);
            h.FcT[r * nx + k] = acc;
        }
    }
    // Ginv = (2 blkdiag(Px, Pc))^-1, symmetrised
    std::vector<double> P2((size_t)nz * nz, 0.0), Ginv((size_t)nz * nz);
    for (int i = 0; i < nx; ++i) for (int j = 0; j < nx; ++j) P2[i * nz + j] = 2.0 * CMX(p->Px, nx, i, j);
    const int ncv = N * nu;
    for (int i = 0; i < ncv; ++i) for (int j = 0; j < ncv; ++j) P2[(nx + i) * nz + nx + j] = 2.0 * CMX(p->Pc, ncv, i, j);
    if (!invert_gj(nz, P2.data(), Ginv.data())) { set_error("lbmpc_uqmpc_create: cost matrix is singular"); delete c; return LBMPC_ERR_ARG; }
    for (int i = 0; i < nz; ++i) for (int j = i + 1; j < nz; ++j) { double s = 0.5 * (Ginv[i * nz + j] + Ginv[j * nz + i]); Ginv[i * nz + j] = Ginv[j * nz + i] = s; }
    // the one-QP-per-lane solver with H rows recomputes Ginv c from the row: Ginv must be blkdiag(dense nx x nx, diagonal)
    // with exact zeros elsewhere (the inverse of a block-diagonal matrix by invert_gj keeps them exact), and nz = nx + 8
    {
        bool bd = nz == nx + 8 && (nx == 2 || nx == 3);
        for (int i = 0; i < nz && bd; ++i)
            for (int j = 0; j < nz; ++j) {
                const bool inblock = (i < nx && j < nx) || i == j;
                if (!inblock && Ginv[i * nz + j] != 0.0) { bd = false; break; }
            }
        const char* env_g = getenv("LBMPC_GEN_SOLVER");   // 0: skip the general lane solver (testing)
        h.gen_ok = bd && !(env_g && atoi(env_g) == 0);
        for (int i = 0; i < nx; ++i) for (int j = 0; j < nx; ++j) h.GiX[i * nx + j] = Ginv[i * nz + j];
        for (int i = 0; i < LBMPC_MAX_NZ; ++i) h.Gd[i] = (i >= nx && i < nz) ? Ginv[i * nz + i] : 0.0;
        h.nHs = nH + (h.gen_ok ? 1 : 0);   // the general lane solver reads the tail of an S row's coefficients from a row of zeros
    }
    // S rows [S_b | S_A]; pad rows (b = 1e300, a = 0) can never be violated
    std::vector<double> Srow((size_t)h.nSp * h.RL, 0.0);
    for (int j = nS; j < h.nSp; ++j) Srow[(size_t)j * h.RL] = 1e300;
    for (int j = 0; j < nS; ++j) { Srow[(size_t)j * h.RL] = p->S_b[j]; for (int k = 0; k < nx; ++k) Srow[(size_t)j * h.RL + 1 + k] = CMX(p->S_A, nS, j, k); }
    // fp32 filter copy, row pairs interleaved; pad rows: b = 3e38, a = 0
    std::vector<float> S32((size_t)h.nSp * h.RL, 0.0f);
    h.bmax = 0.0;
    for (int k = 0; k < LBMPC_MAX_NX; ++k) h.amax[k] = 0.0;
    for (int j = 0; j < h.nSp; ++j) {
        float* pr = S32.data() + (size_t)(j >> 1) * 2 * h.RL + (j & 1);
        pr[0] = j < nS ? (float)p->S_b[j] : 3.0e38f;
        for (int k = 0; k < nx; ++k) pr[2 + 2 * k] = j < nS ? (float)CMX(p->S_A, nS, j, k) : 0.0f;
        if (j < nS) {
            h.bmax = std::max(h.bmax, std::fabs(p->S_b[j]));
            for (int k = 0; k < nx; ++k) h.amax[k] = std::max(h.amax[k], std::fabs(CMX(p->S_A, nS, j, k)));
        }
    }
    inscribed_ellipsoid(nx, nS, p->S_A, p->S_b, &h);
    {
        const char* env_ls = getenv("LBMPC_LANE_SOLVER");   // 0: 16-lane-row solver only (testing)
        h.lane_solver = !(env_ls && atoi(env_ls) == 0);
    }
    // H rows: F_bar Psi^i, i = 0..Nu  (UQRobustMPC.m:153-155)
    std::vector<double> Hm((size_t)h.nHs * h.nzp, 0.0), M((size_t)nc * nz), M2((size_t)nc * nz);
    for (int r = 0; r < nc; ++r) for (int k = 0; k < nz; ++k) M[r * nz + k] = CMX(p->F_bar, nc, r, k);
    for (int i = 0; i <= p->Nu; ++i) {
        for (int r = 0; r < nc; ++r) for (int k = 0; k < nz; ++k) Hm[((size_t)i * nc + r) * h.nzp + k] = M[r * nz + k];
        for (int r = 0; r < nc; ++r) for (int k = 0; k < nz; ++k) {
            double acc = 0.0;
            for (int j = 0; j < nz; ++j) acc = std::fma(M[r * nz + j], CMX(p->Psi, nz, j, k), acc);
            M2[r * nz + k] = acc;
        }
        M.swap(M2);
    }
    for (int k = 0; k < LBMPC_MAX_NZ; ++k) h.cmH[k] = 0.0;
    for (int r = 0; r < nH; ++r) for (int k = 0; k < nz; ++k) h.cmH[k] = std::max(h.cmH[k], std::fabs(Hm[(size_t)r * h.nzp + k]));
    for (int m = 0; m < 32; ++m) for (int k = 0; k < LBMPC_MAX_NZ; ++k) h.cmHc[m][k] = 0.0;
    for (int r = 0; r < nH; ++r) for (int k = 0; k < nz; ++k) h.cmHc[r >> 6][k] = std::max(h.cmHc[r >> 6][k], std::fabs(Hm[(size_t)r * h.nzp + k]));
    if (hipGetDevice(&c->device) != hipSuccess) { set_error("lbmpc_uqmpc_create: hipGetDevice failed"); delete c; return LBMPC_ERR_HIP; }
    auto fail = [&](hipError_t e, const char* what) { set_error("lbmpc_uqmpc_create: %s -> %s", what, hipGetErrorString(e)); lbmpc_destroy(c); return LBMPC_ERR_HIP; };
    hipError_t e;
    if ((e = hipMalloc((void**)&c->dS, Srow.size() * 8)) != hipSuccess) return fail(e, "hipMalloc");
    if ((e = hipMalloc((void**)&c->dH, Hm.size() * 8)) != hipSuccess) return fail(e, "hipMalloc");
    if ((e = hipMalloc((void**)&c->dG, Ginv.size() * 8)) != hipSuccess) return fail(e, "hipMalloc");
    if ((e = hipMalloc((void**)&c->dS32, S32.size() * 4)) != hipSuccess) return fail(e, "hipMalloc");
    if ((e = hipMemcpy(c->dS32, S32.data(), S32.size() * 4, hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "hipMemcpy");
    if ((e = hipMalloc((void**)&c->d, sizeof(CtxDev))) != hipSuccess) return fail(e, "hipMalloc");
    if ((e = hipMemcpy(c->dS, Srow.data(), Srow.size() * 8, hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "hipMemcpy");
    if ((e = hipMemcpy(c->dH, Hm.data(), Hm.size() * 8, hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "hipMemcpy");
    if ((e = hipMemcpy(c->dG, Ginv.data(), Ginv.size() * 8, hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "hipMemcpy");
    h.Srow = c->dS; h.Hm = c->dH; h.Ginv = c->dG; h.S32 = c->dS32;
    if ((e = hipMemcpy(c->d, &h, sizeof(CtxDev), hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "hipMemcpy");
    c->lds_const_bytes = (size_t)lds_const_doubles(h) * 8;
    *out = c;
    return 0;
}

namespace lbmpc {
static const size_t LDS_LIMIT = 160 * 1024;

// waves per block so that constants + per-wave workspaces (+ the block's histograms) fit in 160 KiB of LDS
static int pick_waves(const lbmpc_ctx* c, int want) {
    const size_t wsb = (size_t)ws_doubles(c->h.nz) * 8;
    int w = want;
    while (w > 1 && c->lds_const_bytes + (size_t)w * wsb + (MC_HIST_INTS + MC_PROG_INTS) * 4 > LDS_LIMIT) w >>= 1;
    if (c->lds_const_bytes + (size_t)w * wsb + (MC_HIST_INTS + MC_PROG_INTS) * 4 > LDS_LIMIT) return 0;
    return w;
}
// lanes of the per-wave LDS region of the general lane solver: the largest power of two <= min(lanes_per_wave, 32)
// that fits beside the constants and the row-solver slots of `waves` waves; 0 when fewer than 4 fit or the solver
// does not apply to this context.  LBMPC_GEN_LANES caps it (testing).
static int pick_gen_lanes(const lbmpc_ctx* c, int waves, int lanes_per_wave) {
    if (!c->h.gen_ok) return 0;
    int cap = 32;
    if (const char* e = getenv("LBMPC_GEN_LANES")) { const int v = atoi(e); if (v >= 0 && v <= 64) cap = v; }
    const size_t base = c->lds_const_bytes + (size_t)waves * ws_doubles(c->h.nz) * 8 + (MC_HIST_INTS + MC_PROG_INTS) * 4;
    int gl = 4;
    while (gl < lanes_per_wave && gl < 64) gl <<= 1;   // the smallest power of two that holds the wave's trials ...
    while (gl > cap) gl >>= 1;                          // ... up to the cap (more trials are served in groups)
    while (gl >= 4 && base + (size_t)waves * lanegen_doubles(gl) * 8 > LDS_LIMIT) gl >>= 1;
    return gl >= 4 ? gl : 0;
}
static size_t block_lds_bytes(const lbmpc_ctx* c, int waves, int gen_lanes) {
    return c->lds_const_bytes + (size_t)waves * (ws_doubles(c->h.nz) + lanegen_doubles(gen_lanes)) * 8 + MC_HIST_INTS * 4;
}

template <class Kern>
static int allow_lds(Kern k, size_t bytes) {
    LB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}
}  // namespace lbmpc

static int step_dev(lbmpc_ctx* c, int64_t batch, int uq, const StepIO& io, hipStream_t st) {
    const int waves = pick_waves(c, 4);
    if (!waves) { set_error("constants (%zu B) do not fit in LDS", c->lds_const_bytes); return LBMPC_ERR_DIM; }
    const int gen_lanes = pick_gen_lanes(c, waves, 64);
    const size_t lds = block_lds_bytes(c, waves, gen_lanes);
    int64_t blocks = (batch + (int64_t)waves * 64 - 1) / ((int64_t)waves * 64);
    if (blocks > 1024) blocks = 1024;
    dim3 grid((unsigned)blocks), blk(waves * 64);
#define LAUNCH_STEP(NXV)                                                                             \
    do {                                                                                             \
        if (lds > c->step_lds_set) {   /* once per context: the attribute call costs microseconds on the per-step path */ \
            if (int rc = allow_lds(uqmpc_step_kernel<NXV>, lds)) return rc;                          \
            c->step_lds_set = lds;                                                                   \
        }                                                                                            \
        hipLaunchKernelGGL(uqmpc_step_kernel<NXV>, grid, blk, lds, st, c->d, batch, uq, gen_lanes, io); \
    } while (0)
#ifdef LBMPC_DEV_NX4    // development build (Makefile `dev4`): nx = 4 only
    switch (c->h.nx) {
        case 4: LAUNCH_STEP(4); break;
        default: set_error("development build: nx = 4 only"); return LBMPC_ERR_UNSUPPORTED;
    }
#elif defined(LBMPC_DEV_FAST)   // development build (Makefile `dev`): nx = 2, 3 only
    switch (c->h.nx) {
        case 2: LAUNCH_STEP(2); break;
        case 3: LAUNCH_STEP(3); break;
        default: set_error("development build: nx = 2, 3 only"); return LBMPC_ERR_UNSUPPORTED;
    }
#else
    switch (c->h.nx) {
        case 1: LAUNCH_STEP(1); break;
        case 2: LAUNCH_STEP(2); break;
        case 3: LAUNCH_STEP(3); break;
        default: LAUNCH_STEP(4); break;
    }
#endif
#undef LAUNCH_STEP
    LB_HIP(hipGetLastError());
    return 0;
}

// Host-pointer form of one UQ-RMPC step.  This is the call a script makes once per closed-loop step (Case_3...:69-72,
// batch = 1), so it allocates nothing after the first call of a given size: inputs and outputs go through one pinned
// host block and one device block owned by the context (grow-only), one host-to-device copy, one launch, one
// device-to-host copy of the outputs that were asked for.
extern "C" int lbmpc_uqmpc_step(lbmpc_ctx* c, int64_t batch, int uq_update, const double* x, const double* alpha_prev,
                                const double* v_prev, const double* w_new, double* s0, double* cc, double* u,
                                double* alpha, double* v, double* hk, double* Sb_hat, double* Wb_hat, int* status,
                                int* n_active, int* active_set, int* qp_iters) {
    if (int rc = check_any_device()) return rc;
    if (!c || !x || !alpha_prev || !v_prev || (uq_update && !w_new) || batch < 0) { set_error("lbmpc_uqmpc_step: bad argument"); return LBMPC_ERR_ARG; }
    if (batch == 0) return 0;
    DeviceGuard guard(c->device);
    if (guard.rc()) return guard.rc();
    const CtxDev& h = c->h;
    const size_t B = (size_t)batch;
    const int ncv = h.nz - h.nx;
    // block layout (bytes): inputs x | alpha | v | w, then the outputs in the order below; every piece 8-byte aligned
    const size_t in_bytes = B * (3 * (size_t)h.nx + 1) * 8;
    struct Out { void* host; size_t bytes, off; };
    Out outs[12] = {{s0, B * h.nx * 8, 0}, {cc, B * ncv * 8, 0}, {u, B * h.nu * 8, 0}, {alpha, B * 8, 0},
                    {v, B * h.nx * 8, 0}, {hk, B * h.nc * 8, 0}, {Sb_hat, B * h.nS * 8, 0}, {Wb_hat, B * h.nW * 8, 0},
                    {status, B * 4, 0}, {n_active, B * 4, 0}, {active_set, B * NZMAX * 4, 0}, {qp_iters, B * 4, 0}};
    size_t total = in_bytes;
    for (auto& o : outs) { o.off = total; if (o.host) total += (o.bytes + 7) & ~(size_t)7; }
    if (total > c->stg_bytes) {
        if (c->stg_d) LB_HIP(hipFree(c->stg_d));
        if (c->stg_h) LB_HIP(hipHostFree(c->stg_h));
        c->stg_d = nullptr; c->stg_h = nullptr; c->stg_bytes = 0;
        const size_t cap = total < 65536 ? 65536 : total;
        LB_HIP(hipMalloc(&c->stg_d, cap));
        LB_HIP(hipHostMalloc(&c->stg_h, cap, hipHostMallocDefault));
        c->stg_bytes = cap;
    }
    char* hb = static_cast<char*>(c->stg_h);
    char* db = static_cast<char*>(c->stg_d);
    double* hin = reinterpret_cast<double*>(hb);
    std::memcpy(hin, x, B * h.nx * 8);
    std::memcpy(hin + B * h.nx, alpha_prev, B * 8);
    std::memcpy(hin + B * (h.nx + 1), v_prev, B * h.nx * 8);
    if (w_new) std::memcpy(hin + B * (2 * h.nx + 1), w_new, B * h.nx * 8);
    LB_HIP(hipMemcpyAsync(db, hb, in_bytes, hipMemcpyHostToDevice, nullptr));
    double* din = reinterpret_cast<double*>(db);
    StepIO io;
    io.x = din; io.alpha_prev = din + B * h.nx; io.v_prev = din + B * (h.nx + 1); io.w_new = w_new ? din + B * (2 * h.nx + 1) : nullptr;
    auto dptr_of = [&](int i) -> void* { return outs[i].host ? db + outs[i].off : nullptr; };
    io.s0 = (double*)dptr_of(0); io.c = (double*)dptr_of(1); io.u = (double*)dptr_of(2); io.alpha = (double*)dptr_of(3);
    io.v = (double*)dptr_of(4); io.hk = (double*)dptr_of(5); io.Sb_hat = (double*)dptr_of(6); io.Wb_hat = (double*)dptr_of(7);
    io.status = (int*)dptr_of(8); io.n_active = (int*)dptr_of(9); io.active_set = (int*)dptr_of(10); io.qp_iters = (int*)dptr_of(11);
    if (int rc = step_dev(c, batch, uq_update, io, nullptr)) return rc;
    if (total > in_bytes) LB_HIP(hipMemcpyAsync(hb + in_bytes, db + in_bytes, total - in_bytes, hipMemcpyDeviceToHost, nullptr));
    LB_HIP(hipStreamSynchronize(nullptr));
    for (auto& o : outs) if (o.host) std::memcpy(o.host, hb + o.off, o.bytes);
    return 0;
}

#if defined(LBMPC_DEV_NX4) && !defined(LBMPC_DEV_NX4_GW)
#define LBMPC_DEV_NX4_GW true   // diagnostic: false = the box-only instantiation for nx = 4
#endif
namespace lbmpc {
// launch shape of the closed-loop kernels for a batch (shared by lbmpc_mc_run_dev and lbmpc_mc_plan)
struct McPlan {
    bool wide;        // two trials per lane
    int waves, tpw, gen_lanes, per_block;
    size_t lds;       // dynamic LDS per block of mc_kernel
    int ksplit;       // steps run by the transient kernel
    bool lean;        // the rest by mc_steady_kernel
};
static int mc_plan(const lbmpc_ctx* c, int64_t trials, int steps, McPlan* p) {
    const char* env_t = getenv("LBMPC_MC_T");
    // T = 2 trials per lane once 2-per-lane still fills every CU
    p->wide = env_t ? (atoi(env_t) == 2) : (trials >= (int64_t)256 * 8 * 128 * 2);
    if (c->h.genW) p->wide = false;   // the two-trials-per-lane instantiations carry the box set update only
    // trials per wave: 64 once that still gives every SIMD a wave of its own (1024 SIMDs, with some slack), else fewer
    // lanes per wave so that small batches spread over the chip (every solver tier costs per wave-trip, whatever the
    // number of live lanes).  Batches so small that 8-wave blocks would leave a quarter of the CUs idle run as 4-wave
    // blocks, one wave per SIMD, with the trials spread evenly (C3 as written, 10000 trials: 250 blocks of 4 x 10
    // trials instead of 157 of 8 x 8: -8 %; tools/c3_shape_sweep.sh).  Any packing gives bitwise the same results.
    int want_waves = 8, tpw = 64;
    while (tpw > 4 && (trials + tpw - 1) / tpw < 1088) tpw >>= 1;
    if ((trials + 8 * tpw - 1) / (8 * tpw) < 192) {
        want_waves = 4;
        tpw = (int)((trials + 1023) / 1024);
        tpw = tpw < 4 ? 4 : tpw;
    }
    if (const char* env_wv = getenv("LBMPC_MC_WAVES")) { const int v = atoi(env_wv); if (v >= 1 && v <= 8) want_waves = v; }   // testing
    if (const char* env_w = getenv("LBMPC_MC_TPW")) { const int v = atoi(env_w); if (v >= 1 && v <= 64) tpw = v; }
    p->waves = pick_waves(c, want_waves);
    if (!p->waves) { set_error("constants (%zu B) do not fit in LDS", c->lds_const_bytes); return LBMPC_ERR_DIM; }
    p->tpw = tpw;
    p->gen_lanes = pick_gen_lanes(c, p->waves, tpw);
    p->lds = block_lds_bytes(c, p->waves, p->gen_lanes);
    p->per_block = p->waves * tpw * (p->wide ? 2 : 1);
    // Long horizons run in step segments: [0, ksplit) (transient, QPs) by mc_kernel, the rest by mc_steady_kernel from
    // the saved state, then a resume launch of mc_kernel for the trials the steady kernel had to hand back (normally
    // none).  LBMPC_MC_SPLIT moves the split point and LBMPC_MC_LEAN=0 leaves the steady kernel out (testing).
    p->ksplit = steps >= 64 ? 32 : steps;
    if (const char* env_s = getenv("LBMPC_MC_SPLIT")) { const int v = atoi(env_s); p->ksplit = (v > 0 && v < steps) ? v : steps; }
    const char* env_l = getenv("LBMPC_MC_LEAN");
    // the steady-state kernel is written for an exact box W and one input (every reference configuration)
    p->lean = !(env_l && atoi(env_l) == 0) && c->h.box.exact && c->h.nu == 1;
    return 0;
}
template <class Kern>
static int kern_attrs(Kern k, int* vgprs, int* scratch, int* static_lds) {
    hipFuncAttributes fa;
    LB_HIP(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k)));
    *vgprs = fa.numRegs; *scratch = (int)fa.localSizeBytes; *static_lds = (int)fa.sharedSizeBytes;
    return 0;
}
}  // namespace lbmpc

// out[0] trials per wave, [1] trials per lane, [2] waves per block, [3] dynamic LDS bytes per block (= per CU: one
// block per CU), [4] lanes of the general lane solver's LDS region per wave (0: tier not available), [5] steps run by
// the transient kernel, [6] 1 if mc_steady_kernel runs the rest; of the mc_kernel instantiation this batch uses:
// [7] registers per lane (hipFuncAttributes::numRegs), [8] scratch bytes per lane, [9] resident waves per SIMD
// = min(512 / registers, waves per block / 4 x blocks per CU by LDS); [10] blocks of the transient launch.
extern "C" int lbmpc_mc_plan(lbmpc_ctx* c, int64_t trials, int steps, int32_t* out, int n) {
    if (int rc = check_any_device()) return rc;
    if (!c || !out || n < 0 || trials < 1 || steps < 0) { set_error("lbmpc_mc_plan: bad argument"); return LBMPC_ERR_ARG; }
    DeviceGuard guard(c->device);
    if (guard.rc()) return guard.rc();
    McPlan p;
    if (int rc = mc_plan(c, trials, steps, &p)) return rc;
    int vg = 0, sc = 0, sl = 0, rc = 0;
#ifdef LBMPC_DEV_NX4
    rc = kern_attrs(mc_kernel<4, 1, LBMPC_DEV_NX4_GW>, &vg, &sc, &sl);
#elif defined(LBMPC_DEV_FAST)
    switch (c->h.nx * 2 + (c->h.genW ? 1 : 0)) {
        case 4: rc = kern_attrs(mc_kernel<2, 1, false>, &vg, &sc, &sl); break;
        case 5: rc = kern_attrs(mc_kernel<2, 1, true>, &vg, &sc, &sl); break;
        case 6: rc = kern_attrs(mc_kernel<3, 1, false>, &vg, &sc, &sl); break;
        default: rc = kern_attrs(mc_kernel<3, 1, true>, &vg, &sc, &sl); break;
    }
#else
    switch (c->h.nx * 2 + (p.wide ? 1 : 0)) {
        case 2: rc = c->h.genW ? kern_attrs(mc_kernel<1, 1, true>, &vg, &sc, &sl) : kern_attrs(mc_kernel<1, 1, false>, &vg, &sc, &sl); break;
        case 3: rc = kern_attrs(mc_kernel<1, 2, false>, &vg, &sc, &sl); break;
        case 4: rc = c->h.genW ? kern_attrs(mc_kernel<2, 1, true>, &vg, &sc, &sl) : kern_attrs(mc_kernel<2, 1, false>, &vg, &sc, &sl); break;
        case 5: rc = kern_attrs(mc_kernel<2, 2, false>, &vg, &sc, &sl); break;
        case 6: rc = c->h.genW ? kern_attrs(mc_kernel<3, 1, true>, &vg, &sc, &sl) : kern_attrs(mc_kernel<3, 1, false>, &vg, &sc, &sl); break;
        case 7: rc = kern_attrs(mc_kernel<3, 2, false>, &vg, &sc, &sl); break;
        case 8: rc = kern_attrs(mc_kernel<4, 1, true>, &vg, &sc, &sl); break;
        default: rc = kern_attrs(mc_kernel<4, 2, false>, &vg, &sc, &sl); break;
    }
#endif
    if (rc) return rc;
    const int by_regs = vg > 0 ? 512 / vg : 8;
    const int blocks_per_cu = (int)(LDS_LIMIT / p.lds) < 1 ? 1 : (int)(LDS_LIMIT / p.lds);
    int by_lds = p.waves * blocks_per_cu / 4;
    if (by_lds < 1) by_lds = 1;
    int64_t blocks = (trials + p.per_block - 1) / p.per_block;
    const int vals[11] = {p.tpw, p.wide ? 2 : 1, p.waves, (int)p.lds, p.gen_lanes, p.ksplit, p.lean ? 1 : 0, vg, sc,
                          by_regs < by_lds ? by_regs : by_lds, (int)(blocks > 0x7fffffff ? 0x7fffffff : blocks)};
    for (int i = 0; i < n && i < 11; ++i) out[i] = vals[i];
    return 0;
}

extern "C" int lbmpc_mc_run_dev(lbmpc_ctx* c, const lbmpc_sampler* smp, const lbmpc_mc_cfg* cfg, lbmpc_mc_out* out,
                                void* stream) {
    if (int rc = check_any_device()) return rc;
    if (!c || !smp || !cfg || !out || !cfg->x0) { set_error("lbmpc_mc_run: null argument"); return LBMPC_ERR_ARG; }
    DeviceGuard guard(c->device);
    if (guard.rc()) return guard.rc();
    if (cfg->trials < 0 || cfg->steps < 0 || cfg->card_I0 < 1 || cfg->sampler_max_attempts < 1 || cfg->traj_trials < 0) { set_error("lbmpc_mc_run: bad configuration"); return LBMPC_ERR_ARG; }
    SamplerDev sd;
    if (int rc = pack_sampler(smp, &sd)) return rc;
    if (sd.nx != c->h.nx || sd.nu != c->h.nu) { set_error("lbmpc_mc_run: sampler dims differ from the controller's"); return LBMPC_ERR_ARG; }
    if (cfg->trials == 0) return 0;
    if (!c->smp_d) LB_HIP(hipMalloc((void**)&c->smp_d, sizeof(SamplerDev)));
    LB_HIP(hipMemcpyAsync(c->smp_d, &sd, sizeof(SamplerDev), hipMemcpyHostToDevice, (hipStream_t)stream));   // pageable source: staged before the call returns
    McArgs a;
    a.trials = cfg->trials; a.trial0 = cfg->trial0; a.steps = cfg->steps; a.card_I0 = cfg->card_I0; a.seed = cfg->seed;
    for (int k = 0; k < LBMPC_MAX_NX; ++k) a.x0[k] = k < c->h.nx ? cfg->x0[k] : 0.0;
    a.acc_bound = cfg->acc_bound; a.distance_error = cfg->distance_error;
    a.traj_trials = cfg->traj_trials; a.sampler_max_attempts = cfg->sampler_max_attempts;
    a.out = *out;
    hipStream_t st = (hipStream_t)stream;
    McPlan plan;
    if (int rc = mc_plan(c, cfg->trials, cfg->steps, &plan)) return rc;
    const bool wide = plan.wide;
    const int waves = plan.waves, tpw = plan.tpw, per_block = plan.per_block;
    a.tpw = tpw;
    a.gen_lanes = plan.gen_lanes;
    const size_t lds = plan.lds;
    // trial chunks: the pre-drawn disturbances of one chunk take steps * 2 nx * 8 B per trial
    const size_t per_trial = xi_bytes_for(c->h.nx, cfg->steps, 64) / 64;
    const char* env_g = getenv("LBMPC_MC_XI_GIB");
    const double cap_gib = env_g ? atof(env_g) : 16.0;
    int64_t chunk = cfg->trials;
    if (per_trial > 0) {
        const int64_t fit = (int64_t)(cap_gib * 1073741824.0 / (double)per_trial);
        if (fit < chunk) chunk = fit < per_block ? per_block : fit / per_block * per_block;
    }
    if (chunk > cfg->trials) chunk = cfg->trials;
    {
        const size_t need = xi_bytes_for(c->h.nx, cfg->steps, chunk);
        if (need > c->xi_bytes) {
            if (c->xi) LB_HIP(hipFree(c->xi));
            c->xi = nullptr; c->xi_bytes = 0;
            hipError_t e = hipMalloc((void**)&c->xi, need);
            if (e != hipSuccess) { set_error("lbmpc_mc_run: %zu B of disturbance scratch -> %s (lower LBMPC_MC_XI_GIB)", need, hipGetErrorString(e)); return LBMPC_ERR_HIP; }
            c->xi_bytes = need;
        }
        if ((size_t)chunk > c->xi_status_n) {
            if (c->xi_status) LB_HIP(hipFree(c->xi_status));
            c->xi_status = nullptr; c->xi_status_n = 0;
            LB_HIP(hipMalloc((void**)&c->xi_status, (size_t)chunk * sizeof(int)));
            c->xi_status_n = (size_t)chunk;
        }
    }
    if (!c->counters) LB_HIP(hipMalloc((void**)&c->counters, MC_STATS_N * sizeof(unsigned long long)));
    LB_HIP(hipMemsetAsync(c->counters, 0, MC_STATS_N * sizeof(unsigned long long), st));
    if ((size_t)chunk > c->hint_n) {
        if (c->hint) LB_HIP(hipFree(c->hint));
        c->hint = nullptr; c->hint_n = 0;
        LB_HIP(hipMalloc((void**)&c->hint, (size_t)chunk * 17 * sizeof(int)));
        c->hint_n = (size_t)chunk;
    }
    a.hint = c->hint; a.nhint = c->hint + (size_t)c->hint_n * 16;
    const int ksplit = plan.ksplit;
    const bool lean = plan.lean;
    const size_t lds_steady = steady_lds_bytes(c->h.nSp, c->h.RL, c->h.nx);
    if (ksplit < cfg->steps && (size_t)chunk > c->st_n) {
        if (c->st_d) LB_HIP(hipFree(c->st_d));
        if (c->st_i) LB_HIP(hipFree(c->st_i));
        c->st_d = nullptr; c->st_i = nullptr; c->st_n = 0;
        LB_HIP(hipMalloc((void**)&c->st_d, (size_t)chunk * (3 * c->h.nx + 1) * sizeof(double)));
        LB_HIP(hipMalloc((void**)&c->st_i, (size_t)chunk * 5 * sizeof(int)));
        c->st_n = (size_t)chunk;
    }
    if ((size_t)chunk > c->m0_n) {
        if (c->m0) LB_HIP(hipFree(c->m0));
        if (c->i0_status) LB_HIP(hipFree(c->i0_status));
        c->m0 = nullptr; c->i0_status = nullptr; c->m0_n = 0;
        LB_HIP(hipMalloc((void**)&c->m0, (size_t)chunk * LBMPC_MAX_NW * sizeof(double)));   // 2 nx rows (box) or nW <= 16 (general W)
        LB_HIP(hipMalloc((void**)&c->i0_status, (size_t)chunk * sizeof(int)));
        c->m0_n = (size_t)chunk;
    }
    if (!c->smp_ph) LB_HIP(hipMalloc((void**)&c->smp_ph, sizeof(SmpPhases)));
    {
        SmpPhases hp;
        if (c->h.genW) pack_sample_phases(sd, c->h.gw.WA, c->h.gw.nW, &hp);   // rows in the caller's order
        else pack_sample_phases(sd, c->h.box.WA, 2 * c->h.nx, &hp);           // rows in the canonical order of the box
        LB_HIP(hipMemcpyAsync(c->smp_ph, &hp, sizeof(SmpPhases), hipMemcpyHostToDevice, st));   // pageable source: staged before the call returns
    }
    a.m0 = c->m0; a.i0_status = c->i0_status;
    a.st_d = c->st_d; a.st_i = c->st_i;
    a.xi = c->xi; a.xi_status = c->xi_status; a.counters = c->counters;
    for (int64_t c0 = 0; c0 < cfg->trials; c0 += chunk) {
        const int64_t cn = cfg->trials - c0 < chunk ? cfg->trials - c0 : chunk;
        a.chunk0 = c0; a.chunk_n = cn;
        LB_HIP(hipMemsetAsync(a.nhint, 0, (size_t)cn * sizeof(int), st));
        // information-set pre-pass: set (seed, trial) of card_I0 samples per trial, attempt cap 100 (InitialSetComputation.m:108)
        if (int rc = launch_sample_sets(c->h.nx, c->h.nu, c->smp_ph, c->h.genW ? c->h.gw.nW : 2 * c->h.nx, cfg->card_I0, cn, cfg->seed, (uint64_t)(cfg->trial0 + c0), 100,
                                        nullptr, c->m0, c->i0_status, st)) return rc;
        if (cfg->steps > 0) {
            LB_HIP(hipMemsetAsync(c->xi_status, 0, (size_t)cn * sizeof(int), st));
            // enough segments to put about 8 waves on every SIMD
            const int64_t sblocks = (cn + 255) / 256;
            int segs = (int)((8192 + sblocks * 4 - 1) / (sblocks * 4));
            segs = segs < 1 ? 1 : (segs > 64 ? 64 : segs);
            if (segs > cfg->steps) segs = cfg->steps;
            dim3 sg((unsigned)sblocks, (unsigned)segs), sb(256);
            if (int rc = launch_mc_sample(c->h.nx, sd.LV_hasU, sd.EV_hasU, sg, sb, st, c->smp_d, cn, cfg->trial0 + c0, cfg->steps, cfg->seed, cfg->sampler_max_attempts, c->xi, c->xi_status)) return rc;
        }
        int64_t blocks = (cn + per_block - 1) / per_block;
        const int64_t resident = 256 * (int64_t)(LDS_LIMIT / lds > 4 ? 4 : (LDS_LIMIT / lds < 1 ? 1 : LDS_LIMIT / lds));
        if (blocks > resident * 4) blocks = resident * 4;
        dim3 grid((unsigned)blocks), blk(waves * 64);
        {
            const size_t need = (size_t)blocks * waves * 64 * 2 * (3 * LBMPC_MAX_NX + 4) * 8;  // T <= 2
            if (need > c->park_bytes) {
                if (c->park) LB_HIP(hipFree(c->park));
                c->park = nullptr; c->park_bytes = 0;
                LB_HIP(hipMalloc((void**)&c->park, need));
                c->park_bytes = need;
            }
            a.park = c->park;
        }
#define LAUNCH_MC_GW(NXV, TV, GWV)                                                              \
    do {                                                                                        \
        if (int rc = allow_lds(mc_kernel<NXV, TV, GWV>, lds)) return rc;                        \
        hipLaunchKernelGGL((mc_kernel<NXV, TV, GWV>), grid, blk, lds, st, c->d, a);             \
    } while (0)
// one trial per lane: GW instantiation for a general W, and for nx = 4 always (see mc_kernel)
#define LAUNCH_MC1(NXV, BOXGW)                                                                  \
    do { if (c->h.genW || BOXGW) LAUNCH_MC_GW(NXV, 1, true); else LAUNCH_MC_GW(NXV, 1, false); } while (0)
#ifdef LBMPC_DEV_NX4
#ifndef LBMPC_DEV_NX4_GW
#define LBMPC_DEV_NX4_GW true
#endif
#define LAUNCH_MC_ALL()                                                                         \
    switch (c->h.nx * 2 + (wide ? 1 : 0)) {                                                     \
        case 8: LAUNCH_MC_GW(4, 1, LBMPC_DEV_NX4_GW); break;                                    \
        default: set_error("development build: nx = 4, one trial per lane"); return LBMPC_ERR_UNSUPPORTED; \
    }
#elif defined(LBMPC_DEV_FAST)
#define LAUNCH_MC_ALL()                                                                         \
    switch (c->h.nx * 2 + (wide ? 1 : 0)) {                                                     \
        case 4: LAUNCH_MC1(2, false); break;                                                    \
        case 6: LAUNCH_MC1(3, false); break;                                                    \
        default: set_error("development build: nx = 2, 3, one trial per lane"); return LBMPC_ERR_UNSUPPORTED; \
    }
#else
#define LAUNCH_MC_ALL()                                                                         \
    switch (c->h.nx * 2 + (wide ? 1 : 0)) {                                                     \
        case 2: LAUNCH_MC1(1, false); break;                                                    \
        case 3: LAUNCH_MC_GW(1, 2, false); break;                                               \
        case 4: LAUNCH_MC1(2, false); break;                                                    \
        case 5: LAUNCH_MC_GW(2, 2, false); break;                                               \
        case 6: LAUNCH_MC1(3, false); break;                                                    \
        case 7: LAUNCH_MC_GW(3, 2, false); break;                                               \
        case 8: LAUNCH_MC_GW(4, 1, true); break;                                                \
        default: LAUNCH_MC_GW(4, 2, false); break;                                              \
    }
#endif
        a.kseg0 = 0; a.kseg1 = ksplit; a.resume = 0;
        LAUNCH_MC_ALL();
        if (ksplit < cfg->steps) {
            a.kseg0 = ksplit; a.kseg1 = cfg->steps; a.resume = 1;
            if (lean) {
                dim3 g2((unsigned)((cn + 255) / 256)), b2(256);
                if (int rc = launch_mc_steady(c, g2, b2, lds_steady, st, a)) return rc;
            }
            LAUNCH_MC_ALL();   // trials whose saved step is before the horizon's end (all of them without the lean kernel)
        }
#undef LAUNCH_MC_ALL
#undef LAUNCH_MC1
#undef LAUNCH_MC_GW
        LB_HIP(hipGetLastError());
    }
    return 0;
}

extern "C" int lbmpc_mc_counters(lbmpc_ctx* c, uint64_t* out12) {
    if (int rc = check_any_device()) return rc;
    if (!c || !out12) { set_error("lbmpc_mc_counters: null argument"); return LBMPC_ERR_ARG; }
    for (int i = 0; i < 12; ++i) out12[i] = 0;
    if (!c->counters) return 0;
    DeviceGuard guard(c->device);
    if (guard.rc()) return guard.rc();
    LB_HIP(hipDeviceSynchronize());
    LB_HIP(hipMemcpy(out12, c->counters, 12 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int lbmpc_mc_stats(lbmpc_ctx* c, uint64_t* out, int n) {
    if (int rc = check_any_device()) return rc;
    if (!c || !out || n < 0) { set_error("lbmpc_mc_stats: bad argument"); return LBMPC_ERR_ARG; }
    for (int i = 0; i < n; ++i) out[i] = 0;
    if (!c->counters) return 0;
    DeviceGuard guard(c->device);
    if (guard.rc()) return guard.rc();
    LB_HIP(hipDeviceSynchronize());
    LB_HIP(hipMemcpy(out, c->counters, (size_t)(n < MC_STATS_N ? n : MC_STATS_N) * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int lbmpc_mc_run(lbmpc_ctx* c, const lbmpc_sampler* smp, const lbmpc_mc_cfg* cfg, lbmpc_mc_out* out) {
    if (int rc = check_any_device()) return rc;
    if (!c || !cfg || !out) { set_error("lbmpc_mc_run: null argument"); return LBMPC_ERR_ARG; }
    if (cfg->trials < 0 || cfg->steps < 0 || cfg->traj_trials < 0) { set_error("lbmpc_mc_run: bad configuration"); return LBMPC_ERR_ARG; }
    DeviceGuard guard(c->device);
    if (guard.rc()) return guard.rc();
    const size_t Tn = (size_t)cfg->trials, nx = c->h.nx, nu = c->h.nu, K = (size_t)cfg->steps;
    const size_t R = (size_t)(cfg->traj_trials < cfg->trials ? cfg->traj_trials : cfg->trials);
    struct Out { void* host; size_t bytes; DevBuf buf; };
    Out o[11] = {{out->infeasible, Tn * 4, {}}, {out->status, Tn * 4, {}}, {out->qp_iters, Tn * 4, {}},
                 {out->alpha_final, Tn * 8, {}}, {out->v_final, Tn * nx * 8, {}}, {out->alpha0, Tn * 8, {}},
                 {out->x_final, Tn * nx * 8, {}}, {out->traj_x, R * (K + 1) * nx * 8, {}}, {out->traj_u, R * K * nu * 8, {}},
                 {out->traj_alpha, R * (K + 1) * 8, {}}, {out->traj_s0, R * K * nx * 8, {}}};
    for (auto& x : o) if (x.host) LB_HIP(x.buf.alloc(x.bytes));
    lbmpc_mc_out d;
    d.infeasible = o[0].buf.as<int32_t>(); d.status = o[1].buf.as<int32_t>(); d.qp_iters = o[2].buf.as<int32_t>();
    d.alpha_final = o[3].buf.as<double>(); d.v_final = o[4].buf.as<double>(); d.alpha0 = o[5].buf.as<double>();
    d.x_final = o[6].buf.as<double>(); d.traj_x = o[7].buf.as<double>(); d.traj_u = o[8].buf.as<double>();
    d.traj_alpha = o[9].buf.as<double>(); d.traj_s0 = o[10].buf.as<double>();
    lbmpc_mc_cfg cf = *cfg;
    cf.traj_trials = (int32_t)R;
    if (int rc = lbmpc_mc_run_dev(c, smp, &cf, &d, nullptr)) return rc;
    LB_HIP(hipDeviceSynchronize());
    for (auto& x : o) if (x.host) LB_HIP(hipMemcpy(x.host, x.buf.p, x.bytes, hipMemcpyDeviceToHost));
    return 0;
}
