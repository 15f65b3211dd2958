// mc_steady.hip -- the closed loop outside the transient: the step-disturbance pre-pass (mc_sample_kernel) and the
// steady-state kernel (mc_steady_kernel: set update, kept tightening, first-pass filter, plant; no solver).
// See uqmpc.hip for the execution model and DESIGN.md "Kernel 3".
#include "uqmpc_dev.hpp"

namespace lbmpc {

// ---- step-disturbance pre-pass ------------------------------------------------------------------------
// w = (xi_EV - xi_LV) - B u_LV of every (trial, step): xi_* by rejection sampling (ModelingCar.m:36-43 via :47-91), in
// the form the information set uses (InitialSetComputation.m:51-52).  The draws depend on (seed, trial, step, attempt)
// only, not on the closed-loop state, so they are produced ahead of the closed loop, one trial per lane, step by step:
// the wave's stores of a step are one contiguous 64*nx*8-byte run of the step-major layout.  A wave repeats a draw
// until its last lane has accepted: free with the tight proposal boxes (lbmpc_sampler.EV_U / LV_U: acceptance ~1), a
// few extra attempts per draw with loose axis-aligned ones.  grid.y splits the steps into equal segments (more waves
// in flight).  3 Philox blocks per trial-step.
//
// Measured on C4: neither the Philox rounds (2 instead of 10: -3 %) nor the stores (-6 %) bound this kernel; the
// control flow around them did (row loops, divergent attempt bodies, the two proposal forms: scalar-unit and branch
// instructions, shared by the four SIMDs of a CU).  Hence: the descriptor is read from LDS as broadcasts (by value it
// lived in SGPRs that spilled to vector lanes), rows are tested four at a time (the descriptor pads them with rows
// 0.x <= inf), every lane evaluates every attempt and a select keeps the accepted point (no divergent body), the
// proposal form is a template parameter and the ten round keys of the seed are computed once.
struct PhiloxKeys { uint32_t k0[10], k1[10]; };
__device__ __forceinline__ void philox_keys(uint64_t seed, PhiloxKeys& K) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) { K.k0[r] = k0; K.k1[r] = k1; k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
}
// philox4x32_10 (common.hpp) with the key schedule precomputed: same words
__device__ __forceinline__ void philox4x32_10_keys(uint32_t c[4], const PhiloxKeys& K) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c[1], K.k0[r], 0x96);
        const uint32_t n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c[3], K.k1[r], 0x96);
        c[0] = n0; c[1] = (uint32_t)p1; c[2] = n2; c[3] = (uint32_t)p0;
    }
}
// sample_polytope (common.hpp) in the form described above: same draws, same arithmetic, same accepted point
template <int NX, bool HASU>
__device__ __forceinline__ bool sample_polytope_rows4(const double* XA, const double* Xb, int nr /* by value: diag may zero it */, const double* lo,
                                                      const double* hi, const double* U, const PhiloxKeys& K,
                                                      uint64_t trial, uint32_t index, uint32_t purpose, int max_attempts,
                                                      double* x) {
    bool done = false;
    for (int a = 0; a < max_attempts; ++a) {
        uint32_t c[4] = {(uint32_t)trial, (uint32_t)(trial >> 32), index, (purpose << 28) | ((uint32_t)a << 4)};
        philox4x32_10_keys(c, K);
#ifdef SAMPLE_DIAG_DOUBLE_PHILOX
        philox4x32_10_keys(c, K);
#endif
        double t[NX], xt[NX];
#pragma unroll
        for (int k = 0; k < NX; ++k) t[k] = lo[k] + (hi[k] - lo[k]) * u01(c[k]);
        if (HASU) {
#pragma unroll
            for (int k = 0; k < NX; ++k) {
                double acc = 0.0;
#pragma unroll
                for (int i = 0; i < NX; ++i) acc = fma(U[k * NX + i], t[i], acc);
                xt[k] = acc;
            }
        } else {
#pragma unroll
            for (int k = 0; k < NX; ++k) xt[k] = t[k];
        }
        bool ok = true;
#ifdef SAMPLE_DIAG_NOROWS
        nr = 0;
#endif
        for (int r = 0; r < nr; r += 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < NX; ++k) acc = fma(XA[(r + j) * NX + k], xt[k], acc);
                ok = ok && (acc <= Xb[r + j]);
            }
        }
#pragma unroll
        for (int k = 0; k < NX; ++k) x[k] = done ? x[k] : xt[k];
        done = done || ok;
        if (__builtin_amdgcn_ballot_w64(!done) == 0) break;  // every lane accepted
    }
    return !done;
}

#ifndef SAMPLE_MINWAVES
#define SAMPLE_MINWAVES 1
#endif
template <int NX, bool LVU, bool EVU>
__global__ __launch_bounds__(256, SAMPLE_MINWAVES) void mc_sample_kernel(const SamplerDev* __restrict__ smp_g, int64_t chunk_n,
                                                        int64_t trial_first, int K, uint64_t seed, int max_attempts,
                                                        double* __restrict__ xi, int* __restrict__ xi_status) {
    __shared__ SamplerDev smp;
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(smp_g);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&smp);
        for (int i = threadIdx.x; i < (int)(sizeof(SamplerDev) / 4); i += blockDim.x) dst[i] = src[i];
        __syncthreads();
    }
    const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = l < chunk_n;
    const int64_t cl = valid ? l : chunk_n - 1;   // idle lanes of the last wave repeat the last trial (no stores)
    const uint64_t tid = (uint64_t)(trial_first + cl);
    const int seg = (K + (int)gridDim.y - 1) / (int)gridDim.y;
    const int kbeg = (int)blockIdx.y * seg < K ? (int)blockIdx.y * seg : K;
    const int kend = kbeg + seg < K ? kbeg + seg : K;
    PhiloxKeys keys;
    philox_keys(seed, keys);
    const int nu = smp.nu, nLV = smp.nLV, nEV = smp.nEV;
    bool cap = false;
    for (int k = kbeg; k < kend; ++k) {
        double lv[NX], ev[NX], Bul[NX];
        // read where used: without the laundered offset the compiler keeps the whole descriptor in ~150 vector registers
        int off = 0;
        asm volatile("" : "+v"(off));
        const SamplerDev* S = reinterpret_cast<const SamplerDev*>(reinterpret_cast<const char*>(&smp) + off);
        cap |= sample_polytope_rows4<NX, LVU>(S->LV_A, S->LV_b, nLV, S->LV_lo, S->LV_hi, S->LV_U, keys, tid, (uint32_t)k, P_ST_LV, max_attempts, lv);
        {   // sample_Bu_lv (common.hpp)
            uint32_t c[4] = {(uint32_t)tid, (uint32_t)(tid >> 32), (uint32_t)k, ((uint32_t)P_ST_U << 28)};
            philox4x32_10_keys(c, keys);
#pragma unroll
            for (int i = 0; i < NX; ++i) Bul[i] = 0.0;
#pragma unroll
            for (int j = 0; j < LBMPC_MAX_NU; ++j)
                if (j < nu) {
                    const double uj = (S->max_u[j] - S->min_u[j]) * u01(c[j]) + S->min_u[j];
#pragma unroll
                    for (int i = 0; i < NX; ++i) Bul[i] = fma(S->B[i * nu + j], uj, Bul[i]);
                }
        }
        asm volatile("" : "+v"(off));
        S = reinterpret_cast<const SamplerDev*>(reinterpret_cast<const char*>(&smp) + off);
        cap |= sample_polytope_rows4<NX, EVU>(S->EV_A, S->EV_b, nEV, S->EV_lo, S->EV_hi, S->EV_U, keys, tid, (uint32_t)k, P_ST_EV, max_attempts, ev);
#ifdef SAMPLE_DIAG_NOSTORE
        if (valid && (ev[0] == 123.456 || k == kend - 1)) {
#else
        if (valid) {
#endif
            double* dst = xi + xi_index<NX>(cl, k, chunk_n);
#pragma unroll
            for (int i = 0; i < NX; ++i) dst[i] = (ev[i] - lv[i]) - Bul[i];
        }
    }
    if (valid && cap) atomicMax(&xi_status[cl], (int)LBMPC_ST_SAMPLER_CAP);
}

// ---- steady-state closed loop -----------------------------------------------------------------------
// After the transient no trial needs a QP (C4: none after step 8 of 1000): a step is the set update, the kept
// tightening, the first-pass filter (ellipsoid test; packed-fp32 row scan when a lane is outside) and the plant update
// with z = 0.  This kernel runs exactly those steps, lane per trial, for contexts with an exact box W and nu = 1 (every
// reference configuration; others run these steps in mc_kernel's resume mode).  A trial whose filter asks for a QP stops
// with its state saved BEFORE that step; the resume launch of mc_kernel takes it from there (same arithmetic, so the
// split is invisible in the results).
//
// What bounds it is latency, not arithmetic (two waves per SIMD at C4's 131072 trials), so the step is built to expose
// none: the per-step constants live in vector registers (StepConst), everything derived from (alpha, v) -- b_hat,
// (1-alpha) T v, min(1-h_k), the ellipsoid radius -- is kept and redone only in the lanes whose set moved, the step
// disturbances arrive through a ring of STEADY_RING register slots loaded that many steps ahead, and the rare row
// scan gives the constants' registers up (they are read again after it).
#ifndef STEADY_RING
#define STEADY_RING 2
#endif
template <int NX>
__global__ __launch_bounds__(256, 2) void mc_steady_kernel(const CtxDev* __restrict__ C, McArgs a) {
    extern __shared__ double lds[];
    float* s32 = reinterpret_cast<float*>(lds);
    // LDS: fp32 S rows | SteadyLds (constants of the branch taken when (alpha, v) moves: read as broadcasts)
    double* tcon0 = lds + ((C->nSp * C->RL + 1) >> 1);
    typedef SteadyLds<NX> SL;
    const int nc = C->nc;
    {
        const int n = C->nSp * C->RL;
        for (int i = threadIdx.x; i < n; i += blockDim.x) s32[i] = C->S32[i];
        for (int i = threadIdx.x; i < NX * NX; i += blockDim.x) tcon0[SL::T + i] = C->T[i];
        for (int i = threadIdx.x; i < nc * NX; i += blockDim.x) tcon0[SL::FcT + i] = C->FcT[i];
        for (int i = threadIdx.x; i < nc; i += blockDim.x) tcon0[SL::hs + i] = C->hs[i];
        for (int i = threadIdx.x; i < 2 * NX; i += blockDim.x) { tcon0[SL::wb + i] = C->box.Wb[i]; tcon0[SL::ic + i] = C->box.ic[i]; }
        __syncthreads();
    }
    StepConst<NX> sc;
    load_step_const<NX>(C, sc);
    LdsView L;
    L.Srow = nullptr; L.Hm = nullptr; L.Ginv = nullptr; L.S32 = s32;
    const int K = a.steps;
    const int64_t cn = a.chunk_n;
    const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = l < cn;
    const int64_t c = valid ? l : cn - 1;
    const int64_t tl = a.chunk0 + c;
    unsigned nfilter = 0, nsteps = 0;
    double x[NX], v[NX], wst[NX], alpha;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        x[i] = a.st_d[(0 * NX + i) * cn + c]; v[i] = a.st_d[(1 * NX + i) * cn + c]; wst[i] = a.st_d[(2 * NX + i) * cn + c];
    }
    alpha = a.st_d[3 * NX * cn + c];
    int status = a.st_i[0 * cn + c], infeasible = a.st_i[1 * cn + c];
    bool hashint = a.st_i[3 * cn + c] != 0;
    bool active = valid && a.st_i[4 * cn + c] == a.kseg0;
    const double accb = a.acc_bound, derr = a.distance_error;
    // kept between steps, functions of (alpha, v) alone
    double bh[2 * NX], bTv[NX], as2;
    bool ell_ok, need_h;   // need_h: some 1 - h_k[r] < -tol (the H rows alone ask for the solver)
    // bound for the tightening rows: 1 - h_k[r] >= 1 - (alpha max_r hs[r] + (1-alpha) max_r |FcT[r,:]|_1 max_k |v_k|)
    double hs_max = -LB_INF, g1_max = 0.0;
    for (int r = 0; r < nc; ++r) {
        double g1 = 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k) g1 += fabs(tcon0[SL::FcT + r * NX + k]);
        g1_max = g1 > g1_max ? g1 : g1_max;
        hs_max = tcon0[SL::hs + r] > hs_max ? tcon0[SL::hs + r] : hs_max;
    }
    auto refresh = [&](const double* tcon) {
        w_hat_b_regs<NX>(sc, tcon + SL::wb, alpha, v, bh);
        const double beta = 1.0 - alpha;
        double vmax = 0.0;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < NX; ++k) acc = fma(tcon[SL::T + i * NX + k], v[k], acc);
            bTv[i] = beta * acc;   // as tighten_set_rows()
            vmax = fabs(v[i]) > vmax ? fabs(v[i]) : vmax;
        }
        // min_r (1 - h_k[r]) only matters through its sign here: the rows are walked only when the bound cannot rule
        // a negative value out (never in the reference configurations, where 1 - h_k stays near 0.9)
        need_h = false;
        if (!(1.0 - (alpha * hs_max + beta * g1_max * vmax) > 1e-6)) {
            double bTv2[NX], momh;
            int aomh;
            tighten_set_rows<NX>(tcon + SL::T, tcon + SL::FcT, tcon + SL::hs, nc, alpha, v, bTv2, momh, aomh);
            need_h = momh < -QP_TOL_VIOL;
        }
        const double as = alpha * (1.0 - 1e-6);
        as2 = as * as;
        ell_ok = sc.hasE && alpha * 1e-6 * sc.bmin >= 1e-9;   // as in ellipsoid_outside()
    };
    refresh(tcon0);
    // ring of step disturbances, slot j holds step kb + j
    double wk[STEADY_RING][NX];
#pragma unroll
    for (int j = 0; j < STEADY_RING; ++j) {
        const int kj = a.kseg0 + j < a.kseg1 ? a.kseg0 + j : a.kseg1 - 1;
        const double* xw = a.xi + xi_index<NX>(c, kj, cn);
#pragma unroll
        for (int i = 0; i < NX; ++i) wk[j][i] = xw[i];
    }
    bool stop = false;
    for (int kb = a.kseg0; kb < a.kseg1 && !stop; kb += STEADY_RING) {
#pragma unroll
        for (int j = 0; j < STEADY_RING; ++j) {
            const int k = kb + j;
            if (k >= a.kseg1 || stop) continue;
            if (__builtin_amdgcn_ballot_w64(active) == 0) { stop = true; continue; }
            // ---- online set update (UQRobustMPC.m:124-139), in place; b_hat of the current set is kept
            double hw[2 * NX];
            bool changed = false;
#pragma unroll
            for (int r = 0; r < 2 * NX; ++r) {
                hw[r] = fma(sc.wa[r], wst[r >> 1], 0.0);
                changed = changed || hw[r] > bh[r];
            }
            int st = 0;
#ifdef STEADY_DIAG_NOCHANGE
            changed = false;
#endif
            if (__builtin_amdgcn_ballot_w64(changed) != 0) {
                int toff = 0;
                asm volatile("" : "+v"(toff));   // LDS constants are read here, not hoisted into registers for the loop
                const double* tcon = tcon0 + toff;
                if (changed) {
                    if (active) {  // should this trial stop at this step, its saved state is the one before the update
                        a.st_d[3 * NX * cn + c] = alpha;
#pragma unroll
                        for (int i = 0; i < NX; ++i) a.st_d[(1 * NX + i) * cn + c] = v[i];
                    }
                    double mm[2 * NX];
#pragma unroll
                    for (int r = 0; r < 2 * NX; ++r) mm[r] = hw[r] > bh[r] ? hw[r] : bh[r];
                    uint32_t mask;
                    st = homothety_box_rows<NX>(tcon + SL::wb, tcon + SL::ic, mm, alpha, v, mask);
                    refresh(tcon);
                }
            }
            // ---- first-pass filter at z = 0
            double q[1][NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) q[0][i] = bTv[i] - x[i];
            double e2 = 0.0;
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                double ti = 0.0;
#pragma unroll
                for (int kk = 0; kk < NX; ++kk) ti = fma(sc.ME[i * NX + kk], q[0][kk], ti);
                e2 = fma(q[0][i], ti, e2);
            }
            const bool outside = !(ell_ok && e2 <= as2) && active;
            bool need = need_h;
            nsteps += 1;
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(outside) != 0, 0)) {
                bool nf[1];
                const double an[1] = {alpha};
                filter_scan<NX, 1, 2>(C, L, an, q, nf);
                need = need || (nf[0] && outside);
                nfilter += 1;
                load_step_const<NX>(C, sc);   // the scan had their registers
            }
            if (active && need) {  // hand this trial to the solver kernel, state as it was before this step
#pragma unroll
                for (int i = 0; i < NX; ++i) { a.st_d[(0 * NX + i) * cn + c] = x[i]; a.st_d[(2 * NX + i) * cn + c] = wst[i]; }
                if (!changed) {
                    a.st_d[3 * NX * cn + c] = alpha;
#pragma unroll
                    for (int i = 0; i < NX; ++i) a.st_d[(1 * NX + i) * cn + c] = v[i];
                }
                a.st_i[0 * cn + c] = status; a.st_i[1 * cn + c] = infeasible;
                a.st_i[3 * cn + c] = hashint ? 1 : 0; a.st_i[4 * cn + c] = k;
                active = false;
            }
            if (active) {
                status = st > status ? st : status;
                if (hashint) { a.nhint[c] = 0; hashint = false; }  // no QP this step: no pivot candidates next step
                // u = K x (c = 0); plant update in relative coordinates, as in mc_kernel
                double u0 = 0.0, Ax[NX], Bu[NX], xn[NX];
#pragma unroll
                for (int i = 0; i < NX; ++i) u0 = fma(sc.K[i], x[i], u0);
                if (u0 > accb || u0 < -accb) infeasible = 1;
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    double ax = 0.0;
#pragma unroll
                    for (int kk = 0; kk < NX; ++kk) ax = fma(sc.A[i * NX + kk], x[kk], ax);
                    Ax[i] = ax; Bu[i] = fma(sc.B[i], u0, 0.0);
                    xn[i] = (Ax[i] + Bu[i]) + wk[j][i];
                }
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    wst[i] = (xn[i] - Ax[i]) - Bu[i];
                    x[i] = xn[i];
                    if (xn[i] > derr || xn[i] < -derr) infeasible = 1;
                }
                if (tl < a.traj_trials) {
                    if (a.out.traj_x) for (int i = 0; i < NX; ++i) a.out.traj_x[(tl * (K + 1) + k + 1) * NX + i] = x[i];
                    if (a.out.traj_u) a.out.traj_u[tl * K + k] = u0;
                    if (a.out.traj_alpha) a.out.traj_alpha[tl * (K + 1) + k + 1] = alpha;
                    if (a.out.traj_s0) for (int i = 0; i < NX; ++i) a.out.traj_s0[(tl * K + k) * NX + i] = 0.0;
                }
            }
            // slot j is free: fetch step k + RING into it
            if (k + STEADY_RING < a.kseg1) {
                const double* xw = a.xi + xi_index<NX>(c, k + STEADY_RING, cn);
#pragma unroll
                for (int i = 0; i < NX; ++i) wk[j][i] = xw[i];
            }
        }
    }
    if (active) {  // reached the end of the segment (= the horizon: this kernel always runs to a.steps)
        a.st_i[4 * cn + c] = a.kseg1;
        if (a.out.infeasible) a.out.infeasible[tl] = infeasible;
        if (a.out.status) a.out.status[tl] = status;
        if (a.out.qp_iters) a.out.qp_iters[tl] = a.st_i[2 * cn + c];
        if (a.out.alpha_final) a.out.alpha_final[tl] = alpha;
        if (a.out.v_final) for (int i = 0; i < NX; ++i) a.out.v_final[tl * NX + i] = v[i];
        if (a.out.x_final) for (int i = 0; i < NX; ++i) a.out.x_final[tl * NX + i] = x[i];
    }
    if (lane_id() == 0 && a.counters) {
        atomicAdd(a.counters + 0, (unsigned long long)nfilter);
        atomicAdd(a.counters + 1, (unsigned long long)nsteps);
    }
}

template <class Kern>
static int allow_lds_steady(Kern k, size_t bytes) {
    LB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}

int launch_mc_sample(int nx, int lv_hasU, int ev_hasU, dim3 sg, dim3 sb, hipStream_t st, const SamplerDev* sd, int64_t cn,
                     int64_t trial_first, int K, uint64_t seed, int max_attempts, double* xi, int* xi_status) {
#define LS(NXV, L, E) hipLaunchKernelGGL((mc_sample_kernel<NXV, L, E>), sg, sb, 0, st, sd, cn, trial_first, K, seed, max_attempts, xi, xi_status)
#define LS_NX(NXV)                                           \
    do {                                                     \
        if (lv_hasU) { if (ev_hasU) LS(NXV, true, true); else LS(NXV, true, false); } \
        else { if (ev_hasU) LS(NXV, false, true); else LS(NXV, false, false); }       \
    } while (0)
    switch (nx) {
        case 1: LS_NX(1); break;
        case 2: LS_NX(2); break;
        case 3: LS_NX(3); break;
        default: LS_NX(4); break;
    }
#undef LS_NX
#undef LS
    LB_HIP(hipGetLastError());
    return 0;
}

int launch_mc_steady(const lbmpc_ctx* c, dim3 g2, dim3 b2, size_t lds_steady, hipStream_t st, const McArgs& a) {
#define LAUNCH_STEADY(KERN)                                                                     \
    do {                                                                                        \
        if (int rc = allow_lds_steady(KERN, lds_steady)) return rc;                             \
        hipLaunchKernelGGL((KERN), g2, b2, lds_steady, st, c->d, a);                            \
    } while (0)
    switch (c->h.nx) {
        case 1: LAUNCH_STEADY(mc_steady_kernel<1>); break;
        case 2: LAUNCH_STEADY(mc_steady_kernel<2>); break;
        case 3: LAUNCH_STEADY(mc_steady_kernel<3>); break;
        default: LAUNCH_STEADY(mc_steady_kernel<4>); break;
    }
#undef LAUNCH_STEADY
    LB_HIP(hipGetLastError());
    return 0;
}

}  // namespace lbmpc
