// common.hpp -- shared host/device helpers of liblbmpc_hip.so (gfx950 only).
//
// Arithmetic contract (DESIGN.md "Arithmetic contract"): the library is built with
// -ffp-contract=off; every dot product is an explicit fma chain in ascending index order from a
// stated initial value; max/min reductions are order-free; argmin ties go to the lowest row.  The
// CPU oracle follows the same contract, so flags, active sets and maxima agree exactly.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lbmpc.h"

namespace lbmpc {

// ---- host-side error plumbing -----------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_device();        // 0 if lbmpc_init validated the calling thread's current device, else LBMPC_ERR_NODEVICE
int check_any_device();    // 0 if lbmpc_init validated some device (entry points that take a context)
int device_ready(int device);  // lbmpc_init validated `device`
int resolve_noalloc_scratch(int device);  // set_learning.hip: device address of the static scratch array on `device` (current)

// Makes `device` current for the lifetime of the object and restores the caller's device afterwards
// (entry points that take a context run on the context's device whatever the caller had selected).
class DeviceGuard {
public:
    explicit DeviceGuard(int device);
    ~DeviceGuard();
    int rc() const { return rc_; }  // 0 or LBMPC_ERR_HIP (message set)
private:
    int prev_ = -1, rc_ = 0;
    bool switched_ = false;
};

#define LB_HIP(expr)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            ::lbmpc::set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
            return LBMPC_ERR_HIP;                                                         \
        }                                                                                 \
    } while (0)

// RAII device buffer for the host-pointer entry points
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
    template <class T> T* as() { return static_cast<T*>(p); }
};

// ---- device helpers ---------------------------------------------------------------------------
#define LB_WAVE 64
#define LB_INF __builtin_inf()

__device__ __forceinline__ int lane_id() { return threadIdx.x & (LB_WAVE - 1); }

// Issue priority of the two waves that share a SIMD in an 8-wave block (waves w and w + 4).  The arbiter serves the
// older wave first, so the younger one of every pair falls behind and then runs alone, at the single-wave issue rate;
// and the kernel ends with its slowest wave.  prio_tick() is called at every closed-loop step and solver trip:
// LBMPC_PRIO_MODE 1: the two take turns, periods of 2^SHIFT shader clocks; 3: the wave that is behind (lower step
// index, published in LDS: lb_prog) goes first, turns as in mode 1 while they are level; 0: hardware order.
#if defined(LBMPC_PROFILE) && !defined(LBMPC_PROFILE_LIGHT)
#define LBMPC_PROFILE_FULL   // per-phase clocks inside the closed-loop kernels; LBMPC_PROFILE_LIGHT keeps only the per-wave lifetime
#endif
#ifndef LBMPC_PRIO_MODE
#define LBMPC_PRIO_MODE 3
#endif
#ifndef LBMPC_PRIO_SHIFT
#define LBMPC_PRIO_SHIFT 18    // nx <= 3
#endif
#ifndef LBMPC_PRIO_SHIFT4
#define LBMPC_PRIO_SHIFT4 20   // nx = 4
#endif
#if LBMPC_PRIO_MODE == 3
__shared__ int lb_prog[8];   // step index of each wave of an 8-wave block (static LDS: 32 B on top of the dynamic allocation)
#endif
template <int SHIFT>
__device__ __forceinline__ void prio_tick(int mine /* step index; < 0: the one published last */) {
#if LBMPC_PRIO_MODE != 0
    if (blockDim.x != 512) return;   // one wave per SIMD: nothing to arbitrate
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned t = (unsigned)(__builtin_readcyclecounter() >> SHIFT);
    bool hi = ((t ^ (unsigned)(wave >> 2)) & 1u) != 0;
#if LBMPC_PRIO_MODE == 3
    volatile int* vp = lb_prog;
    if (mine >= 0) { if ((threadIdx.x & 63) == 0) vp[wave] = mine; } else mine = vp[wave];
    mine = __builtin_amdgcn_readfirstlane(mine);
    const int other = __builtin_amdgcn_readfirstlane(vp[wave ^ 4]);
    hi = mine < other || (mine == other && hi);
#endif
    if (hi) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#endif
}

// Orders LDS traffic of one wave: LDS ops of a wave execute in issue order, so a compiler-level
// fence plus a wave barrier is enough to hand data between lanes of the same wave through LDS.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ double readlane_f64(double v, int src_lane /* wave-uniform */) {
    union { double d; int i[2]; } a, b;
    a.d = v;
    b.i[0] = __builtin_amdgcn_readlane(a.i[0], src_lane);
    b.i[1] = __builtin_amdgcn_readlane(a.i[1], src_lane);
    return b.d;
}

// whole-wave rotate by one lane (DPP wave_rol:1): lane l receives lane (l+1) & 63
__device__ __forceinline__ double wave_rol1_f64(double v) {
    union { double d; int i[2]; } a, b;
    a.d = v;
    b.i[0] = __builtin_amdgcn_update_dpp(a.i[0], a.i[0], 0x134, 0xf, 0xf, false);
    b.i[1] = __builtin_amdgcn_update_dpp(a.i[1], a.i[1], 0x134, 0xf, 0xf, false);
    return b.d;
}

// DPP lane exchanges inside a 16-lane row: CTRL 0xB1 = quad_perm[1,0,3,2] (lane^1), 0x4E = quad_perm[2,3,0,1]
// (lane^2), 0x141 = row_half_mirror (7-i within 8), 0x140 = row_mirror (15-i within 16).  All lanes must be active.
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, true); }
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    union { double d; int i[2]; } a, b;
    a.d = v;
    b.i[0] = dpp_i32<CTRL>(a.i[0]);
    b.i[1] = dpp_i32<CTRL>(a.i[1]);
    return b.d;
}

// Wave-wide min / max (every lane gets the result): 4 DPP steps reduce each 16-lane row, the 4 row results are
// combined through readlane.  No LDS traffic (the __shfl_xor forms go through ds_bpermute).
__device__ __forceinline__ double wave_min(double v) {
    v = __builtin_fmin(v, dpp_f64<0xB1>(v));
    v = __builtin_fmin(v, dpp_f64<0x4E>(v));
    v = __builtin_fmin(v, dpp_f64<0x141>(v));
    v = __builtin_fmin(v, dpp_f64<0x140>(v));
    return __builtin_fmin(__builtin_fmin(readlane_f64(v, 0), readlane_f64(v, 16)),
                          __builtin_fmin(readlane_f64(v, 32), readlane_f64(v, 48)));
}
__device__ __forceinline__ double wave_max(double v) {
    v = __builtin_fmax(v, dpp_f64<0xB1>(v));
    v = __builtin_fmax(v, dpp_f64<0x4E>(v));
    v = __builtin_fmax(v, dpp_f64<0x141>(v));
    v = __builtin_fmax(v, dpp_f64<0x140>(v));
    return __builtin_fmax(__builtin_fmax(readlane_f64(v, 0), readlane_f64(v, 16)),
                          __builtin_fmax(readlane_f64(v, 32), readlane_f64(v, 48)));
}
__device__ __forceinline__ int wave_min_i32(int v) {
    int o;
    o = dpp_i32<0xB1>(v); v = o < v ? o : v;
    o = dpp_i32<0x4E>(v); v = o < v ? o : v;
    o = dpp_i32<0x141>(v); v = o < v ? o : v;
    o = dpp_i32<0x140>(v); v = o < v ? o : v;
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    const int ab = a < b ? a : b, cd = c < d ? c : d;
    return ab < cd ? ab : cd;
}

// (value, index) arg-min with ties to the lowest index; every lane gets the result.
__device__ __forceinline__ void wave_argmin(double& v, int& idx) {
    const double best = wave_min(v);
    idx = wave_min_i32(v == best ? idx : 0x7fffffff);
    v = best;
}

// ---- RNG: Philox4x32-10, bit-identical to oracle/lbmpc_oracle.c ------------------------------
#ifndef PHILOX_DIAG_ROUNDS
#define PHILOX_DIAG_ROUNDS 10
#endif
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < PHILOX_DIAG_ROUNDS; ++r) {
        // one 32x32->64 product per multiplier (v_mad_u64_u32) instead of separate mul_hi / mul_lo
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        // three-input xor in one v_bitop3_b32 (truth table 0x96) instead of two v_xor_b32
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c[1], k0, 0x96);
        const uint32_t n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c[3], k1, 0x96);
        c[0] = n0; c[1] = (uint32_t)p1; c[2] = n2; c[3] = (uint32_t)p0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

// One 32-bit word -> uniform in (0, 1): (w + 1/2) 2^-32, exact in fp64 (granularity 2^-32 of the box width; the
// end points are never returned).  One Philox block therefore serves a whole proposal point (nx <= 4) or input
// draw (nu <= 4): a trial-step costs 3 blocks (xi_LV, u_LV, xi_EV) instead of the 5 of 53-bit uniforms.
// (one fma: w 2^-32 + 2^-33 has 33 significant bits, so it is exact and equal to the two-operation form of the oracle)
__device__ __forceinline__ double u01(uint32_t w) { return fma((double)w, 0x1.0p-32, 0x1.0p-33); }

enum { P_I0_U = 0, P_I0_EV = 1, P_I0_LV = 2, P_ST_LV = 3, P_ST_U = 4, P_ST_EV = 5 };

// N <= 4 uniforms of stream (seed, trial): draw (index, purpose, attempt) is ONE Philox4x32-10 block
template <int N>
__device__ __forceinline__ void draw_uniforms(uint64_t seed, uint64_t trial, uint32_t index,
                                              uint32_t purpose, uint32_t attempt, double* out) {
    static_assert(N <= 4, "one Philox block per draw");
    uint32_t c[4] = {(uint32_t)trial, (uint32_t)(trial >> 32), index, (purpose << 28) | (attempt << 4)};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
    for (int k = 0; k < N; ++k) out[k] = u01(c[k]);
}

// Device-resident description of the two disturbance polytopes + input range (lbmpc_sampler),
// rows stored row-major.  Read through wave-uniform (scalar) loads.
struct SamplerDev {
    int nx, nu, nEV, nLV;
    double EV_A[LBMPC_MAX_NXI * LBMPC_MAX_NX], EV_b[LBMPC_MAX_NXI], EV_lo[LBMPC_MAX_NX], EV_hi[LBMPC_MAX_NX];
    double LV_A[LBMPC_MAX_NXI * LBMPC_MAX_NX], LV_b[LBMPC_MAX_NXI], LV_lo[LBMPC_MAX_NX], LV_hi[LBMPC_MAX_NX];
    double B[LBMPC_MAX_NX * LBMPC_MAX_NU];  // row-major nx x nu
    double min_u[LBMPC_MAX_NU], max_u[LBMPC_MAX_NU];
    double EV_U[LBMPC_MAX_NX * LBMPC_MAX_NX], LV_U[LBMPC_MAX_NX * LBMPC_MAX_NX];  // row-major proposal bases
    int EV_hasU, LV_hasU;                                                         // 0: U = I (axis-aligned box)
};
int pack_sampler(const lbmpc_sampler* s, SamplerDev* out);  // host: validate + repack

// The three draws of one information-set sample as descriptors of one layout (sample_sets_kernel, set_learning.hip)
struct SmpPhase {
    double lo[LBMPC_MAX_NX], hi[LBMPC_MAX_NX], U[LBMPC_MAX_NX * LBMPC_MAX_NX];
    double A[LBMPC_MAX_NXI * LBMPC_MAX_NX], b[LBMPC_MAX_NXI];
    int hasU, purpose, pad0, pad1;
};
struct SmpPhases {
    SmpPhase ph[3];   // 0: u_LV, 1: xi_EV, 2: xi_LV
    int nr4, pad[3];  // rows to test, a multiple of 4 (rows beyond a polytope's own read 0.x <= inf)
    double W[LBMPC_MAX_NW * LBMPC_MAX_NX];   // W_A rows, row-major with stride nx (by value they crowd the SGPRs)
};
int pack_sample_phases(const SamplerDev& h, const double* W_rows /* nW x nx row-major */, int nW, SmpPhases* o);  // host
// m[b, j] = max_i W_rows[j].w_{b,i} over N generated samples of information sets set0 + b (one Philox stream per set), desc on the device
int launch_sample_sets(int nx, int nu, const SmpPhases* desc_dev, int nW, int64_t N, int64_t batch, uint64_t seed, uint64_t set0,
                       int max_attempts, double* samples_out_dev, double* m_out_dev, int* status_out_dev, hipStream_t st);

// candidate x = U t, t = lo + (hi - lo) u  (U = I when null)
template <int NX>
__device__ __forceinline__ void proposal_point(const double* __restrict__ lo, const double* __restrict__ hi,
                                               const double* __restrict__ U, const double* u, double* x) {
    double t[NX];
#pragma unroll
    for (int k = 0; k < NX; ++k) t[k] = lo[k] + (hi[k] - lo[k]) * u[k];
    if (!U) {
#pragma unroll
        for (int k = 0; k < NX; ++k) x[k] = t[k];
    } else {
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            double acc = 0.0;
#pragma unroll
            for (int i = 0; i < NX; ++i) acc = fma(U[k * NX + i], t[i], acc);
            x[k] = acc;
        }
    }
}

// Rejection sampling from the vertex bounding box against A x <= b
// (InitialSetComputation.m:101-122, ModelingCar.m:47-91).  Returns true at the attempt cap.
template <int NX>
__device__ __forceinline__ bool sample_polytope(const double* __restrict__ XA, const double* __restrict__ Xb,
                                                int nr, const double* __restrict__ lo,
                                                const double* __restrict__ hi, const double* __restrict__ U /* row-major or null */,
                                                uint64_t seed, uint64_t trial,
                                                uint32_t index, uint32_t purpose, int max_attempts, double* x) {
    bool done = false;
    for (int a = 0; a < max_attempts; ++a) {
        if (!done) {
            double u[NX];
            double xt[NX];
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
        if (__builtin_amdgcn_ballot_w64(!done) == 0) break;  // every active lane accepted
    }
    return !done;
}

// u_LV ~ U[min_u, max_u]^nu; returns B u (nx).  One Philox block (nu <= 4).
template <int NX>
__device__ __forceinline__ void sample_Bu_lv(const SamplerDev* __restrict__ s, uint64_t seed, uint64_t trial,
                                             uint32_t index, uint32_t purpose, double* Bu) {
    double acc[NX], r[LBMPC_MAX_NU];
#pragma unroll
    for (int i = 0; i < NX; ++i) acc[i] = 0.0;
    const int nu = s->nu;
    draw_uniforms<LBMPC_MAX_NU>(seed, trial, index, purpose, 0u, r);
#pragma unroll
    for (int j = 0; j < LBMPC_MAX_NU; ++j)
        if (j < nu) {
            const double uj = (s->max_u[j] - s->min_u[j]) * r[j] + s->min_u[j];
#pragma unroll
            for (int i = 0; i < NX; ++i) acc[i] = fma(s->B[i * nu + j], uj, acc[i]);
        }
#pragma unroll
    for (int i = 0; i < NX; ++i) Bu[i] = acc[i];
}

// One information-set sample w = xi_EV - xi_LV - B u_LV (InitialSetComputation.m:51-52)
template <int NX>
__device__ __forceinline__ bool sample_w_I0(const SamplerDev* __restrict__ s, uint64_t seed, uint64_t set,
                                            uint32_t i, int max_attempts, double* w) {
    double ev[NX], lv[NX], Bu[NX];
    sample_Bu_lv<NX>(s, seed, set, i, P_I0_U, Bu);
    bool cap = sample_polytope<NX>(s->EV_A, s->EV_b, s->nEV, s->EV_lo, s->EV_hi, s->EV_hasU ? s->EV_U : nullptr, seed, set, i, P_I0_EV, max_attempts, ev);
    cap |= sample_polytope<NX>(s->LV_A, s->LV_b, s->nLV, s->LV_lo, s->LV_hi, s->LV_hasU ? s->LV_U : nullptr, seed, set, i, P_I0_LV, max_attempts, lv);
#pragma unroll
    for (int k = 0; k < NX; ++k) w[k] = (ev[k] - lv[k]) - Bu[k];
    return cap;
}

// Box structure of the prior W in canonical row order: row 2k = +e_k row, row 2k+1 = -e_k row.
struct BoxW {
    int nx, nW;
    int perm[LBMPC_MAX_NW];                    // canonical row r is caller row perm[r]
    double WA[LBMPC_MAX_NW * LBMPC_MAX_NX];    // canonical rows, row-major (caller's values)
    double Wb[LBMPC_MAX_NW];
    double ic[LBMPC_MAX_NW];                   // 1 / |axis coefficient|
    int exact;                                 // every off-axis coefficient is an exact zero: row r = WA[r][r/2] e_{r/2}
};
int pack_box(const double* WA, const double* Wb, int nW, int nx, BoxW* out);  // host

// Closed form of the homothety LP for a box W with the LBMPC_TIE_CENTER rule; `m` in canonical
// row order.  Same operation order as orc_homothety_box.
template <int NX>
__device__ __forceinline__ int homothety_box_rows(const double* Wb, const double* ic, const double* m, double& alpha,
                                                  double* v, uint32_t& mask) {
    double Bp[NX], Bq[NX], Mp[NX], Mq[NX], rho[NX];
    bool infeas = false;
    double a = 0.0;
#pragma unroll
    for (int k = 0; k < NX; ++k) {
        const int p = 2 * k, q = 2 * k + 1;
        Bp[k] = Wb[p] * ic[p]; Bq[k] = Wb[q] * ic[q];
        Mp[k] = m[p] * ic[p];      Mq[k] = m[q] * ic[q];
        infeas = infeas || (m[p] > Wb[p]) || (m[q] > Wb[q]);
        rho[k] = (Mp[k] + Mq[k]) / (Bp[k] + Bq[k]);
        a = rho[k] > a ? rho[k] : a;
    }
    if (infeas) {
        alpha = 1.0; mask = 0;
#pragma unroll
        for (int k = 0; k < NX; ++k) v[k] = 0.0;
        return LBMPC_ST_LP_INFEASIBLE;
    }
    const double beta = 1.0 - a;
    uint32_t mk = 0;
#pragma unroll
    for (int k = 0; k < NX; ++k) {
        double lo1 = Mp[k] - a * Bp[k], lo2 = -(beta * Bq[k]);
        double hi1 = beta * Bp[k], hi2 = a * Bq[k] - Mq[k];
        double lo = lo1 > lo2 ? lo1 : lo2;
        double hi = hi1 < hi2 ? hi1 : hi2;
        double y = 0.5 * (lo + hi);
        v[k] = beta > 0.0 ? y / beta : 0.0;
        if (a > 0.0 && rho[k] >= a) mk |= 1u << k;
    }
    alpha = a; mask = mk;
    return 0;
}

template <int NX>
__device__ __forceinline__ int homothety_box(const BoxW* __restrict__ bw, const double* m, double& alpha,
                                             double* v, uint32_t& mask) {
    return homothety_box_rows<NX>(bw->Wb, bw->ic, m, alpha, v, mask);
}

}  // namespace lbmpc
