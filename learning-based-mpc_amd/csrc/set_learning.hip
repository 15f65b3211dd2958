// set_learning.hip -- subsystem 2: homothetic disturbance-set learning over I_0^w.
//
//   halfspace_max_kernel : m[b,j] = max_i W_A[j,:].w_{b,i}   (HBM stream, 8*nx bytes/sample)
//   sample_sets_kernel   : on-device generation of the information sets + their maxima
//   homothety_lp_kernel  : the (nx+1)-variable LP in closed form, one lane per information set
//
// Replaces Functions_general/InitialSetComputation.m:48-99 (sampling loop, LP build, IPOPT solve)
// and the sample rows of Functions_general/UQRobustMPC.m:124-139.
#include <atomic>

#include "common.hpp"
#include "homothety_gen.hpp"

namespace lbmpc {

struct RowsW {  // W_A rows (caller order), row-major, passed by value -> scalar registers
    double a[LBMPC_MAX_NW * LBMPC_MAX_NX];
};

// ---- halfspace max ----------------------------------------------------------------------------
// One wave per (information set, chunk) unit.  Lane l owns SPL consecutive samples per step, so a
// lane's load is SPL*NX*8 bytes (a multiple of 16 when VEC: read as double2), wave-contiguous.
// UNR independent steps are kept in flight.  Partial maxima go to part[(b*chunks + c)*NW + j].
template <int NX, int NW, int SPL, bool VEC>
__global__ __launch_bounds__(256) void halfspace_max_kernel(const double* __restrict__ samples, int64_t N,
                                                            int64_t batch, int64_t stride, int chunks,
                                                            int64_t spc /* samples per chunk, multiple of 64*SPL */,
                                                            RowsW W, double* __restrict__ part) {
    const int lane = lane_id();
    const int64_t unit = (int64_t)blockIdx.x * (blockDim.x / LB_WAVE) + (threadIdx.x / LB_WAVE);
    if (unit >= batch * chunks) return;
    const int64_t b = unit / chunks;
    const int c = (int)(unit % chunks);
    const double* __restrict__ S = samples + b * stride;
    const int64_t i0 = (int64_t)c * spc;
    const int64_t i1 = (i0 + spc < N) ? i0 + spc : N;

    double m[NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) m[j] = -LB_INF;

    constexpr int UNR = 4;
    constexpr int PER = SPL * NX;  // doubles per lane per step
    int64_t i = i0 + (int64_t)lane * SPL;
    // full steps: all SPL samples of all UNR steps of this lane in range
    for (; i + (int64_t)(UNR - 1) * 64 * SPL + SPL <= i1; i += (int64_t)UNR * 64 * SPL) {
        double w[UNR][PER];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const double* p = S + (i + (int64_t)u * 64 * SPL) * NX;
            if constexpr (VEC) {
#pragma unroll
                for (int q = 0; q < PER / 2; ++q) {
                    double2 t = reinterpret_cast<const double2*>(p)[q];
                    w[u][2 * q] = t.x; w[u][2 * q + 1] = t.y;
                }
            } else {
#pragma unroll
                for (int q = 0; q < PER; ++q) w[u][q] = p[q];
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
#pragma unroll
            for (int s = 0; s < SPL; ++s)
#pragma unroll
                for (int j = 0; j < NW; ++j) {
                    double acc = 0.0;
#pragma unroll
                    for (int k = 0; k < NX; ++k) acc = fma(W.a[j * NX + k], w[u][s * NX + k], acc);
                    m[j] = acc > m[j] ? acc : m[j];
                }
    }
    // tail: one sample at a time
    for (; i < i1; i += 64 * SPL) {
        for (int s = 0; s < SPL && i + s < i1; ++s) {
            const double* p = S + (i + s) * NX;
            double w[NX];
#pragma unroll
            for (int k = 0; k < NX; ++k) w[k] = p[k];
#pragma unroll
            for (int j = 0; j < NW; ++j) {
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < NX; ++k) acc = fma(W.a[j * NX + k], w[k], acc);
                m[j] = acc > m[j] ? acc : m[j];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NW; ++j) m[j] = wave_max(m[j]);
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < NW; ++j) part[unit * NW + j] = m[j];
    }
}

// generic row count (nW not 2*nx): rows looped at run time in groups of 4
template <int NX>
__global__ __launch_bounds__(256) void halfspace_max_generic_kernel(const double* __restrict__ samples, int64_t N,
                                                                    int64_t batch, int64_t stride, int chunks,
                                                                    int64_t spc, int nW, RowsW W,
                                                                    double* __restrict__ part) {
    const int lane = lane_id();
    const int64_t unit = (int64_t)blockIdx.x * (blockDim.x / LB_WAVE) + (threadIdx.x / LB_WAVE);
    if (unit >= batch * chunks) return;
    const int64_t b = unit / chunks;
    const int c = (int)(unit % chunks);
    const double* __restrict__ S = samples + b * stride;
    const int64_t i0 = (int64_t)c * spc;
    const int64_t i1 = (i0 + spc < N) ? i0 + spc : N;
    double m[LBMPC_MAX_NW];
#pragma unroll
    for (int j = 0; j < LBMPC_MAX_NW; ++j) m[j] = -LB_INF;
    for (int64_t i = i0 + lane; i < i1; i += 64) {
        double w[NX];
#pragma unroll
        for (int k = 0; k < NX; ++k) w[k] = S[i * NX + k];
#pragma unroll
        for (int j = 0; j < LBMPC_MAX_NW; ++j)
            if (j < nW) {
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < NX; ++k) acc = fma(W.a[j * NX + k], w[k], acc);
                m[j] = acc > m[j] ? acc : m[j];
            }
    }
#pragma unroll
    for (int j = 0; j < LBMPC_MAX_NW; ++j)
        if (j < nW) {
            double r = wave_max(m[j]);
            if (lane == 0) part[unit * nW + j] = r;
        }
}

__global__ void finish_max_kernel(const double* __restrict__ part, int64_t batch, int chunks, int nW,
                                  double* __restrict__ m_out) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= batch * nW) return;
    int64_t b = t / nW;
    int j = (int)(t % nW);
    double m = -LB_INF;
    for (int c = 0; c < chunks; ++c) {
        double x = part[(b * chunks + c) * nW + j];
        m = x > m ? x : m;
    }
    m_out[t] = m;
}

static void pack_rows(const double* WA, int nW, int nx, RowsW* r) {
    for (int j = 0; j < nW; ++j)
        for (int k = 0; k < nx; ++k) r->a[j * nx + k] = WA[j + (size_t)nW * k];
}

template <int NX, int NW, int SPL>
static void launch_hmax(bool vec, dim3 grid, hipStream_t st, const double* s, int64_t N, int64_t batch,
                        int64_t stride, int chunks, int64_t spc, const RowsW& W, double* part) {
    if (vec)
        hipLaunchKernelGGL((halfspace_max_kernel<NX, NW, SPL, true>), grid, dim3(256), 0, st, s, N, batch, stride, chunks, spc, W, part);
    else
        hipLaunchKernelGGL((halfspace_max_kernel<NX, NW, SPL, false>), grid, dim3(256), 0, st, s, N, batch, stride, chunks, spc, W, part);
}

// chunking: enough wave-units to fill 256 CUs x 32 waves twice, chunk a multiple of the lane step
static void plan_chunks(int64_t N, int64_t batch, int spl, int* chunks, int64_t* spc) {
    const int64_t step = 64 * spl * 4;  // one full unrolled step of a wave
    const int64_t want_units = 256 * 32 * 2;
    int64_t c = (want_units + batch - 1) / batch;
    int64_t maxc = (N + step - 1) / step;
    if (c > maxc) c = maxc;
    if (c < 1) c = 1;
    if (c > 65535) c = 65535;
    int64_t per = (N + c - 1) / c;
    per = ((per + step - 1) / step) * step;
    c = (N + per - 1) / per;
    *chunks = (int)c;
    *spc = per;
}

}  // namespace lbmpc

using namespace lbmpc;

// Shared body of lbmpc_halfspace_max_dev / lbmpc_halfspace_max_noalloc: `part` receives the per-(set, chunk) partial
// maxima when the plan splits sets over several waves (units * nW doubles), else the kernel writes m_out directly.
static int hmax_validate(const double* WA, int nW, int nx, const double* samples_dev, int64_t N, int64_t batch,
                         int64_t batch_stride, double* m_out_dev) {
    if (!WA || !samples_dev || !m_out_dev || N < 1 || batch < 0 || nW < 1 || nx < 1) {
        set_error("lbmpc_halfspace_max: bad argument");
        return LBMPC_ERR_ARG;
    }
    if (nx > LBMPC_MAX_NX || nW > LBMPC_MAX_NW) { set_error("lbmpc_halfspace_max: nx<=%d, nW<=%d", LBMPC_MAX_NX, LBMPC_MAX_NW); return LBMPC_ERR_DIM; }
    if (batch > 1 && batch_stride < N * nx) { set_error("lbmpc_halfspace_max: batch_stride < N*nx"); return LBMPC_ERR_ARG; }
    return 0;
}
static int hmax_spl(int nx) { return (nx == 1) ? 2 : (nx == 3 ? 2 : 1); }

static int hmax_launch(const double* WA, int nW, int nx, const double* samples_dev, int64_t N, int64_t batch,
                       int64_t batch_stride, double* m_out_dev, int chunks, int64_t spc, double* part_buf,
                       hipStream_t st) {
    RowsW W;
    pack_rows(WA, nW, nx, &W);
    const int64_t units = batch * chunks;
    dim3 grid((unsigned)((units + 3) / 4));
    double* part = chunks > 1 ? part_buf : m_out_dev;
    const bool vec = ((uintptr_t)samples_dev % 16 == 0) && ((batch_stride * 8) % 16 == 0 || batch == 1);
    const bool boxrows = (nW == 2 * nx);
    if (boxrows && nx == 1) launch_hmax<1, 2, 2>(vec, grid, st, samples_dev, N, batch, batch_stride, chunks, spc, W, part);
    else if (boxrows && nx == 2) launch_hmax<2, 4, 1>(vec, grid, st, samples_dev, N, batch, batch_stride, chunks, spc, W, part);
    else if (boxrows && nx == 3) launch_hmax<3, 6, 2>(vec, grid, st, samples_dev, N, batch, batch_stride, chunks, spc, W, part);
    else if (boxrows && nx == 4) launch_hmax<4, 8, 1>(vec, grid, st, samples_dev, N, batch, batch_stride, chunks, spc, W, part);
    else {
        // generic kernel steps one sample per lane: chunk boundaries need no SPL alignment
        switch (nx) {
            case 1: hipLaunchKernelGGL(halfspace_max_generic_kernel<1>, grid, dim3(256), 0, st, samples_dev, N, batch, batch_stride, chunks, spc, nW, W, part); break;
            case 2: hipLaunchKernelGGL(halfspace_max_generic_kernel<2>, grid, dim3(256), 0, st, samples_dev, N, batch, batch_stride, chunks, spc, nW, W, part); break;
            case 3: hipLaunchKernelGGL(halfspace_max_generic_kernel<3>, grid, dim3(256), 0, st, samples_dev, N, batch, batch_stride, chunks, spc, nW, W, part); break;
            default: hipLaunchKernelGGL(halfspace_max_generic_kernel<4>, grid, dim3(256), 0, st, samples_dev, N, batch, batch_stride, chunks, spc, nW, W, part); break;
        }
    }
    LB_HIP(hipGetLastError());
    if (chunks > 1) {
        const int64_t n = batch * nW;
        hipLaunchKernelGGL(finish_max_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, part, batch, chunks, nW, m_out_dev);
        LB_HIP(hipGetLastError());
    }
    return 0;
}

// An information set split over several waves (chunks > 1) leaves partial maxima in a stream-ordered scratch buffer
// (hipMallocAsync / hipFreeAsync on the caller's stream: no synchronisation), folded by finish_max_kernel.
extern "C" int lbmpc_halfspace_max_dev(const double* WA, int nW, int nx, const double* samples_dev, int64_t N,
                                       int64_t batch, int64_t batch_stride, double* m_out_dev, void* stream) {
    if (int rc = check_device()) return rc;
    if (int rc = hmax_validate(WA, nW, nx, samples_dev, N, batch, batch_stride, m_out_dev)) return rc;
    if (batch == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    int chunks; int64_t spc;
    plan_chunks(N, batch, hmax_spl(nx), &chunks, &spc);
    void* tmp = nullptr;
    if (chunks > 1) LB_HIP(hipMallocAsync(&tmp, sizeof(double) * batch * chunks * nW, st));
    const int rc = hmax_launch(WA, nW, nx, samples_dev, N, batch, batch_stride, m_out_dev, chunks, spc, (double*)tmp, st);
    if (tmp) LB_HIP(hipFreeAsync(tmp, st));
    return rc;
}

// ---- allocation-free variant (no counterpart in the reference; see include/lbmpc.h) --------------------------------
// Same kernels, same chunk plan and therefore the same partition of the work as lbmpc_halfspace_max_dev; the partial
// maxima of sets split over several waves go to a file-scope device array instead of a hipMallocAsync buffer.  Every
// GPU holds its own copy of the array at its own address: the address is resolved once per device by
// resolve_noalloc_scratch (called from lbmpc_init with that device current) and host code only ever uses that pointer.
// Nothing in the array has to start at zero: every partial a launch reads was written by that launch's first kernel.
static __device__ __align__(256) double scratch[LBMPC_NOALLOC_MAX_PARTIALS];
static std::atomic<double*> g_scratch_ptr[64];   // per device: device address of `scratch` on that GPU (nullptr: not resolved); lbmpc_init may run on several threads

namespace lbmpc {
int resolve_noalloc_scratch(int device /* current on the calling thread */) {
    if (device < 0 || device >= 64) { set_error("resolve_noalloc_scratch: device %d out of range", device); return LBMPC_ERR_ARG; }
    if (g_scratch_ptr[device].load(std::memory_order_acquire)) return 0;
    double* ptr = nullptr;
    const hipError_t e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !ptr) {
        set_error("hipGetSymbolAddress(scratch) on device %d -> %s", device, hipGetErrorString(e));
        return LBMPC_ERR_HIP;
    }
    g_scratch_ptr[device].store(ptr, std::memory_order_release);
    return 0;
}
}  // namespace lbmpc

extern "C" int lbmpc_halfspace_max_noalloc(const double* WA, int nW, int nx, const double* samples_dev, int64_t N,
                                           int64_t batch, int64_t batch_stride, double* m_out_dev, void* stream) {
    if (int rc = check_device()) return rc;
    if (int rc = hmax_validate(WA, nW, nx, samples_dev, N, batch, batch_stride, m_out_dev)) return rc;
    if (batch == 0) return 0;
    int device = -1;
    LB_HIP(hipGetDevice(&device));
    double* ptr = (device >= 0 && device < 64) ? g_scratch_ptr[device].load(std::memory_order_acquire) : nullptr;
    if (!ptr) { set_error("lbmpc_halfspace_max_noalloc: scratch of device %d was not resolved by lbmpc_init", device); return LBMPC_ERR_NODEVICE; }
    int chunks; int64_t spc;
    plan_chunks(N, batch, hmax_spl(nx), &chunks, &spc);
    if (chunks > 1 && batch * chunks * nW > (int64_t)LBMPC_NOALLOC_MAX_PARTIALS) {
        set_error("lbmpc_halfspace_max_noalloc: %lld sets x %d chunks x %d rows = %lld partial maxima exceed the static scratch "
                  "(LBMPC_NOALLOC_MAX_PARTIALS = %d doubles); split the batch or use lbmpc_halfspace_max_dev",
                  (long long)batch, chunks, nW, (long long)(batch * chunks * nW), (int)LBMPC_NOALLOC_MAX_PARTIALS);
        return LBMPC_ERR_DIM;
    }
    return hmax_launch(WA, nW, nx, samples_dev, N, batch, batch_stride, m_out_dev, chunks, spc, ptr, (hipStream_t)stream);
}

extern "C" int lbmpc_halfspace_max(const double* WA, int nW, int nx, const double* samples, int64_t N,
                                   int64_t batch, int64_t batch_stride, double* m_out) {
    if (int rc = check_device()) return rc;
    if (!samples || !m_out || N < 1 || batch < 0 || nx < 1 || nW < 1) { set_error("lbmpc_halfspace_max: bad argument"); return LBMPC_ERR_ARG; }
    if (batch == 0) return 0;
    if (batch > 1 && batch_stride < N * nx) { set_error("lbmpc_halfspace_max: batch_stride < N*nx"); return LBMPC_ERR_ARG; }
    const size_t total = (size_t)((batch - 1) * batch_stride + N * nx);
    DevBuf ds, dm;
    LB_HIP(ds.alloc(total * sizeof(double)));
    LB_HIP(dm.alloc((size_t)batch * nW * sizeof(double)));
    LB_HIP(hipMemcpy(ds.p, samples, total * sizeof(double), hipMemcpyHostToDevice));
    int rc = lbmpc_halfspace_max_dev(WA, nW, nx, ds.as<double>(), N, batch, batch_stride, dm.as<double>(), nullptr);
    if (rc) return rc;
    LB_HIP(hipMemcpy(m_out, dm.p, (size_t)batch * nW * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

// ---- on-device information-set generation -------------------------------------------------------
namespace lbmpc {

// One wave per information set; lane l draws samples l, l+64, ...  (stream (seed, set0+b), index i).
// A sample is three draws -- u_LV (always accepted), xi_EV, xi_LV (rejection) -- and every lane walks ITS OWN sequence of
// (sample, draw, attempt): one loop trip is one Philox block, one proposal and one membership test for every lane,
// whatever draw the lane is at, so no lane waits for another lane's rejections (in lock step per draw a wave pays the
// maximum attempt count of its 64 lanes, 5-6 instead of 1.7 for the trapezoids of the 2x2 system).  The three draws
// share one code path: their boxes, bases and rows sit in LDS as three descriptors of one layout and the lane
// addresses the one of its draw (u_LV as a "polytope" without rows whose basis is B).  Same blocks, same arithmetic
// and same accepted values as sample_w_I0().
// lps lanes per information set (a power of two, 64 / lps sets per wave): 64 for large N; small information sets
// (the card(I0) = 10 ... 100 of the closed-loop Monte-Carlo, one set per trial) take fewer, so that the lanes of a
// wave are busy and one wave learns several sets.  NT: length of the proposal vector t (NX, or LBMPC_MAX_NU when the
// input draw has more components than the state: nu > nx).
template <int NX, int NT, bool SHARED /* several sets per wave (lps_arg < 64) */>
__global__ __launch_bounds__(256) void sample_sets_kernel(const SmpPhases* __restrict__ desc_g, int nW,
                                                          int64_t N, int64_t batch, uint64_t seed, uint64_t set0,
                                                          int max_attempts, int lps_arg, double* __restrict__ samples_out,
                                                          double* __restrict__ m_out, int* __restrict__ status_out) {
    const int lps = SHARED ? lps_arg : LB_WAVE;
    __shared__ SmpPhases D;
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(desc_g);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&D);
        for (int i = threadIdx.x; i < (int)(sizeof(SmpPhases) / 4); i += blockDim.x) dst[i] = src[i];
        __syncthreads();
    }
    const int lane = lane_id();
    const int spw = LB_WAVE / lps;   // sets per wave
    const int64_t wv = (int64_t)blockIdx.x * (blockDim.x / LB_WAVE) + (threadIdx.x / LB_WAVE);
    if (wv * spw >= batch) return;
    const int64_t b = SHARED ? wv * spw + lane / lps : wv;
    const uint64_t set = set0 + (uint64_t)b;
    const int nr4 = D.nr4;
    double m[LBMPC_MAX_NW];
#pragma unroll
    for (int j = 0; j < LBMPC_MAX_NW; ++j) m[j] = -LB_INF;
    bool cap = false;
    int64_t i = lane & (lps - 1);
    int ph = 0, att = 0;
    bool active = b < batch && i < N;
    double bu[NX], ev[NX];
#pragma unroll
    for (int k = 0; k < NX; ++k) { bu[k] = 0.0; ev[k] = 0.0; }
    while (__builtin_amdgcn_ballot_w64(active) != 0) {
        const SmpPhase* P = &D.ph[ph];
        uint32_t c[4] = {(uint32_t)set, (uint32_t)(set >> 32), (uint32_t)i, ((uint32_t)P->purpose << 28) | ((uint32_t)att << 4)};
        philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
        double t[NT], x[NX];
#pragma unroll
        for (int k = 0; k < NT; ++k) t[k] = P->lo[k] + (P->hi[k] - P->lo[k]) * u01(c[k]);
        const bool hasU = P->hasU != 0;
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < NT; ++q) acc = fma(P->U[k * LBMPC_MAX_NX + q], t[q], acc);   // columns >= nx (nu) of a basis are exact zeros
            x[k] = hasU ? acc : t[k];
        }
        bool ok = true;
        for (int r = 0; r < nr4; r += 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < NX; ++k) acc = fma(P->A[(r + j) * LBMPC_MAX_NX + k], x[k], acc);
                ok = ok && (acc <= P->b[r + j]);
            }
        }
        const bool last = att + 1 >= max_attempts;
        const bool take = active && (ok || last);   // at the attempt cap the last candidate is kept, as sample_polytope() does
        cap = cap || (active && !ok && last);
        if (take && ph == 0) {
#pragma unroll
            for (int k = 0; k < NX; ++k) bu[k] = x[k];
        }
        if (take && ph == 1) {
#pragma unroll
            for (int k = 0; k < NX; ++k) ev[k] = x[k];
        }
        if (take && ph == 2) {
            double w[NX];
#pragma unroll
            for (int k = 0; k < NX; ++k) w[k] = (ev[k] - x[k]) - bu[k];
            if (samples_out) {
#pragma unroll
                for (int k = 0; k < NX; ++k) samples_out[(b * N + i) * NX + k] = w[k];
            }
            int woff = 0;
            asm volatile("" : "+v"(woff));   // read here, not kept in registers for the whole loop
            const double* Wr = D.W + woff;
#pragma unroll
            for (int j = 0; j < LBMPC_MAX_NW; ++j)
                if (j < nW) {
                    double acc = 0.0;
#pragma unroll
                    for (int k = 0; k < NX; ++k) acc = fma(Wr[j * NX + k], w[k], acc);
                    m[j] = acc > m[j] ? acc : m[j];
                }
            i += lps;
            active = i < N;
        }
        att = take ? 0 : att + 1;
        ph = take ? (ph == 2 ? 0 : ph + 1) : ph;
    }
    if (!SHARED) {
        const bool anycap = __builtin_amdgcn_ballot_w64(cap) != 0;
#pragma unroll
        for (int j = 0; j < LBMPC_MAX_NW; ++j)
            if (j < nW) {
                double r = wave_max(m[j]);
                if (lane == 0 && m_out) m_out[b * nW + j] = r;
            }
        if (lane == 0 && status_out) status_out[b] = anycap ? LBMPC_ST_SAMPLER_CAP : 0;
    } else {   // several sets per wave: butterfly inside each group of lps lanes
        int capi = cap ? 1 : 0;
        for (int off = lps >> 1; off > 0; off >>= 1) capi |= __shfl_xor(capi, off);
#pragma unroll
        for (int j = 0; j < LBMPC_MAX_NW; ++j)
            if (j < nW) {
                double r = m[j];
                for (int off = lps >> 1; off > 0; off >>= 1) { const double o = __shfl_xor(r, off); r = o > r ? o : r; }
                if ((lane & (lps - 1)) == 0 && b < batch && m_out) m_out[b * nW + j] = r;
            }
        if ((lane & (lps - 1)) == 0 && b < batch && status_out) status_out[b] = capi ? LBMPC_ST_SAMPLER_CAP : 0;
    }
}

// host: the three draw descriptors of sample_sets_kernel from a packed sampler
static void pack_phases(const SamplerDev& h, SmpPhases* o) {
    *o = SmpPhases{};
    const int nx = h.nx, nu = h.nu;
    for (int p = 0; p < 3; ++p)
        for (int r = 0; r < LBMPC_MAX_NXI; ++r) o->ph[p].b[r] = INFINITY;
    SmpPhase& u = o->ph[0];
    for (int j = 0; j < nu; ++j) { u.lo[j] = h.min_u[j]; u.hi[j] = h.max_u[j]; }
    for (int k = 0; k < nx; ++k) for (int j = 0; j < nu; ++j) u.U[k * LBMPC_MAX_NX + j] = h.B[k * nu + j];
    u.hasU = 1; u.purpose = P_I0_U;
    auto fill = [&](SmpPhase& d, const double* A, const double* b, int nr, const double* lo, const double* hi, const double* U,
                    int hasU, int purpose) {
        for (int k = 0; k < nx; ++k) { d.lo[k] = lo[k]; d.hi[k] = hi[k]; }
        for (int k = 0; k < nx; ++k) for (int q = 0; q < nx; ++q) d.U[k * LBMPC_MAX_NX + q] = U[k * nx + q];
        for (int r = 0; r < nr; ++r) { for (int k = 0; k < nx; ++k) d.A[r * LBMPC_MAX_NX + k] = A[r * nx + k]; d.b[r] = b[r]; }
        d.hasU = hasU; d.purpose = purpose;
    };
    fill(o->ph[1], h.EV_A, h.EV_b, h.nEV, h.EV_lo, h.EV_hi, h.EV_U, h.EV_hasU, P_I0_EV);
    fill(o->ph[2], h.LV_A, h.LV_b, h.nLV, h.LV_lo, h.LV_hi, h.LV_U, h.LV_hasU, P_I0_LV);
    const int nr = h.nEV > h.nLV ? h.nEV : h.nLV;
    o->nr4 = (nr + 3) & ~3;
}

int pack_sampler(const lbmpc_sampler* s, SamplerDev* o) {
    if (!s || !s->EV_A || !s->EV_b || !s->EV_lo || !s->EV_hi || !s->LV_A || !s->LV_b || !s->LV_lo || !s->LV_hi ||
        !s->B || !s->min_u_LV || !s->max_u_LV) { set_error("lbmpc_sampler: null field"); return LBMPC_ERR_ARG; }
    if (s->nx < 1 || s->nu < 1 || s->nEV < 1 || s->nLV < 1) { set_error("lbmpc_sampler: bad size"); return LBMPC_ERR_ARG; }
    if (s->nx > LBMPC_MAX_NX || s->nu > LBMPC_MAX_NU || s->nEV > LBMPC_MAX_NXI || s->nLV > LBMPC_MAX_NXI) { set_error("lbmpc_sampler: dimension limit"); return LBMPC_ERR_DIM; }
    *o = SamplerDev{};
    // rows beyond nEV / nLV read 0.x <= inf: kernels that test rows four at a time need no tail
    for (int r = 0; r < LBMPC_MAX_NXI; ++r) { o->EV_b[r] = INFINITY; o->LV_b[r] = INFINITY; }
    o->nx = s->nx; o->nu = s->nu; o->nEV = s->nEV; o->nLV = s->nLV;
    // Rows that no proposal can violate are left out (they are most of the rows when the proposal box hugs the
    // polytope: all of them for a cuboid in its own axes).  Proposals are t = lo + (hi-lo) u with u = (w + 1/2) 2^-32,
    // so they stay half a granule inside the box: a row whose maximum over that inset box, plus a bound on the rounding
    // of the evaluated a.x, is <= b holds for every candidate, and dropping it changes no accept/reject decision
    // (the oracle evaluates all rows; streams stay bit-identical).
    auto keep_rows = [&](const double* A, const double* b, int nr, const double* lo, const double* hi, const double* U,
                         double* Ao, double* bo) -> int {
        const int nx = s->nx;
        int nk = 0;
        for (int r = 0; r < nr; ++r) {
            // mag bounds the magnitudes that enter the kernel's evaluation a.(U t): sum_k |a_k| sum_i |U_ki| max|t_i| (not
            // |g_i| = |a.U_i|, which cancellation can make tiny for a skewed basis while the rounding stays O(|a||U||t|))
            double hmax = 0.0, mag = std::fabs(b[r]);
            for (int i = 0; i < nx; ++i) {   // g_i = a . (column i of U): the row in proposal coordinates
                double g = 0.0, gabs = 0.0;
                for (int k = 0; k < nx; ++k) {
                    const double uki = U ? U[k + (size_t)nx * i] : (k == i ? 1.0 : 0.0);
                    g += A[r + (size_t)nr * k] * uki;
                    gabs += std::fabs(A[r + (size_t)nr * k]) * std::fabs(uki);
                }
                const double w = hi[i] - lo[i], inset = 0x1.0p-33 * w;
                hmax += g > 0.0 ? g * (hi[i] - inset) : g * (lo[i] + inset);
                mag += gabs * (std::fabs(lo[i]) > std::fabs(hi[i]) ? std::fabs(lo[i]) : std::fabs(hi[i]));
            }
            if (std::isfinite(hmax) && hmax + 1e-13 * mag <= b[r]) continue;   // never violated
            for (int k = 0; k < nx; ++k) Ao[nk * nx + k] = A[r + (size_t)nr * k];
            bo[nk] = b[r];
            ++nk;
        }
        return nk;
    };
    o->nEV = keep_rows(s->EV_A, s->EV_b, s->nEV, s->EV_lo, s->EV_hi, s->EV_U, o->EV_A, o->EV_b);
    o->nLV = keep_rows(s->LV_A, s->LV_b, s->nLV, s->LV_lo, s->LV_hi, s->LV_U, o->LV_A, o->LV_b);
    for (int k = 0; k < s->nx; ++k) { o->EV_lo[k] = s->EV_lo[k]; o->EV_hi[k] = s->EV_hi[k]; o->LV_lo[k] = s->LV_lo[k]; o->LV_hi[k] = s->LV_hi[k]; }
    for (int i = 0; i < s->nx; ++i) for (int j = 0; j < s->nu; ++j) o->B[i * s->nu + j] = s->B[i + (size_t)s->nx * j];
    for (int j = 0; j < s->nu; ++j) { o->min_u[j] = s->min_u_LV[j]; o->max_u[j] = s->max_u_LV[j]; }
    o->EV_hasU = s->EV_U != nullptr; o->LV_hasU = s->LV_U != nullptr;
    for (int k = 0; k < s->nx; ++k) for (int i = 0; i < s->nx; ++i) {
        o->EV_U[k * s->nx + i] = s->EV_U ? s->EV_U[k + (size_t)s->nx * i] : (k == i ? 1.0 : 0.0);
        o->LV_U[k * s->nx + i] = s->LV_U ? s->LV_U[k + (size_t)s->nx * i] : (k == i ? 1.0 : 0.0);
    }
    return 0;
}

}  // namespace lbmpc

extern "C" int lbmpc_sample_information_sets_dev(const lbmpc_sampler* smp, const double* WA, int nW, int64_t N,
                                                 int64_t batch, uint64_t seed, uint64_t set0, int max_attempts,
                                                 double* samples_out_dev, double* m_out_dev, int* status_out_dev,
                                                 void* stream) {
    if (int rc = check_device()) return rc;
    if (!WA || N < 1 || batch < 0 || nW < 1 || max_attempts < 1) { set_error("lbmpc_sample_information_sets: bad argument"); return LBMPC_ERR_ARG; }
    if (nW > LBMPC_MAX_NW) return LBMPC_ERR_DIM;
    if (N > 0xffffffffLL) { set_error("lbmpc_sample_information_sets: N exceeds the 32-bit sample index"); return LBMPC_ERR_DIM; }
    SamplerDev h;
    if (int rc = pack_sampler(smp, &h)) return rc;
    if (batch == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    RowsW W;
    pack_rows(WA, nW, h.nx, &W);
    SmpPhases hp;
    pack_phases(h, &hp);
    for (int i = 0; i < LBMPC_MAX_NW * LBMPC_MAX_NX; ++i) hp.W[i] = W.a[i];
    SmpPhases* d = nullptr;   // stream-ordered: the descriptors live until the kernel that reads them has run
    LB_HIP(hipMallocAsync((void**)&d, sizeof(SmpPhases), st));
    LB_HIP(hipMemcpyAsync(d, &hp, sizeof(SmpPhases), hipMemcpyHostToDevice, st));   // pageable source: staged before the call returns
    const int rc = launch_sample_sets(h.nx, h.nu, d, nW, N, batch, seed, set0, max_attempts, samples_out_dev, m_out_dev, status_out_dev, st);
    (void)hipFreeAsync(d, st);
    return rc;
}

namespace lbmpc {
// lanes per information set (a power of two; results do not depend on it: maxima are order-free, samples are keyed by
// (set, index)).  A lane walks its samples independently of the others, so the more samples a lane has the closer the
// wave's trip count (the slowest lane's) is to the mean and the fewer lanes idle in the remainder round: take as few
// lanes per set as still leave LBMPC_SS_WAVES (8 per SIMD) waves in the launch, at most the smallest power of two >= N.
// LBMPC_SS_LPS forces a value (diagnostic).
static int lanes_per_set(int64_t N, int64_t batch) {
    int by_n = 4;
    while (by_n < LB_WAVE && by_n < N) by_n <<= 1;
    static const int forced = [] { const char* e = getenv("LBMPC_SS_LPS"); return e ? atoi(e) : 0; }();
    if (forced >= 4 && forced <= LB_WAVE && (forced & (forced - 1)) == 0) return forced < by_n ? forced : by_n;
    const int64_t target_waves = 8192;
    int by_occ = 4;
    while (by_occ < LB_WAVE && batch * by_occ < target_waves * LB_WAVE) by_occ <<= 1;
    return by_occ < by_n ? by_occ : by_n;
}
int launch_sample_sets(int nx, int nu, const SmpPhases* d, int nW, int64_t N, int64_t batch, uint64_t seed, uint64_t set0,
                       int max_attempts, double* samples_out_dev, double* m_out_dev, int* status_out_dev, hipStream_t st) {
    const int lps = lanes_per_set(N, batch);
    const int64_t waves = (batch + (LB_WAVE / lps) - 1) / (LB_WAVE / lps);
    dim3 grid((unsigned)((waves + 3) / 4));
#define LAUNCH_SS(NXV, NTV)                                                                                                      \
    do {                                                                                                                         \
        if (lps < LB_WAVE) hipLaunchKernelGGL((sample_sets_kernel<NXV, NTV, true>), grid, dim3(256), 0, st, d, nW, N, batch, seed, set0, max_attempts, lps, samples_out_dev, m_out_dev, status_out_dev); \
        else hipLaunchKernelGGL((sample_sets_kernel<NXV, NTV, false>), grid, dim3(256), 0, st, d, nW, N, batch, seed, set0, max_attempts, lps, samples_out_dev, m_out_dev, status_out_dev); \
    } while (0)
    const bool wide = nu > nx;   // the input draw has more components than the state
    switch (nx) {
        case 1: if (wide) LAUNCH_SS(1, LBMPC_MAX_NU); else LAUNCH_SS(1, 1); break;
        case 2: if (wide) LAUNCH_SS(2, LBMPC_MAX_NU); else LAUNCH_SS(2, 2); break;
        case 3: if (wide) LAUNCH_SS(3, LBMPC_MAX_NU); else LAUNCH_SS(3, 3); break;
        default: LAUNCH_SS(4, 4); break;
    }
#undef LAUNCH_SS
    LB_HIP(hipGetLastError());
    return 0;
}
int pack_sample_phases(const SamplerDev& h, const double* W_rows /* nW x nx row-major */, int nW, SmpPhases* o) {
    pack_phases(h, o);
    for (int i = 0; i < LBMPC_MAX_NW * LBMPC_MAX_NX; ++i) o->W[i] = 0.0;
    for (int i = 0; i < nW * h.nx; ++i) o->W[i] = W_rows[i];
    return 0;
}
}  // namespace lbmpc

extern "C" int lbmpc_sample_information_sets(const lbmpc_sampler* smp, const double* WA, int nW, int64_t N,
                                             int64_t batch, uint64_t seed, uint64_t set0, int max_attempts,
                                             double* samples_out, double* m_out, int* status_out) {
    if (int rc = check_device()) return rc;
    if (!smp || batch < 0 || N < 1 || nW < 1) { set_error("lbmpc_sample_information_sets: bad argument"); return LBMPC_ERR_ARG; }
    if (batch == 0) return 0;
    const int nx = smp->nx;
    DevBuf ds, dm, dst;
    if (samples_out) LB_HIP(ds.alloc((size_t)batch * N * nx * sizeof(double)));
    LB_HIP(dm.alloc((size_t)batch * nW * sizeof(double)));
    LB_HIP(dst.alloc((size_t)batch * sizeof(int)));
    int rc = lbmpc_sample_information_sets_dev(smp, WA, nW, N, batch, seed, set0, max_attempts,
                                               samples_out ? ds.as<double>() : nullptr, dm.as<double>(), dst.as<int>(), nullptr);
    if (rc) return rc;
    LB_HIP(hipDeviceSynchronize());
    if (samples_out) LB_HIP(hipMemcpy(samples_out, ds.p, (size_t)batch * N * nx * sizeof(double), hipMemcpyDeviceToHost));
    if (m_out) LB_HIP(hipMemcpy(m_out, dm.p, (size_t)batch * nW * sizeof(double), hipMemcpyDeviceToHost));
    if (status_out) LB_HIP(hipMemcpy(status_out, dst.p, (size_t)batch * sizeof(int), hipMemcpyDeviceToHost));
    return 0;
}

// ---- stand-alone polytope sampler (ModelingCar.m:47-91) -------------------------------------------
namespace lbmpc {
struct PolyDev {
    int nr;
    double A[LBMPC_MAX_NXI * LBMPC_MAX_NX], b[LBMPC_MAX_NXI], lo[LBMPC_MAX_NX], hi[LBMPC_MAX_NX];
};
template <int NX>
__global__ __launch_bounds__(256) void sample_polytope_kernel(PolyDev P, int64_t n, uint64_t seed, uint64_t stream,
                                                              uint32_t index0, uint32_t purpose, int max_attempts,
                                                              double* __restrict__ out, int* __restrict__ status) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x[NX];
    const bool cap = sample_polytope<NX>(P.A, P.b, P.nr, P.lo, P.hi, nullptr, seed, stream, index0 + (uint32_t)i, purpose, max_attempts, x);
#pragma unroll
    for (int k = 0; k < NX; ++k) out[i * NX + k] = x[k];
    if (status) status[i] = cap ? LBMPC_ST_SAMPLER_CAP : 0;
}
}  // namespace lbmpc

extern "C" int lbmpc_sample_polytope(const double* XA, const double* Xb, int nr, int nx, const double* lo,
                                     const double* hi, int64_t n, uint64_t seed, uint64_t stream, uint32_t index0,
                                     int purpose, int max_attempts, double* out, int* status_out) {
    if (int rc = check_device()) return rc;
    if (!XA || !Xb || !lo || !hi || !out || n < 0 || nr < 1 || nx < 1 || purpose < 0 || purpose > 15 || max_attempts < 1) { set_error("lbmpc_sample_polytope: bad argument"); return LBMPC_ERR_ARG; }
    if (nx > LBMPC_MAX_NX || nr > LBMPC_MAX_NXI) { set_error("lbmpc_sample_polytope: nx<=%d rows<=%d", LBMPC_MAX_NX, LBMPC_MAX_NXI); return LBMPC_ERR_DIM; }
    if (n == 0) return 0;
    PolyDev P;
    P.nr = nr;
    for (int r = 0; r < nr; ++r) { for (int k = 0; k < nx; ++k) P.A[r * nx + k] = XA[r + (size_t)nr * k]; P.b[r] = Xb[r]; }
    for (int k = 0; k < nx; ++k) { P.lo[k] = lo[k]; P.hi[k] = hi[k]; }
    DevBuf dout, dst;
    LB_HIP(dout.alloc((size_t)n * nx * 8));
    LB_HIP(dst.alloc((size_t)n * 4));
    dim3 grid((unsigned)((n + 255) / 256));
    switch (nx) {
        case 1: hipLaunchKernelGGL(sample_polytope_kernel<1>, grid, dim3(256), 0, nullptr, P, n, seed, stream, index0, (uint32_t)purpose, max_attempts, dout.as<double>(), dst.as<int>()); break;
        case 2: hipLaunchKernelGGL(sample_polytope_kernel<2>, grid, dim3(256), 0, nullptr, P, n, seed, stream, index0, (uint32_t)purpose, max_attempts, dout.as<double>(), dst.as<int>()); break;
        case 3: hipLaunchKernelGGL(sample_polytope_kernel<3>, grid, dim3(256), 0, nullptr, P, n, seed, stream, index0, (uint32_t)purpose, max_attempts, dout.as<double>(), dst.as<int>()); break;
        default: hipLaunchKernelGGL(sample_polytope_kernel<4>, grid, dim3(256), 0, nullptr, P, n, seed, stream, index0, (uint32_t)purpose, max_attempts, dout.as<double>(), dst.as<int>()); break;
    }
    LB_HIP(hipGetLastError());
    LB_HIP(hipMemcpy(out, dout.p, (size_t)n * nx * 8, hipMemcpyDeviceToHost));
    if (status_out) LB_HIP(hipMemcpy(status_out, dst.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return 0;
}

// ---- homothety LP -------------------------------------------------------------------------------
namespace lbmpc {

int pack_box(const double* WA, const double* Wb, int nW, int nx, BoxW* o) {
    if (!WA || !Wb) { set_error("W: null pointer"); return LBMPC_ERR_ARG; }
    if (nx < 1 || nx > LBMPC_MAX_NX || nW > LBMPC_MAX_NW) return LBMPC_ERR_DIM;
    if (nW != 2 * nx) { set_error("prior W must be an axis-aligned box (nW == 2 nx); got nW=%d nx=%d", nW, nx); return LBMPC_ERR_UNSUPPORTED; }
    o->nx = nx; o->nW = nW;
    for (int r = 0; r < nW; ++r) o->perm[r] = -1;
    for (int j = 0; j < nW; ++j) {
        int ax = 0;
        for (int k = 1; k < nx; ++k) if (fabs(WA[j + (size_t)nW * k]) > fabs(WA[j + (size_t)nW * ax])) ax = k;
        const double a = WA[j + (size_t)nW * ax];
        if (a == 0.0) { set_error("W row %d is zero", j); return LBMPC_ERR_UNSUPPORTED; }
        for (int k = 0; k < nx; ++k)
            if (k != ax && fabs(WA[j + (size_t)nW * k]) > 1e-12 * fabs(a)) { set_error("W row %d is not axis-aligned", j); return LBMPC_ERR_UNSUPPORTED; }
        const int r = 2 * ax + (a > 0 ? 0 : 1);
        if (o->perm[r] >= 0) { set_error("W has two rows for the same face"); return LBMPC_ERR_UNSUPPORTED; }
        o->perm[r] = j;
        o->ic[r] = 1.0 / fabs(a);
    }
    o->exact = 1;
    for (int r = 0; r < nW; ++r) {
        if (o->perm[r] < 0) { set_error("W is unbounded along an axis"); return LBMPC_ERR_UNSUPPORTED; }
        for (int k = 0; k < nx; ++k) {
            o->WA[r * nx + k] = WA[o->perm[r] + (size_t)nW * k];
            if (k != r / 2 && o->WA[r * nx + k] != 0.0) o->exact = 0;
        }
        o->Wb[r] = Wb[o->perm[r]];
    }
    return 0;
}

template <int NX>
__global__ void homothety_lp_kernel(BoxW bw, const double* __restrict__ m, int64_t batch, double* __restrict__ alpha,
                                    double* __restrict__ v, uint32_t* __restrict__ mask_out, int* __restrict__ status) {
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    double mm[2 * NX], vv[NX], a;
    uint32_t mk;
#pragma unroll
    for (int r = 0; r < 2 * NX; ++r) mm[r] = m[b * 2 * NX + bw.perm[r]];
    int st = homothety_box<NX>(&bw, mm, a, vv, mk);
    alpha[b] = a;
#pragma unroll
    for (int k = 0; k < NX; ++k) v[b * NX + k] = vv[k];
    if (mask_out) mask_out[b] = mk;
    if (status) status[b] = st;
}

// ---- general polytope W: vertex enumeration, one lane per LP (homothety_gen.hpp) ----------------------------
__global__ __launch_bounds__(64) void homothety_lp_general_kernel(GenW W, const double* __restrict__ m_in, int64_t batch,
                                                                  double* __restrict__ alpha, double* __restrict__ v,
                                                                  uint32_t* __restrict__ mask_out, int* __restrict__ status) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const int nW = W.nW, nx = W.nx;
    double m[LBMPC_MAX_NW], a1, vv[LBMPC_MAX_NX];
    for (int j = 0; j < nW; ++j) m[j] = m_in[b * nW + j];
    uint32_t mask;
    const int st = homothety_general_dev(W, m, a1, vv, mask);
    alpha[b] = a1;
    for (int k = 0; k < nx; ++k) v[b * nx + k] = vv[k];
    if (mask_out) mask_out[b] = mask;
    if (status) status[b] = st;
}

}  // namespace lbmpc

extern "C" int lbmpc_homothety_lp_dev(const double* WA, const double* Wb, int nW, int nx, const double* m_dev,
                                      int64_t batch, int tiebreak, double* alpha_dev, double* v_dev,
                                      uint32_t* active_mask_dev, int* status_dev, void* stream) {
    if (int rc = check_device()) return rc;
    if (!m_dev || !alpha_dev || !v_dev || batch < 0) { set_error("lbmpc_homothety_lp: bad argument"); return LBMPC_ERR_ARG; }
    if (tiebreak != LBMPC_TIE_CENTER) { set_error("lbmpc_homothety_lp: unknown tiebreak %d", tiebreak); return LBMPC_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    BoxW bw;
    const int rcb = pack_box(WA, Wb, nW, nx, &bw);
    if (rcb == LBMPC_ERR_UNSUPPORTED) {   // not an axis-aligned box: vertex enumeration, one lane per LP
        if (nx < 1 || nx > LBMPC_MAX_NX || nW < nx + 1 || nW > LBMPC_MAX_NW) { set_error("lbmpc_homothety_lp: general W needs nx<=%d, nx+1<=nW<=%d", LBMPC_MAX_NX, LBMPC_MAX_NW); return LBMPC_ERR_DIM; }
        if (batch == 0) return 0;
        GenW gw;
        gw.nW = nW; gw.nx = nx;
        for (int j = 0; j < nW; ++j) { for (int k = 0; k < nx; ++k) gw.WA[j * nx + k] = WA[j + (size_t)nW * k]; gw.Wb[j] = Wb[j]; }
        hipLaunchKernelGGL(homothety_lp_general_kernel, dim3((unsigned)((batch + 63) / 64)), dim3(64), 0, st, gw, m_dev, batch, alpha_dev, v_dev, active_mask_dev, status_dev);
        LB_HIP(hipGetLastError());
        return 0;
    }
    if (rcb) return rcb;
    if (batch == 0) return 0;
    dim3 grid((unsigned)((batch + 255) / 256));
    switch (nx) {
        case 1: hipLaunchKernelGGL(homothety_lp_kernel<1>, grid, dim3(256), 0, st, bw, m_dev, batch, alpha_dev, v_dev, active_mask_dev, status_dev); break;
        case 2: hipLaunchKernelGGL(homothety_lp_kernel<2>, grid, dim3(256), 0, st, bw, m_dev, batch, alpha_dev, v_dev, active_mask_dev, status_dev); break;
        case 3: hipLaunchKernelGGL(homothety_lp_kernel<3>, grid, dim3(256), 0, st, bw, m_dev, batch, alpha_dev, v_dev, active_mask_dev, status_dev); break;
        default: hipLaunchKernelGGL(homothety_lp_kernel<4>, grid, dim3(256), 0, st, bw, m_dev, batch, alpha_dev, v_dev, active_mask_dev, status_dev); break;
    }
    LB_HIP(hipGetLastError());
    return 0;
}

extern "C" int lbmpc_homothety_lp(const double* WA, const double* Wb, int nW, int nx, const double* m,
                                  int64_t batch, int tiebreak, double* alpha, double* v, uint32_t* active_mask,
                                  int* status) {
    if (int rc = check_device()) return rc;
    if (!m || !alpha || !v || batch < 0 || nW < 1 || nx < 1) { set_error("lbmpc_homothety_lp: bad argument"); return LBMPC_ERR_ARG; }
    if (batch == 0) return 0;
    DevBuf dm, da, dv, dk, ds;
    LB_HIP(dm.alloc((size_t)batch * nW * sizeof(double)));
    LB_HIP(da.alloc((size_t)batch * sizeof(double)));
    LB_HIP(dv.alloc((size_t)batch * nx * sizeof(double)));
    LB_HIP(dk.alloc((size_t)batch * sizeof(uint32_t)));
    LB_HIP(ds.alloc((size_t)batch * sizeof(int)));
    LB_HIP(hipMemcpy(dm.p, m, (size_t)batch * nW * sizeof(double), hipMemcpyHostToDevice));
    int rc = lbmpc_homothety_lp_dev(WA, Wb, nW, nx, dm.as<double>(), batch, tiebreak, da.as<double>(), dv.as<double>(),
                                    dk.as<uint32_t>(), ds.as<int>(), nullptr);
    if (rc) return rc;
    LB_HIP(hipMemcpy(alpha, da.p, (size_t)batch * sizeof(double), hipMemcpyDeviceToHost));
    LB_HIP(hipMemcpy(v, dv.p, (size_t)batch * nx * sizeof(double), hipMemcpyDeviceToHost));
    if (active_mask) LB_HIP(hipMemcpy(active_mask, dk.p, (size_t)batch * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (status) LB_HIP(hipMemcpy(status, ds.p, (size_t)batch * sizeof(int), hipMemcpyDeviceToHost));
    return 0;
}
