// support.hip -- subsystem 1: batched support functions for the mRPI set / tube tightening.
//
//   h_S(d) = (1-alpha)^-1 sum_{t<s} h_W(P_t d),  h_W(p) = max_v V_W[v,:].p,  P_t = (A^t)' or A^t
//
// Replaces MPT3 `W.support(...)` (one LP / vertex scan per column) at Functions_general/MRPISet.m:29,45
// and the nc support LPs per MPC step of UQRobustMPC.m:109-122 when S is the Minkowski-sum polytope.
// The s*nV stacked rows G_t = V_W P_t' are wave-uniform: they are read through scalar loads (they
// fit the scalar cache for every W of the reference: nV <= 16, s <= ~30) and feed v_fma_f64 as SGPR
// operands, so the sweep is bound by the fp64 vector rate, not by LDS or HBM (DESIGN.md "Kernel 1").
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "common.hpp"

namespace lbmpc {

// One lane = DPL directions (lane + 64 u), all s*nV rows applied to each.
template <int NX, int DPL, bool VEC>
__global__ __launch_bounds__(256) void support_kernel(const double* __restrict__ G /* s x nV x NX row-major */,
                                                      int s, int nV, double scale, const double* __restrict__ D,
                                                      int64_t nd, double* __restrict__ h_out) {
    const int64_t base = ((int64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63)) * DPL + (threadIdx.x & 63);
    double d[DPL][NX], sum[DPL];
#pragma unroll
    for (int u = 0; u < DPL; ++u) {
        const int64_t i = base + (int64_t)u * 64;
        const int64_t ii = i < nd ? i : (nd - 1);
        const double* p = D + ii * NX;
        if constexpr (VEC && (NX % 2 == 0)) {
#pragma unroll
            for (int q = 0; q < NX / 2; ++q) {
                double2 t = reinterpret_cast<const double2*>(p)[q];
                d[u][2 * q] = t.x; d[u][2 * q + 1] = t.y;
            }
        } else {
#pragma unroll
            for (int k = 0; k < NX; ++k) d[u][k] = p[k];
        }
        sum[u] = 0.0;
    }
    const double* __restrict__ g = G;
    if ((nV & 3) == 0) {
        // Rows in groups of four, double-buffered in SGPRs: the scalar loads of the next group are in flight while the
        // current one feeds the FMAs (scalar loads return out of order, so the only wait there is waits for all of
        // them: issued one group ahead, that wait falls after a group's worth of arithmetic).  nV % 4 == 0: a group
        // never straddles two t.
        const int gpt = nV >> 2, ngroups = s * gpt;
        double aA[4][NX], aB[4][NX], best[DPL];
#pragma unroll
        for (int u = 0; u < DPL; ++u) best[u] = -LB_INF;
        auto load4 = [&](double (*a)[NX], const double* src) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int k = 0; k < NX; ++k) a[r][k] = src[r * NX + k];
        };
        int gin = 0;   // group index inside the current t
        auto compute4 = [&](const double (*a)[NX]) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int u = 0; u < DPL; ++u) {
                    double acc = 0.0;
#pragma unroll
                    for (int k = 0; k < NX; ++k) acc = fma(a[r][k], d[u][k], acc);
                    best[u] = __builtin_fmax(acc, best[u]);
                }
            if (++gin == gpt) {
                gin = 0;
#pragma unroll
                for (int u = 0; u < DPL; ++u) { sum[u] += best[u]; best[u] = -LB_INF; }
            }
        };
        load4(aA, g);
        const int npairs = ngroups >> 1;
        for (int pi = 0; pi < npairs; ++pi) {   // branch-free body: the last pair loads group ngroups-1 (or itself) again
            asm volatile("" ::"s"(aA[0][0]));   // the wait for A comes before B's loads are issued
            load4(aB, g + 4 * NX);
            __builtin_amdgcn_sched_barrier(0);
            compute4(aA);
            asm volatile("" ::"s"(aB[0][0]));
            load4(aA, 2 * pi + 2 < ngroups ? g + 8 * NX : g);
            __builtin_amdgcn_sched_barrier(0);
            compute4(aB);
            g += 8 * NX;
        }
        if (ngroups & 1) compute4(aA);
    } else {
        for (int t = 0; t < s; ++t) {
            double best[DPL];
#pragma unroll
            for (int u = 0; u < DPL; ++u) best[u] = -LB_INF;
#pragma unroll 4  // four G rows per scalar-load wait
            for (int v = 0; v < nV; ++v, g += NX) {
                double a[NX];
#pragma unroll
                for (int k = 0; k < NX; ++k) a[k] = g[k];
#pragma unroll
                for (int u = 0; u < DPL; ++u) {
                    double acc = 0.0;
#pragma unroll
                    for (int k = 0; k < NX; ++k) acc = fma(a[k], d[u][k], acc);
                    best[u] = __builtin_fmax(acc, best[u]);   // one v_max_f64 (a compare + two selects otherwise); same value for non-NaN
                }
            }
#pragma unroll
            for (int u = 0; u < DPL; ++u) sum[u] += best[u];
        }
    }
#pragma unroll
    for (int u = 0; u < DPL; ++u) {
        const int64_t i = base + (int64_t)u * 64;
        if (i < nd) h_out[i] = sum[u] * scale;
    }
}

// host: G[t] = Vw P_t'  (nV x nx row-major per t), P_t = A^t (transpose_flag: (A^t)')
static void stack_G(const double* A, int nx, const double* Vw, int nV, int s, int transpose_flag, std::vector<double>& G) {
    G.assign((size_t)s * nV * nx, 0.0);
    double P[16], Q[16];
    for (int i = 0; i < nx; ++i) for (int j = 0; j < nx; ++j) P[i + nx * j] = (i == j) ? 1.0 : 0.0;  // column-major A^t
    for (int t = 0; t < s; ++t) {
        double* Gt = G.data() + (size_t)t * nV * nx;
        for (int v = 0; v < nV; ++v)
            for (int k = 0; k < nx; ++k) {
                double acc = 0.0;
                for (int j = 0; j < nx; ++j) {
                    const double pjk = transpose_flag ? P[k + nx * j] : P[j + nx * k];
                    acc = std::fma(Vw[v + (size_t)nV * j], pjk, acc);
                }
                Gt[v * nx + k] = acc;
            }
        for (int i = 0; i < nx; ++i)
            for (int j = 0; j < nx; ++j) {
                double acc = 0.0;
                for (int k = 0; k < nx; ++k) acc = std::fma(P[i + nx * k], A[k + nx * j], acc);
                Q[i + nx * j] = acc;
            }
        for (int i = 0; i < nx * nx; ++i) P[i] = Q[i];
    }
}

static int launch_support(const double* Gdev, int nx, int s, int nV, double scale, const double* D, int64_t nd,
                          double* out, hipStream_t st) {
    if (nd == 0) return 0;
    const bool vec = ((uintptr_t)D % 16) == 0;
    int dpl = nd >= (int64_t)256 * 8 * 256 * 4 ? 4 : 1;  // enough work for 4 directions per lane
    if (const char* e = getenv("LBMPC_SUPPORT_DPL")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4) dpl = v; }
#define LS(NXV, DPLV)                                                                                              \
    do {                                                                                                           \
        const int64_t per_block = 256 * (int64_t)DPLV;                                                             \
        dim3 grid((unsigned)((nd + per_block - 1) / per_block));                                                   \
        if (vec) hipLaunchKernelGGL((support_kernel<NXV, DPLV, true>), grid, dim3(256), 0, st, Gdev, s, nV, scale, D, nd, out); \
        else hipLaunchKernelGGL((support_kernel<NXV, DPLV, false>), grid, dim3(256), 0, st, Gdev, s, nV, scale, D, nd, out);    \
    } while (0)
    switch (nx * 8 + dpl) {
        case 9: LS(1, 1); break;   case 10: LS(1, 2); break;  case 12: LS(1, 4); break;
        case 17: LS(2, 1); break;  case 18: LS(2, 2); break;  case 20: LS(2, 4); break;
        case 25: LS(3, 1); break;  case 26: LS(3, 2); break;  case 28: LS(3, 4); break;
        case 33: LS(4, 1); break;  case 34: LS(4, 2); break;  default: LS(4, 4); break;
    }
#undef LS
    LB_HIP(hipGetLastError());
    return 0;
}

// Device copies of the stacked rows G, kept per (device, contents): repeated sweeps with the same (A, V_W, s) -- the
// usual pattern, e.g. a direction batch per solver call -- skip the allocation and the copy (which cost twice the
// kernel's time at 1M directions).  A small LRU per process (SUPPORT_CACHE entries over all devices), so that callers
// that alternate between a few parameter sets (3x3 / 4x4, transpose on/off, one host thread per device) do not thrash
// it.  A miss allocates (hipMalloc) and uploads stream-ordered from the entry's own host copy, which outlives the
// launch; only the eviction of the least recently used entry (more than SUPPORT_CACHE live parameter sets) waits for
// the device (hipFree).  Hits enqueue asynchronously and never synchronise.
struct RowsEntry {
    int device = -1;
    std::vector<double> host;
    void* dev = nullptr;
    hipEvent_t ready = nullptr;   // recorded after the upload: launches on other streams wait for it
    hipStream_t upload_stream = nullptr;
    uint64_t stamp = 0;
};
static const int SUPPORT_CACHE = 8;

static void drop_entry(RowsEntry& e) {
    if (!e.dev) return;
    DeviceGuard g(e.device);
    (void)hipFree(e.dev);   // waits for the launches that still read these rows
    if (e.ready) (void)hipEventDestroy(e.ready);
    e.dev = nullptr; e.ready = nullptr; e.host.clear();
}

static int cached_rows(const std::vector<double>& G, hipStream_t st, const double** out) {
    static std::mutex mu;
    static RowsEntry cache[SUPPORT_CACHE];
    static uint64_t clock_ = 0;
    std::lock_guard<std::mutex> lock(mu);
    int device = 0;
    LB_HIP(hipGetDevice(&device));
    for (RowsEntry& e : cache)
        if (e.dev && e.device == device && e.host == G) {
            e.stamp = ++clock_;
            if (st != e.upload_stream) LB_HIP(hipStreamWaitEvent(st, e.ready, 0));
            *out = (const double*)e.dev;
            return 0;
        }
    RowsEntry* slot = nullptr;
    for (RowsEntry& e : cache) if (!e.dev) { slot = &e; break; }
    if (!slot) {
        slot = &cache[0];
        for (RowsEntry& e : cache) if (e.stamp < slot->stamp) slot = &e;
        drop_entry(*slot);
    }
    RowsEntry& e = *slot;
    LB_HIP(hipMalloc(&e.dev, G.size() * 8));
    e.host = G; e.device = device; e.stamp = ++clock_; e.upload_stream = st;
    hipError_t err = hipEventCreateWithFlags(&e.ready, hipEventDisableTiming);
    if (err == hipSuccess) err = hipMemcpyAsync(e.dev, e.host.data(), G.size() * 8, hipMemcpyHostToDevice, st);
    if (err == hipSuccess) err = hipEventRecord(e.ready, st);
    if (err != hipSuccess) {
        drop_entry(e);
        set_error("support: upload of the stacked rows -> %s", hipGetErrorString(err));
        return LBMPC_ERR_HIP;
    }
    *out = (const double*)e.dev;
    return 0;
}

static int check_alpha(double alpha) {
    if (!(alpha >= 0.0 && alpha < 1.0)) { set_error("support: alpha = %g outside [0, 1) (the scale 1/(1-alpha) would be infinite or negative)", alpha); return LBMPC_ERR_ARG; }
    return 0;
}

static int check_support_args(const double* A, int nx, const double* Vw, int nV, int s) {
    if (!A || !Vw || nx < 1 || nV < 1 || s < 1) { set_error("support: bad argument"); return LBMPC_ERR_ARG; }
    if (nx > LBMPC_MAX_NX || nV > LBMPC_MAX_NV || s > LBMPC_MAX_S || (int64_t)s * nV > 2048) {
        set_error("support: limits nx<=%d nV<=%d s<=%d s*nV<=2048", LBMPC_MAX_NX, LBMPC_MAX_NV, LBMPC_MAX_S);
        return LBMPC_ERR_DIM;
    }
    return 0;
}

}  // namespace lbmpc

using namespace lbmpc;

extern "C" int lbmpc_support_batch_dev(const double* A, int nx, const double* Vw, int nV, int s, double alpha,
                                       const double* D_dev, int64_t nd, int transpose_flag, double* h_out_dev,
                                       void* stream) {
    if (int rc = check_device()) return rc;
    if (int rc = check_support_args(A, nx, Vw, nV, s)) return rc;
    if (int rc = check_alpha(alpha)) return rc;
    if (!D_dev || !h_out_dev || nd < 0) { set_error("lbmpc_support_batch: bad argument"); return LBMPC_ERR_ARG; }
    if (nd == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    std::vector<double> G;
    stack_G(A, nx, Vw, nV, s, transpose_flag, G);
    const double* Gdev = nullptr;
    if (int rc = cached_rows(G, st, &Gdev)) return rc;
    return launch_support(Gdev, nx, s, nV, 1.0 / (1.0 - alpha), D_dev, nd, h_out_dev, st);
}

extern "C" int lbmpc_support_batch(const double* A, int nx, const double* Vw, int nV, int s, double alpha,
                                   const double* D, int64_t nd, int transpose_flag, double* h_out) {
    if (int rc = check_device()) return rc;
    if (int rc = check_support_args(A, nx, Vw, nV, s)) return rc;
    if (int rc = check_alpha(alpha)) return rc;
    if (!D || !h_out || nd < 0) { set_error("lbmpc_support_batch: bad argument"); return LBMPC_ERR_ARG; }
    if (nd == 0) return 0;
    DevBuf dD, dh;
    LB_HIP(dD.alloc((size_t)nd * nx * 8));
    LB_HIP(dh.alloc((size_t)nd * 8));
    LB_HIP(hipMemcpy(dD.p, D, (size_t)nd * nx * 8, hipMemcpyHostToDevice));
    int rc = lbmpc_support_batch_dev(A, nx, Vw, nV, s, alpha, dD.as<double>(), nd, transpose_flag, dh.as<double>(), nullptr);
    if (rc) return rc;
    LB_HIP(hipMemcpy(h_out, dh.p, (size_t)nd * 8, hipMemcpyDeviceToHost));
    return 0;
}

// (s, alpha) search: MRPISet.m:27-63 (mode 0), Functions/MRPISet.m:8-17 (mode 1), Rakovic (mode 2).
// Every support value is evaluated on the device (one small launch per s for the nW + ndirs directions).
extern "C" int lbmpc_mrpi_s_alpha(const double* A, int nx, const double* Vw, int nV, const double* WA,
                                  const double* Wb, int nW, const double* Dirs, int ndirs, double eps, int mode,
                                  int max_s, int* s_out, double* alpha_out, double* Ms_out) {
    if (int rc = check_device()) return rc;
    if (!WA || !Wb || !Dirs || !s_out || !alpha_out || nW < 1 || ndirs < 1 || max_s < 1 || mode < 0 || mode > 2) { set_error("lbmpc_mrpi_s_alpha: bad argument"); return LBMPC_ERR_ARG; }
    if (max_s > LBMPC_MAX_S - 1) max_s = LBMPC_MAX_S - 1;
    if (int rc = check_support_args(A, nx, Vw, nV, 1)) return rc;
    if (nW > 65536) { set_error("lbmpc_mrpi_s_alpha: nW <= 65536"); return LBMPC_ERR_DIM; }   // the rows are only directions here: no fixed-size storage
    const int tr = (mode == LBMPC_MS_RAKOVIC);
    std::vector<double> G;
    stack_G(A, nx, Vw, nV, max_s + 1, tr, G);
    // direction batch: [f_0 .. f_{nW-1} | Dirs]
    const int nb = nW + ndirs;
    std::vector<double> Dh((size_t)nb * nx), hh(nb);
    for (int j = 0; j < nW; ++j) for (int k = 0; k < nx; ++k) Dh[(size_t)j * nx + k] = WA[j + (size_t)nW * k];
    for (int d = 0; d < ndirs; ++d) for (int k = 0; k < nx; ++k) Dh[(size_t)(nW + d) * nx + k] = Dirs[(size_t)d * nx + k];
    DevBuf dG, dD, dh;
    LB_HIP(dG.alloc(G.size() * 8)); LB_HIP(dD.alloc(Dh.size() * 8)); LB_HIP(dh.alloc((size_t)nb * 8));
    LB_HIP(hipMemcpy(dG.p, G.data(), G.size() * 8, hipMemcpyHostToDevice));
    LB_HIP(hipMemcpy(dD.p, Dh.data(), Dh.size() * 8, hipMemcpyHostToDevice));
    std::vector<double> sums(ndirs, 0.0);
    double alpha = 1000.0, Ms = 1000.0, Mmax = -INFINITY;
    int s = 0, rc = 0;
    while (alpha > eps / (eps + Ms)) {
        if (s >= max_s) { rc = LBMPC_ST_ITER_CAP; break; }
        s += 1;
        const int tnew = tr ? s - 1 : s;
        const double* Gs = dG.as<double>() + (size_t)s * nV * nx;
        const double* Gt = dG.as<double>() + (size_t)tnew * nV * nx;
        if (int e = launch_support(Gs, nx, 1, nV, 1.0, dD.as<double>(), nW, dh.as<double>(), nullptr)) return e;
        if (int e = launch_support(Gt, nx, 1, nV, 1.0, dD.as<double>() + (size_t)nW * nx, ndirs, dh.as<double>() + nW, nullptr)) return e;
        LB_HIP(hipMemcpy(hh.data(), dh.p, (size_t)nb * 8, hipMemcpyDeviceToHost));
        alpha = -INFINITY;
        for (int j = 0; j < nW; ++j) { double a = hh[j] / Wb[j]; if (a > alpha) alpha = a; }
        for (int d = 0; d < ndirs; ++d) { double h = hh[nW + d]; sums[d] += h; if (h > Mmax) Mmax = h; }
        if (mode == LBMPC_MS_MAX_LITERAL) Ms = Mmax;
        else { Ms = -INFINITY; for (int d = 0; d < ndirs; ++d) if (sums[d] > Ms) Ms = sums[d]; }
    }
    *s_out = s; *alpha_out = alpha;
    if (Ms_out) *Ms_out = Ms;
    return rc;
}
