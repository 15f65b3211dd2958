// multi.hip -- one node, several GPUs, inside ONE process behind the C ABI (the MATLAB drop-in is a single process:
// lbmpc_mex cannot use torch.distributed).  SURVEY.md 8e:
//   * Monte-Carlo trials: contiguous block shards by global trial id, one host thread + one context per device, no
//     exchange during the closed loop; every device copies its shard of the per-trial results into the caller's host
//     arrays.  The RNG stream is keyed by the global trial id, so the results are those of a single-device run, bitwise.
//   * one batch of information sets sharded BY SAMPLE: every device reduces its slice of every set to nW maxima, then
//     ONE all-reduce(max) of batch*nW doubles over RCCL (max is associative and commutative: bit-identical to one GPU).
// RCCL is loaded lazily (dlopen) so that the library carries no link-time dependency for single-GPU users; the
// communicators are created once per device list (ncclCommInitAll) and kept.
#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "common.hpp"

struct lbmpc_multi {
    std::vector<int> devices;
    std::vector<lbmpc_ctx*> ctx;
};

namespace lbmpc {

void block_shard(int64_t total, int rank, int world, int64_t* first, int64_t* count) {
    const int64_t q = total / world, r = total % world;
    *first = rank * q + std::min<int64_t>(rank, r);
    *count = q + (rank < r ? 1 : 0);
}

// ---- RCCL through dlopen -----------------------------------------------------------------------------------
typedef struct ncclComm* ncclComm_t;
struct Rccl {
    void* lib = nullptr;
    int (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
static const int NCCL_FLOAT64 = 8, NCCL_MAX = 2;   // ncclDataType_t::ncclFloat64, ncclRedOp_t::ncclMax (rccl.h)

static int load_rccl(Rccl** out) {
    static Rccl r;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (!r.lib) {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) { r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (r.lib) break; }
        if (!r.lib) { set_error("multi-GPU set learning needs RCCL: dlopen(librccl.so.1) failed: %s", dlerror()); return LBMPC_ERR_UNSUPPORTED; }
        r.CommInitAll = (int (*)(ncclComm_t*, int, const int*))dlsym(r.lib, "ncclCommInitAll");
        r.CommDestroy = (int (*)(ncclComm_t))dlsym(r.lib, "ncclCommDestroy");
        r.AllReduce = (int (*)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t))dlsym(r.lib, "ncclAllReduce");
        r.GroupStart = (int (*)())dlsym(r.lib, "ncclGroupStart");
        r.GroupEnd = (int (*)())dlsym(r.lib, "ncclGroupEnd");
        r.GetErrorString = (const char* (*)(int))dlsym(r.lib, "ncclGetErrorString");
        if (!r.CommInitAll || !r.CommDestroy || !r.AllReduce || !r.GroupStart || !r.GroupEnd || !r.GetErrorString) {
            set_error("librccl.so lacks an expected symbol");
            dlclose(r.lib); r.lib = nullptr;
            return LBMPC_ERR_UNSUPPORTED;
        }
    }
    *out = &r;
    return 0;
}

// LBMPC_MULTI_ALLOW_DUP=1 (testing on a box with fewer GPUs than shards): a device may be listed more than once, each
// entry still gets its own context, host thread and shard.  The RCCL exchange needs distinct devices: with duplicates
// lbmpc_halfspace_max_multi combines the per-shard maxima on the host instead.
static bool allow_dup() { const char* e = getenv("LBMPC_MULTI_ALLOW_DUP"); return e && atoi(e) == 1; }
static int check_devices(const int* devices, int ngpu, std::vector<int>* out, bool* has_dup = nullptr) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) { set_error("no HIP device visible"); return LBMPC_ERR_NODEVICE; }
    const bool dup_ok = allow_dup();
    if (ngpu < 1 || (ngpu > n && !(dup_ok && devices)) || ngpu > 64) { set_error("ngpu = %d outside [1, %d]", ngpu, n); return LBMPC_ERR_ARG; }
    out->clear();
    if (has_dup) *has_dup = false;
    for (int i = 0; i < ngpu; ++i) {
        const int d = devices ? devices[i] : i;
        if (d < 0 || d >= n) { set_error("device %d out of range [0,%d)", d, n); return LBMPC_ERR_ARG; }
        if (std::find(out->begin(), out->end(), d) != out->end()) {
            if (!dup_ok) { set_error("device %d listed twice", d); return LBMPC_ERR_ARG; }
            if (has_dup) *has_dup = true;
        }
        out->push_back(d);
    }
    return 0;
}

// run fn(i) for i in [0, n) on n host threads (inline when n == 1); returns the first non-zero code with its message
template <class F>
static int per_device(int n, F fn) {
    std::vector<int> rc(n, 0);
    std::vector<std::string> msg(n);
    auto body = [&](int i) { rc[i] = fn(i); if (rc[i]) msg[i] = lbmpc_last_error(); };
    if (n == 1) body(0);
    else {
        std::vector<std::thread> th;
        for (int i = 0; i < n; ++i) th.emplace_back(body, i);
        for (auto& t : th) t.join();
    }
    for (int i = 0; i < n; ++i)
        if (rc[i]) { set_error("device slot %d: %s", i, msg[i].c_str()); return rc[i]; }
    return 0;
}

}  // namespace lbmpc

using namespace lbmpc;

extern "C" int lbmpc_block_shard(int64_t total, int rank, int world, int64_t* first, int64_t* count) {
    if (total < 0 || world < 1 || rank < 0 || rank >= world || !first || !count) { set_error("lbmpc_block_shard: bad argument"); return LBMPC_ERR_ARG; }
    block_shard(total, rank, world, first, count);
    return 0;
}

extern "C" void lbmpc_multi_destroy(lbmpc_multi* m) {
    if (!m) return;
    for (lbmpc_ctx* c : m->ctx) lbmpc_destroy(c);
    delete m;
}

extern "C" int lbmpc_multi_create(const lbmpc_params* p, const int* devices, int ngpu, lbmpc_multi** out) {
    if (out) *out = nullptr;
    if (!p || !out) { set_error("lbmpc_multi_create: null argument"); return LBMPC_ERR_ARG; }
    std::vector<int> devs;
    if (int rc = check_devices(devices, ngpu, &devs)) return rc;
    lbmpc_multi* m = new lbmpc_multi();
    m->devices = devs;
    m->ctx.assign(ngpu, nullptr);
    int prev = -1;
    (void)hipGetDevice(&prev);
    int rc = 0;
    for (int i = 0; i < ngpu && !rc; ++i) {   // sequential: context creation is host work plus a few small copies
        rc = lbmpc_init(devs[i]);             // selects the device for this thread
        if (!rc) rc = lbmpc_uqmpc_create(p, &m->ctx[i]);
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    if (rc) { lbmpc_multi_destroy(m); return rc; }
    *out = m;
    return 0;
}

extern "C" int lbmpc_multi_ngpu(const lbmpc_multi* m) { return m ? (int)m->devices.size() : 0; }

extern "C" int lbmpc_multi_mc_run(lbmpc_multi* m, const lbmpc_sampler* smp, const lbmpc_mc_cfg* cfg, lbmpc_mc_out* out) {
    if (int rc = check_any_device()) return rc;
    if (!m || !smp || !cfg || !out || !cfg->x0) { set_error("lbmpc_multi_mc_run: null argument"); return LBMPC_ERR_ARG; }
    if (cfg->trials < 0 || cfg->steps < 0 || cfg->traj_trials < 0) { set_error("lbmpc_multi_mc_run: bad configuration"); return LBMPC_ERR_ARG; }
    const int G = (int)m->devices.size();
    const int nx = smp->nx, nu = smp->nu;
    const int64_t K = cfg->steps;
    return per_device(G, [&](int g) -> int {
        int64_t first, count;
        block_shard(cfg->trials, g, G, &first, &count);
        if (count == 0) return 0;
        lbmpc_mc_cfg c = *cfg;
        c.trials = count;
        c.trial0 = cfg->trial0 + first;
        const int64_t tr = std::min<int64_t>(std::max<int64_t>((int64_t)cfg->traj_trials - first, 0), count);
        c.traj_trials = (int32_t)tr;
        lbmpc_mc_out o = *out;   // the shard's slice of every caller array
        if (o.infeasible) o.infeasible += first;
        if (o.status) o.status += first;
        if (o.qp_iters) o.qp_iters += first;
        if (o.alpha_final) o.alpha_final += first;
        if (o.v_final) o.v_final += first * nx;
        if (o.alpha0) o.alpha0 += first;
        if (o.x_final) o.x_final += first * nx;
        if (o.traj_x) o.traj_x += first * (K + 1) * nx;
        if (o.traj_u) o.traj_u += first * K * nu;
        if (o.traj_alpha) o.traj_alpha += first * (K + 1);
        if (o.traj_s0) o.traj_s0 += first * K * nx;
        return lbmpc_mc_run(m->ctx[g], smp, &c, &o);   // runs on the context's device (DeviceGuard)
    });
}

// Information sets sharded by SAMPLE over the devices of `m` (or of a bare device list when m is NULL is not offered:
// a device list is all this needs, so it takes one directly).
extern "C" int lbmpc_halfspace_max_multi(const double* WA, int nW, int nx, const double* samples, int64_t N, int64_t batch,
                                         int64_t batch_stride, double* m_out, const int* devices, int ngpu) {
    if (!WA || !samples || !m_out || N < 1 || batch < 0 || nW < 1 || nx < 1) { set_error("lbmpc_halfspace_max_multi: bad argument"); return LBMPC_ERR_ARG; }
    if (nx > LBMPC_MAX_NX || nW > LBMPC_MAX_NW) { set_error("lbmpc_halfspace_max_multi: nx<=%d, nW<=%d", LBMPC_MAX_NX, LBMPC_MAX_NW); return LBMPC_ERR_DIM; }
    if (batch > 1 && batch_stride < N * nx) { set_error("lbmpc_halfspace_max_multi: batch_stride < N*nx"); return LBMPC_ERR_ARG; }
    if (batch == 0) return 0;
    std::vector<int> devs;
    bool has_dup = false;
    if (int rc = check_devices(devices, ngpu, &devs, &has_dup)) return rc;
    const int G = std::min<int64_t>(ngpu, N) < 1 ? 1 : (int)std::min<int64_t>(ngpu, N);   // never an empty slice
    devs.resize(G);
    Rccl* R = nullptr;
    if (!has_dup) if (int rc = load_rccl(&R)) return rc;
    // communicators: one set per device list, created on first use and kept until the process ends
    static std::mutex mu;
    static std::map<std::vector<int>, std::vector<ncclComm_t>> comm_cache;
    std::lock_guard<std::mutex> lock(mu);
    int prev = -1;
    (void)hipGetDevice(&prev);
    struct Restore { int d; ~Restore() { if (d >= 0) (void)hipSetDevice(d); } } restore{prev};
    for (int d : devs) if (int rc = lbmpc_init(d)) return rc;
    std::vector<ncclComm_t>* comms_p = nullptr;
    if (!has_dup) {
        auto it = comm_cache.find(devs);
        if (it == comm_cache.end()) {
            std::vector<ncclComm_t> fresh(G, nullptr);
            const int e = R->CommInitAll(fresh.data(), G, devs.data());
            if (e) { set_error("ncclCommInitAll -> %s", R->GetErrorString(e)); return LBMPC_ERR_HIP; }
            it = comm_cache.emplace(devs, std::move(fresh)).first;
        }
        comms_p = &it->second;
    }
    std::vector<void*> dS(G, nullptr), dM(G, nullptr);
    std::vector<hipStream_t> st(G, nullptr);
    auto cleanup = [&]() {
        for (int g = 0; g < G; ++g) {
            (void)hipSetDevice(devs[g]);
            if (dS[g]) (void)hipFree(dS[g]);
            if (dM[g]) (void)hipFree(dM[g]);
            if (st[g]) (void)hipStreamDestroy(st[g]);
        }
    };
    // local reductions, one host thread per device
    int rc = per_device(G, [&](int g) -> int {
        int64_t first, count;
        block_shard(N, g, G, &first, &count);
        LB_HIP(hipSetDevice(devs[g]));
        LB_HIP(hipStreamCreate(&st[g]));
        LB_HIP(hipMalloc(&dS[g], (size_t)batch * count * nx * 8));
        LB_HIP(hipMalloc(&dM[g], (size_t)batch * nW * 8));
        // slice [first, first+count) of every set: `batch` runs of count*nx doubles, pitch batch_stride
        LB_HIP(hipMemcpy2DAsync(dS[g], (size_t)count * nx * 8, samples + first * nx, (size_t)(batch > 1 ? batch_stride : count * nx) * 8,
                                (size_t)count * nx * 8, (size_t)batch, hipMemcpyHostToDevice, st[g]));
        return lbmpc_halfspace_max_dev(WA, nW, nx, (const double*)dS[g], count, batch, count * nx, (double*)dM[g], st[g]);
    });
    if (!rc && !has_dup) {   // the one exchange of this path
        std::vector<ncclComm_t>& comms = *comms_p;
        int e = R->GroupStart();
        for (int g = 0; g < G && !e; ++g) e = R->AllReduce(dM[g], dM[g], (size_t)batch * nW, NCCL_FLOAT64, NCCL_MAX, comms[g], st[g]);
        const int e2 = R->GroupEnd();
        if (e || e2) { set_error("ncclAllReduce(max) -> %s", R->GetErrorString(e ? e : e2)); rc = LBMPC_ERR_HIP; }
    }
    if (!rc && has_dup) {   // LBMPC_MULTI_ALLOW_DUP: shards that share a device -- the maxima meet on the host
        std::vector<double> acc((size_t)batch * nW, -INFINITY), part((size_t)batch * nW);
        for (int g = 0; g < G && !rc; ++g) {
            if (hipSetDevice(devs[g]) != hipSuccess || hipStreamSynchronize(st[g]) != hipSuccess ||
                hipMemcpy(part.data(), dM[g], part.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) { set_error("copy of shard %d failed", g); rc = LBMPC_ERR_HIP; break; }
            for (size_t i = 0; i < acc.size(); ++i) acc[i] = part[i] > acc[i] ? part[i] : acc[i];
        }
        if (!rc) std::memcpy(m_out, acc.data(), acc.size() * 8);
        cleanup();
        return rc;
    }
    if (!rc) {
        for (int g = 0; g < G && !rc; ++g) {
            if (hipSetDevice(devs[g]) != hipSuccess || hipStreamSynchronize(st[g]) != hipSuccess) { set_error("stream synchronise failed on device %d", devs[g]); rc = LBMPC_ERR_HIP; }
        }
    }
    if (!rc) {
        (void)hipSetDevice(devs[0]);
        if (hipMemcpy(m_out, dM[0], (size_t)batch * nW * 8, hipMemcpyDeviceToHost) != hipSuccess) { set_error("copy of the maxima failed"); rc = LBMPC_ERR_HIP; }
    }
    cleanup();
    return rc;
}
