// runtime.hip -- library init / error plumbing of liblbmpc_hip.so
//
// Device model: readiness is recorded PER DEVICE (lbmpc_init validates one device at a time); entry points without
// a context run on the calling thread's current device and require that device to have been validated; entry points
// with a context switch to the context's device for their duration (DeviceGuard) and restore the caller's afterwards.
// The library never selects device 0 behind the caller's back (one process per GPU under torch.distributed, or one
// host thread per device inside lbmpc_mc_run_multi, both rely on that).
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "common.hpp"

namespace lbmpc {
static thread_local char g_err[512] = "";
static std::atomic<uint64_t> g_ready_mask{0};  // bit d: device d was validated by lbmpc_init (d < 64)

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int device_ready(int device) { return device >= 0 && device < 64 && ((g_ready_mask.load() >> device) & 1u); }

int check_any_device() {
    if (g_ready_mask.load() != 0) return 0;
    set_error("lbmpc_init() has not succeeded in this process (a gfx950 device is required; there is no CPU path)");
    return LBMPC_ERR_NODEVICE;
}

int check_device() {
    if (g_ready_mask.load() == 0) {
        set_error("lbmpc_init() has not succeeded in this process (a gfx950 device is required; there is no CPU path)");
        return LBMPC_ERR_NODEVICE;
    }
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || !device_ready(dev)) {
        set_error("the current device (%d) was not validated by lbmpc_init() (a gfx950 device is required; there is no CPU path)", dev);
        return LBMPC_ERR_NODEVICE;
    }
    return 0;
}

DeviceGuard::DeviceGuard(int device) {
    if (hipGetDevice(&prev_) != hipSuccess) { prev_ = -1; rc_ = LBMPC_ERR_HIP; set_error("hipGetDevice failed"); return; }
    if (prev_ != device) {
        const hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) { rc_ = LBMPC_ERR_HIP; set_error("hipSetDevice(%d) -> %s", device, hipGetErrorString(e)); prev_ = -1; return; }
        switched_ = true;
    }
}
DeviceGuard::~DeviceGuard() {
    if (switched_ && prev_ >= 0) (void)hipSetDevice(prev_);
}
}  // namespace lbmpc

extern "C" int lbmpc_version(void) { return LBMPC_VERSION; }

extern "C" const char* lbmpc_last_error(void) { return lbmpc::g_err; }

extern "C" int lbmpc_init(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) {
        lbmpc::set_error("lbmpc_init: no HIP device visible");
        return LBMPC_ERR_NODEVICE;
    }
    if (device == LBMPC_DEVICE_CURRENT) {
        LB_HIP(hipGetDevice(&device));   // adopt the calling thread's device, leave the selection alone
    } else {
        if (device < 0 || device >= n || device >= 64) {
            lbmpc::set_error("lbmpc_init: device %d out of range [0,%d)", device, n < 64 ? n : 64);
            return LBMPC_ERR_ARG;
        }
        LB_HIP(hipSetDevice(device));
    }
    if (lbmpc::device_ready(device)) return 0;
    hipDeviceProp_t prop;
    LB_HIP(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        lbmpc::set_error("lbmpc_init: device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
        return LBMPC_ERR_NODEVICE;
    }
    if (int rc = lbmpc::resolve_noalloc_scratch(device)) return rc;   // per-device address of the static scratch array
    lbmpc::g_ready_mask.fetch_or((uint64_t)1 << device);
    return 0;
}
