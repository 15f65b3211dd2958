// runtime.hip -- library init / error plumbing of liblbmpc_hip.so
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "common.hpp"

namespace lbmpc {
static thread_local char g_err[512] = "";
static int g_ready = 0;

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_device() {
    if (g_ready) return 0;
    set_error("lbmpc_init() has not succeeded in this process (a gfx950 device is required; there is no CPU path)");
    return LBMPC_ERR_NODEVICE;
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
    if (device < 0 || device >= n) {
        lbmpc::set_error("lbmpc_init: device %d out of range [0,%d)", device, n);
        return LBMPC_ERR_ARG;
    }
    LB_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    LB_HIP(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        lbmpc::set_error("lbmpc_init: device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
        return LBMPC_ERR_NODEVICE;
    }
    lbmpc::g_ready = 1;
    return 0;
}
