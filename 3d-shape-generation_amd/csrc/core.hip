// Error reporting, version and device probe for the C ABI (include/pcd_hip.h).
#include <stdarg.h>
#include <string.h>
#include "common.h"

namespace pcd {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace pcd

extern "C" const char* pcd_last_error(void) { return pcd::g_err; }

extern "C" int pcd_abi_version(void) { return PCD_ABI_VERSION; }

extern "C" int pcd_device_check(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        pcd::set_error("no HIP device visible (%s)", e == hipSuccess ? "count 0" : hipGetErrorString(e));
        return PCD_ERR_HIP;
    }
    int dev = 0;
    PCD_CHECK_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    PCD_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        pcd::set_error("device %d is %s; this library is built for gfx950 (MI355X) only", dev, prop.gcnArchName);
        return PCD_ERR_HIP;
    }
    return PCD_OK;
}
