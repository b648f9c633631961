#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace abr {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
}  // namespace abr

extern "C" const char* abr_last_error(void) { return abr::g_err; }
extern "C" int abr_version(void) { return 100; }
extern "C" int abr_device_info(int32_t* out) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) {
        abr::set_error("abr_device_info: no HIP device");
        return ABR_E_LAUNCH;
    }
    out[0] = p.multiProcessorCount;
    out[1] = (int32_t)p.maxSharedMemoryPerMultiProcessor;
    out[2] = p.warpSize;
    return ABR_OK;
}
