// wr_core.hip — error plumbing, version and device sanity for libwhisprrec_hip.so.
#include <stdarg.h>
#include <string.h>

#include "wr_common.h"

namespace wr {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int32_t fail_hip(hipError_t e, const char *what) {
    set_error("%s failed: %s (%d)", what, hipGetErrorString(e), (int)e);
    return (int32_t)e > 0 ? (int32_t)e : 1;
}

}  // namespace wr

extern "C" {

int32_t wr_abi_version(void) { return WR_ABI_VERSION; }

const char *wr_last_error(void) { return wr::g_err; }

int32_t wr_device_info(int32_t *n_cu, int32_t *wave_size, char *arch, int32_t arch_len) {
    int dev = 0;
    WR_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    WR_HIP(hipGetDeviceProperties(&prop, dev));
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (wave_size) *wave_size = prop.warpSize;
    if (arch && arch_len > 0) {
        strncpy(arch, prop.gcnArchName, (size_t)arch_len - 1);
        arch[arch_len - 1] = '\0';
    }
    return WR_OK;
}

}  // extern "C"
