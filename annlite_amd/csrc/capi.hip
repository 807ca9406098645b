// capi.hip -- library-level C ABI: version, error reporting, device queries.
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace annlite {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char *what) {
    set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    return ANNLITE_ERR_HIP;
}

int device_cu_count() {
    static thread_local int cached_dev = -1;
    static thread_local int cached_cu = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev != cached_dev) {
        int cu = 0;
        if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu <= 0) cu = 256;
        cached_cu = cu;
        cached_dev = dev;
    }
    return cached_cu;
}

}  // namespace annlite

using namespace annlite;

extern "C" int annlite_hip_abi_version(void) { return ANNLITE_HIP_ABI_VERSION; }

extern "C" const char *annlite_hip_last_error(void) { return g_err; }

extern "C" int annlite_hip_device_count(int *count) {
    ANNLITE_REQUIRE(count != nullptr, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        (void)hipGetLastError();
        return hip_fail(e, "hipGetDeviceCount");
    }
    *count = n;
    return ANNLITE_OK;
}

extern "C" int annlite_hip_device_arch(int dev, char *buf, size_t buf_len) {
    ANNLITE_REQUIRE(buf != nullptr && buf_len > 0, "buf is NULL");
    hipDeviceProp_t p;
    ANNLITE_HIP_TRY(hipGetDeviceProperties(&p, dev));
    strncpy(buf, p.gcnArchName, buf_len - 1);
    buf[buf_len - 1] = 0;
    return ANNLITE_OK;
}
