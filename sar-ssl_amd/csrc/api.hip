// Error plumbing + version/probe entry points of the C-ABI library.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void sarssl_set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
extern "C" const char* sarssl_last_error() { return g_err; }
extern "C" int sarssl_abi_version() { return 1; }

// Device probe: returns 0 and fills name/arch info when a gfx950 device is usable.
extern "C" int sarssl_device_info(int device, char* name_out, int name_len, int* cu_count, long* lds_bytes) {
    hipDeviceProp_t p;
    hipError_t e = hipGetDeviceProperties(&p, device);
    if (e != hipSuccess) { sarssl_set_error("hipGetDeviceProperties: %s", hipGetErrorString(e)); return -2; }
    snprintf(name_out, name_len, "%s|%s", p.name, p.gcnArchName);
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (lds_bytes) *lds_bytes = (long)p.sharedMemPerBlock;
    return 0;
}

// Compute-unit count of the current device (cached per device): persistent kernels launch one workgroup per CU.
int sarssl_cu_count() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev] = n;
    }
    return cached[dev];
}
