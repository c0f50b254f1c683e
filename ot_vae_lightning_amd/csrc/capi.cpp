// Host-side utilities of the C ABI: error string, version, device query.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include "../../include/otvae.h"

static thread_local char g_err[512] = "";

void otvae_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* otvae_last_error(void) { return g_err; }
extern "C" int otvae_abi_version(void) { return 1; }

extern "C" int otvae_device_info(int* n_cu, int* wave_size, char* arch, int arch_len) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        otvae_set_error("otvae_device_info: no HIP device");
        return OTVAE_ELAUNCH;
    }
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) {
        otvae_set_error("otvae_device_info: hipGetDeviceProperties failed");
        return OTVAE_ELAUNCH;
    }
    if (n_cu) *n_cu = p.multiProcessorCount;
    if (wave_size) *wave_size = p.warpSize;
    if (arch && arch_len > 0) {
        strncpy(arch, p.gcnArchName, arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    return OTVAE_OK;
}
