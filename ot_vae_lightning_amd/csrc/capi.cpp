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

// A HIP stream of the library's own on the calling thread's current device (non-blocking with respect to the null stream).  Why the
// host side does not take its side streams from its framework's stream pool: such pools are small and hand their streams out round-robin
// (torch: 32 per device), so the 33rd "new" stream IS the first one again -- two lanes of one captured step on the same queue.  Streams
// created here are never destroyed by the library's host mirror (it recycles them); otvae_stream_destroy is for other hosts.
extern "C" int otvae_stream_create(void** stream) {
    if (!stream) {
        otvae_set_error("otvae_stream_create: NULL argument");
        return OTVAE_EINVAL;
    }
    hipStream_t s = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e != hipSuccess) {
        otvae_set_error("otvae_stream_create: %s", hipGetErrorString(e));
        return OTVAE_ELAUNCH;
    }
    *stream = (void*)s;
    return OTVAE_OK;
}

extern "C" int otvae_stream_destroy(void* stream) {
    if (!stream) return OTVAE_OK;
    hipError_t e = hipStreamDestroy((hipStream_t)stream);
    if (e != hipSuccess) {
        otvae_set_error("otvae_stream_destroy: %s", hipGetErrorString(e));
        return OTVAE_ELAUNCH;
    }
    return OTVAE_OK;
}
