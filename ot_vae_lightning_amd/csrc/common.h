// Shared device/host helpers for the gfx950 kernels.  CDNA4 only: wave = 64 lanes, MFMA f32 16x16x4.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/otvae.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

#define OTVAE_WAVE 64

void otvae_set_error(const char* fmt, ...);

#define OTVAE_REQUIRE(cond, ...)                         \
    do {                                                 \
        if (!(cond)) {                                   \
            otvae_set_error(__VA_ARGS__);                \
            return OTVAE_EINVAL;                         \
        }                                                \
    } while (0)

#define OTVAE_CHECK_LAUNCH(name)                                                    \
    do {                                                                            \
        hipError_t e__ = hipGetLastError();                                         \
        if (e__ != hipSuccess) {                                                    \
            otvae_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return OTVAE_ELAUNCH;                                                   \
        }                                                                           \
    } while (0)

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

// ---- wave-level reductions (64 lanes, DPP/ds_swizzle via __shfl_xor) ---------------------------------------
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        T w = __shfl_xor(v, o, 64);
        v = w > v ? w : v;
    }
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_min(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        T w = __shfl_xor(v, o, 64);
        v = w < v ? w : v;
    }
    return v;
}

// D = A(16x4) * B(4x16) + C, fp32 exact.  lane l supplies A[l&15][l>>4] and B[l>>4][l&15];
// D element r of lane l is D[(l>>4)*4 + r][l&15]   (cdna_hip_programming.md section 3).
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
