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

// ---- BatchNorm statistic slots: cross-block sums WITHOUT a finalize launch, bit-reproducible (round 4) -------------------------------
// A producer block adds its per-channel sums (sum x, sum x^2) into one of S <= BN_SLOTS_MAX accumulators as two int64 fixed-point limbs
// (hi: units of 2^-10, lo: the exact remainder in units of 2^-53, < 2^43) with non-returning agent-scope INTEGER atomics.  Integer
// addition is associative: the totals do not depend on the order in which blocks arrive (fp64 atomics would), and 2048 blocks x 2^43 stay
// below 2^63.  A value that is not finite (or beyond 2^51) cannot be represented: it raises the tensor's poison counter instead and the
// consumer reads the statistics as NaN -- the propagation the plain fp64 partials had, which the device-side step guard relies on.
// Layout: long long [S][2 statistics][ld channels][2 limbs] + 2 words {poison counter, unused}; zero before the producer runs.
// S (a power of two <= 64: the caller's choice, the same for the producer and the consumer of a buffer) sets how many blocks add into one
// address: atomics on ONE address from all over the chip are performed one after the other, ~0.1 us each -- 64 of them were a 6 us tail
// on the producing kernel (the first version of this round used S = 16 for 1024 blocks: profiles/r04_bn_slots_ab.txt) -- so S grows
// with the producer's block count, and the consumer spreads the S x 2 x C loads over its threads (bn_fold_prologue).
#define BN_SLOTS_MAX 64

__host__ __device__ inline size_t bn_slot_words(int ld, int S) { return (size_t)S * 2 * ld * 2 + 2; }
static inline bool bn_slots_ok(int nslots) { return nslots >= 1 && nslots <= BN_SLOTS_MAX && (nslots & (nslots - 1)) == 0; }

__device__ __forceinline__ void bn_slot_add(long long* __restrict__ slots, int ld, unsigned S, unsigned slot, int stat, int c, double v) {
    if (!(fabs(v) < 2251799813685248.0)) {  // 2^51; also catches NaN
        __hip_atomic_fetch_add(slots + (size_t)S * 2 * ld * 2, 1LL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const double h = floor(v * 1024.0);
    const double r = v - h * (1.0 / 1024.0);  // exact: 0 <= r < 2^-10
    long long* dst = slots + ((((size_t)slot * 2 + stat) * ld + c) << 1);
    __hip_atomic_fetch_add(dst, (long long)h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(dst + 1, (long long)floor(r * 9007199254740992.0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

typedef long long bn_ll2 __attribute__((ext_vector_type(2)));

// limb sums of both statistics of channel c over slots s0 .. s0 + N - 1, all 2 N 16-byte loads in flight before the first addition
template <int N>
__device__ __forceinline__ void bn_slot_sum_n(const long long* __restrict__ slots, int ld, int c, int s0, bn_ll2& sa, bn_ll2& sb) {
    bn_ll2 a[N], b[N];
#pragma unroll
    for (int s = 0; s < N; ++s) {
        a[s] = *reinterpret_cast<const bn_ll2*>(slots + ((((size_t)(s0 + s) * 2 + 0) * ld + c) << 1));
        b[s] = *reinterpret_cast<const bn_ll2*>(slots + ((((size_t)(s0 + s) * 2 + 1) * ld + c) << 1));
    }
#pragma unroll
    for (int s = 0; s < N; ++s) {
        sa += a[s];
        sb += b[s];
    }
}

__device__ __forceinline__ double bn_limbs_to_double(long long hi, long long lo) {
    return (double)hi * (1.0 / 1024.0) + (double)lo * (1.0 / 9007199254740992.0);
}

// the totals of both statistics of channel c over the S slots in use, by ONE thread (NaN when the producer met a value it could not
// represent): the form for wide layers (few producer blocks, small S) and for the stand-alone finalize kernels
__device__ __forceinline__ void bn_slot_totals(const long long* __restrict__ slots, int ld, int S, int c, double& t0, double& t1) {
    bn_ll2 sa = {0, 0}, sb = {0, 0};
    switch (S) {
        case 1: bn_slot_sum_n<1>(slots, ld, c, 0, sa, sb); break;
        case 2: bn_slot_sum_n<2>(slots, ld, c, 0, sa, sb); break;
        case 4: bn_slot_sum_n<4>(slots, ld, c, 0, sa, sb); break;
        case 8: bn_slot_sum_n<8>(slots, ld, c, 0, sa, sb); break;
        default:
            for (int s0 = 0; s0 < S; s0 += 16) bn_slot_sum_n<16>(slots, ld, c, s0, sa, sb);
            break;
    }
    if (slots[(size_t)S * 2 * ld * 2] != 0) {
        t0 = t1 = __longlong_as_double(0x7ff8000000000000LL);
        return;
    }
    t0 = bn_limbs_to_double(sa.x, sa.y);
    t1 = bn_limbs_to_double(sb.x, sb.y);
}

// Everything the finalize launch took, for ONE branch (ConvLayer) that normalises a tensor whose statistics sit in slots: the consumer
// kernel's prologue turns them into (scale, shift) in LDS -- every block for itself, S x 4 words per channel -- and its first block
// also leaves mean / invstd / scale / shift in global memory for the backward pass and advances the running buffers
// (nn.BatchNorm2d, training mode: momentum, unbiased running variance).
struct BnFold {
    const long long* slots;  // NULL: no fold (scale / shift come from global arrays as before)
    int ld;
    int nslots;              // S
    long long count;         // elements per channel (N * H * W of the normalised tensor)
    float eps, momentum;
    const float* gamma;
    const float* beta;
    float* rmean;            // nullable (with rvar, nbt): running statistics of THIS branch's BatchNorm
    float* rvar;
    long long* nbt;
    float* mean_out;         // nullable: written when this branch is the one that publishes the shared mean / invstd
    float* invstd_out;
    float* scale_out;
    float* shift_out;
};

// s_sc / s_sh: LDS, C floats each.  Ends with a block barrier.
// C <= BN_FOLD_WIDE channels: the S x 2 x C slot reads are spread over the block's threads (consecutive threads, consecutive channels:
// coalesced 16-byte loads, at most a handful per thread, all in flight together) and summed with LDS integer atomics (associative:
// same bits in any order); wider layers have few producer blocks, hence small S, and one thread per channel reads its slots itself.
#define BN_FOLD_WIDE 64

// limb totals of both statistics of C <= BN_FOLD_WIDE channels over the S slots in use -> acc[(stat * C + c) * 2 + limb] (LDS, 4 C
// words), the reads spread over the block's threads.  Block-uniform call; ends with a barrier.
__device__ __forceinline__ void bn_slot_block_totals(const long long* __restrict__ slots, int ld, int S, int C,
                                                     unsigned long long* __restrict__ acc) {
    for (int i = threadIdx.x; i < 4 * C; i += blockDim.x) acc[i] = 0ULL;
    __syncthreads();
    const int W = 2 * C * S;
    for (int w0 = threadIdx.x; w0 < W; w0 += 4 * blockDim.x) {
        bn_ll2 v[4];
        int at[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int w = w0 + u * blockDim.x;
            at[u] = -1;
            if (w < W) {
                const int c = w % C, r = w / C;   // r = s * 2 + stat
                v[u] = *reinterpret_cast<const bn_ll2*>(slots + (((size_t)r * ld + c) << 1));
                at[u] = ((r & 1) * C + c) * 2;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (at[u] >= 0) {
                atomicAdd(&acc[at[u]], (unsigned long long)v[u].x);
                atomicAdd(&acc[at[u] + 1], (unsigned long long)v[u].y);
            }
    }
    __syncthreads();
}

// both statistics of channel c from what bn_slot_block_totals left (NaN when the tensor's poison word is set)
__device__ __forceinline__ void bn_block_totals_read(const unsigned long long* __restrict__ acc, const long long* __restrict__ slots, int ld,
                                                     int S, int C, int c, double& t0, double& t1) {
    if (slots[(size_t)S * 2 * ld * 2] != 0) {
        t0 = t1 = __longlong_as_double(0x7ff8000000000000LL);
        return;
    }
    t0 = bn_limbs_to_double((long long)acc[c * 2], (long long)acc[c * 2 + 1]);
    t1 = bn_limbs_to_double((long long)acc[(C + c) * 2], (long long)acc[(C + c) * 2 + 1]);
}

__device__ __forceinline__ void bn_fold_prologue(const BnFold& f, int C, float* s_sc, float* s_sh, bool first_block) {
    __shared__ unsigned long long bn_acc[2 * BN_FOLD_WIDE * 2];   // [stat][c][limb]
    const bool spread = C <= BN_FOLD_WIDE && f.nslots > 4;
    if (spread) bn_slot_block_totals(f.slots, f.ld, f.nslots, C, bn_acc);
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        double s, q;
        if (spread) {
            bn_block_totals_read(bn_acc, f.slots, f.ld, f.nslots, C, c, s, q);
        } else {
            bn_slot_totals(f.slots, f.ld, f.nslots, c, s, q);
        }
        const double mu = s / (double)f.count;
        double var = q / (double)f.count - mu * mu;
        if (var < 0.0) var = 0.0;
        const float fmu = (float)mu;
        const float fis = (float)(1.0 / sqrt(var + (double)f.eps));
        const float sc = f.gamma[c] * fis;
        const float sh = fmaf(-fmu, sc, f.beta[c]);
        s_sc[c] = sc;
        s_sh[c] = sh;
        if (first_block) {
            f.scale_out[c] = sc;
            f.shift_out[c] = sh;
            if (f.mean_out) {
                f.mean_out[c] = fmu;
                f.invstd_out[c] = fis;
            }
            const double unbiased = f.count > 1 ? var * ((double)f.count / (double)(f.count - 1)) : var;
            const bool fin = isfinite(fmu) && isfinite((float)unbiased);   // (as bn_finalize_kernel: a NaN statistic never enters the buffers)
            if (f.rmean && fin) f.rmean[c] = (1.f - f.momentum) * f.rmean[c] + f.momentum * fmu;
            if (f.rvar && fin) f.rvar[c] = (1.f - f.momentum) * f.rvar[c] + f.momentum * (float)unbiased;
        }
    }
    if (first_block && threadIdx.x == 0 && f.nbt) *f.nbt += 1;
    __syncthreads();
}

// A statistics destination is either the classic [2][ld][P] fp64 partials or the statistic slots above -- then bit 0 of the pointer is
// set and bits 1..3 hold log2 S (slot buffers are 16-byte aligned): a device-code convention between the host launchers of this library
// and its kernels, never part of the C ABI.
__device__ __forceinline__ void bn_stat_out(double* partial, int which, int ld, int c, unsigned P, unsigned p, double t) {
    const uintptr_t a = (uintptr_t)partial;
    if (a & 1) {
        const unsigned S = 1u << ((a >> 1) & 7);
        bn_slot_add(reinterpret_cast<long long*>(a & ~(uintptr_t)15), ld, S, p & (S - 1u), which, c, t);
    }
    else partial[((size_t)which * ld + c) * P + p] = t;
}
static inline double* bn_tag_slots(void* slots, int nslots) {
    int lg = 0;
    while ((1 << lg) < nslots) ++lg;
    return reinterpret_cast<double*>((uintptr_t)slots | 1u | ((uintptr_t)lg << 1));
}
// host-side check of a (slots, nslots) pair handed in through the C ABI
#define OTVAE_REQUIRE_SLOTS(who, slots, nslots) \
    OTVAE_REQUIRE(!(slots) || ((((uintptr_t)(slots)) & 15) == 0 && bn_slots_ok(nslots)), \
                  "%s: statistic slots must be 16-byte aligned with a power of two <= 64 slots in use (got %d)", who, (int)(nslots))

// host side of otvae_bn_fold -> the device descriptor (a.slots == NULL: no fold, *f is cleared)
static inline int bn_fold_from_abi(const char* who, const otvae_bn_fold& a, int C, BnFold* f) {
    *f = BnFold{};
    if (!a.slots) return OTVAE_OK;
    OTVAE_REQUIRE(a.ld >= C && a.count > 0 && a.gamma && a.beta && a.scale_out && a.shift_out, "%s: incomplete BatchNorm fold descriptor", who);
    OTVAE_REQUIRE((a.mean_out == nullptr) == (a.invstd_out == nullptr), "%s: fold.mean_out and fold.invstd_out come together", who);
    OTVAE_REQUIRE_SLOTS(who, a.slots, a.nslots);
    f->slots = (const long long*)a.slots;
    f->ld = a.ld;
    f->nslots = a.nslots;
    f->count = a.count;
    f->eps = a.eps;
    f->momentum = a.momentum;
    f->gamma = a.gamma;
    f->beta = a.beta;
    f->rmean = a.running_mean;
    f->rvar = a.running_var;
    f->nbt = (long long*)a.num_batches_tracked;
    f->mean_out = a.mean_out;
    f->invstd_out = a.invstd_out;
    f->scale_out = a.scale_out;
    f->shift_out = a.shift_out;
    return OTVAE_OK;
}

#define BN_TAB 1024   // channels a forward kernel keeps the BatchNorm affine of its input for in LDS (fold or copy)

// (scale, shift) of the kernel's input into LDS: from the fold (every block computes them from the slots) or copied from the global
// arrays a finalize launch left.  Ends with a block barrier.
__device__ __forceinline__ void bn_tab_fill(const BnFold& f, const float* __restrict__ scale, const float* __restrict__ shift, int C,
                                            float* s_sc, float* s_sh, bool first_block) {
    if (f.slots) {
        bn_fold_prologue(f, C, s_sc, s_sh, first_block);
    } else {
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            s_sc[c] = scale[c];
            s_sh[c] = shift[c];
        }
        __syncthreads();
    }
}
