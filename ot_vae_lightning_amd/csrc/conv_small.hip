// ConvLayer kernels for the few layers whose channel counts are tiny (1, 3 or <= 8 on both sides): the image-side
// layers of the CNN (first encoder block: 1 -> 8 channels at 32x32; last decoder block: 8 -> 1, 1 -> 1, 1 -> 3 at 32x32).
// There an MFMA tile would be >= 75 % padding and the LDS-staged gather cannot use vector loads, so these layers --
// the highest-resolution, purely HBM-bound ones -- run as direct VALU convolutions: one lane per output position,
// weights broadcast from LDS, same fused prologue (BatchNorm-apply, ReLU, nearest up-sampling, zero padding) and
// epilogues (bias, residual; ReLU mask + BatchNorm-backward sums in fp64) as the implicit-GEMM kernels in conv.hip.
// Reference arithmetic: networks/cnn.py:183-192 and its autograd backward.
#include "common.h"
#include "conv_small.h"
#include <type_traits>

#define SMALL_MAXC 16
// The weights stay in LDS: left alone the compiler hoists the (loop-invariant) LDS weight reads of the unrolled tap loops
// out of the position loop into 130-330 registers, which costs the occupancy these latency-bound kernels live on.
#define SMALL_REREAD_LDS() asm volatile("" ::: "memory")
#define SMALL_OCC
#define SMALL_MAXW 1024

bool conv_small_ok(const SmallGeom& g) {
    if (g.KH * g.KW * g.Cs * g.Cn > SMALL_MAXW) return false;
    if (g.Cs <= 8 && g.Cn <= 8) return !((g.Cs % 4 == 0) && (g.Cn % 4 == 0));  // those the vectorised MFMA path handles well
    // the RGB image-side layers of the capacity-16 (CIFAR) network: 3 -> 16, 16 -> 3, 3 -> 9 (qkv); otherwise they fall to
    // the scalar-gather implicit GEMM (3 channels cannot be loaded as float4)
    return (g.Cs == 3 && (g.Cn == 16 || g.Cn == 9)) || (g.Cs == 16 && g.Cn == 3);
}

// (image, y, x) of a flat position index: one multiply per division while the index is below 2^22 (see fast_div in
// conv.hip), the udiv expansion (~25 instructions each, a third of a 1-channel layer's per-pixel work) otherwise
__device__ __forceinline__ unsigned sdiv(unsigned k, unsigned d, float inv_d, bool small) {
    return small ? (unsigned)(int)(((float)(int)k + 0.5f) * inv_d) : k / d;
}

__device__ __forceinline__ float act1(float v, float sc, float sh, bool affine, int relu) {
    if (affine) v = fmaf(v, sc, sh);
    if (relu) v = fmaxf(v, 0.f);
    return v;
}

// ------------------------------------------------------------------------------------------------ forward
// CS / CN > 0: channel counts known at compile time (inner loops fully unrolled, channel vectors loaded as one 16/32-byte
// access); 0: read from the geometry (any Cs, Cn <= SMALL_MAXC).
template <int C>
__device__ __forceinline__ void load_chan(const float* __restrict__ p, float (&v)[C ? C : SMALL_MAXC], int n) {
    if constexpr (C == 16) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 a = reinterpret_cast<const float4*>(p)[q];
            v[4 * q] = a.x; v[4 * q + 1] = a.y; v[4 * q + 2] = a.z; v[4 * q + 3] = a.w;
        }
    } else if constexpr (C == 8) {
        const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else if constexpr (C == 4) {
        const float4 a = *reinterpret_cast<const float4*>(p);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    } else if constexpr (C > 0) {
#pragma unroll
        for (int c = 0; c < C; ++c) v[c] = p[c];
    } else {
#pragma unroll
        for (int c = 0; c < SMALL_MAXC; ++c) v[c] = c < n ? p[c] : 0.f;
    }
}

template <int C>
__device__ __forceinline__ void store_chan(float* __restrict__ p, const float (&v)[C ? C : SMALL_MAXC], int n) {
    if constexpr (C == 16) {
#pragma unroll
        for (int q = 0; q < 4; ++q) reinterpret_cast<float4*>(p)[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    } else if constexpr (C == 8) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else if constexpr (C == 4) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    } else if constexpr (C > 0) {
#pragma unroll
        for (int c = 0; c < C; ++c) p[c] = v[c];
    } else {
#pragma unroll
        for (int c = 0; c < SMALL_MAXC; ++c)
            if (c < n) p[c] = v[c];
    }
}

// KS > 0 (with CS, CN > 0): square KS x KS kernel known at compile time.  The taps' loads are then issued back to back
// from clamped (always valid) addresses and masked afterwards; the generic tap loop below branches around out-of-image
// taps, which makes every tap's load wait for the previous tap's arithmetic (9-16 exposed L2 latencies per position:
// the layers at 32x32 ran at 0.7-1.7 TB/s).
template <int CS, int CN, int KS>
__global__ __launch_bounds__(256) SMALL_OCC void conv_small_fwd_kernel(SmallGeom g, const float* __restrict__ x,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             int relu, const float* __restrict__ wT,
                                                             const float* __restrict__ bias, const float* __restrict__ res,
                                                             float* __restrict__ y, double* __restrict__ partial, int CnPad,
                                                             BnFold fold) {
    constexpr int CSM = CS ? CS : SMALL_MAXC, CNM = CN ? CN : SMALL_MAXC;
    const int Cs = CS ? CS : g.Cs, Cn = CN ? CN : g.Cn;
    __shared__ __align__(16) float w_s[SMALL_MAXW];  // [tap][c][CNM]  (row padded to CNM when CN is generic)
    __shared__ double redf[4][2][SMALL_MAXC];
    __shared__ float sc_s[SMALL_MAXC], sh_s[SMALL_MAXC], b_s[SMALL_MAXC];
    const int T = g.KH * g.KW;
    for (int i = threadIdx.x; i < T * Cs * Cn; i += 256) w_s[i] = wT[i];
    if (threadIdx.x < SMALL_MAXC) {
        const int c = threadIdx.x;
        sc_s[c] = (scale && c < Cs) ? scale[c] : 1.f;
        sh_s[c] = (scale && c < Cs) ? shift[c] : 0.f;
        b_s[c] = (bias && c < Cn) ? bias[c] : 0.f;
    }
    __syncthreads();
    if (fold.slots) bn_fold_prologue(fold, Cs, sc_s, sh_s, blockIdx.x == 0);  // the BatchNorm of x folded in (common.h); ends with a barrier
    const bool affine = scale != nullptr || fold.slots != nullptr;
    float sc[CSM], sh[CSM];
#pragma unroll
    for (int c = 0; c < CSM; ++c) {
        sc[c] = sc_s[c];
        sh[c] = sh_s[c];
    }
    const int Hu = g.Hs * g.up, Wu = g.Ws * g.up, ush = g.up - 1;
    const unsigned M = (unsigned)g.N * g.Ho * g.Wo;
    const float inv_wo = 1.0f / (float)g.Wo, inv_ho = 1.0f / (float)g.Ho;
    const bool small = M < (1u << 22);
    double s1[CNM], s2[CNM];
#pragma unroll
    for (int j = 0; j < CNM; ++j) s1[j] = s2[j] = 0.0;
    for (unsigned m = blockIdx.x * 256 + threadIdx.x; m < M; m += gridDim.x * 256) {
        const unsigned t0 = sdiv(m, g.Wo, inv_wo, small);
        const int ox = m - t0 * g.Wo;
        const int n = sdiv(t0, g.Ho, inv_ho, small);
        const int oy = t0 - (unsigned)n * g.Ho;
        float acc[CNM];
#pragma unroll
        for (int j = 0; j < CNM; ++j) acc[j] = b_s[j];
        if constexpr (KS > 0 && CS > 0 && CN > 0) {
            // taps in groups whose loads (<= 32 registers) are in flight together; the clobber keeps a group's loads from
            // drifting above the previous group's arithmetic (and the weights in LDS, see SMALL_REREAD_LDS)
            constexpr int GS = (KS * KS * CS <= 32) ? KS * KS : KS;  // all taps at once, or one kernel row at a time
#pragma unroll
            for (int t0g = 0; t0g < KS * KS; t0g += GS) {
                SMALL_REREAD_LDS();
                float xv[GS][CS];
                int ok[GS];  // all-ones / zero bit mask: applied with an AND (a select here becomes a branch per channel)
#pragma unroll
                for (int i = 0; i < GS; ++i) {
                    const int t = t0g + i;
                    const int iy = oy * g.stride + t / KS - g.pad, ix = ox * g.stride + t % KS - g.pad;
                    ok[i] = ((unsigned)iy < (unsigned)Hu && (unsigned)ix < (unsigned)Wu) ? -1 : 0;
                    const int cy = min(max(iy, 0), Hu - 1) >> ush, cx = min(max(ix, 0), Wu - 1) >> ush;
                    load_chan<CS>(x + ((size_t)((unsigned)n * g.Hs + cy) * g.Ws + cx) * CS, xv[i], CS);
                }
#pragma unroll
                for (int i = 0; i < GS; ++i) {
                    const float* wp = w_s + (t0g + i) * CS * CN;
#pragma unroll
                    for (int c = 0; c < CS; ++c) {
                        const float a = __int_as_float(__float_as_int(act1(xv[i][c], sc[c], sh[c], affine, relu)) & ok[i]);
#pragma unroll
                        for (int j = 0; j < CN; ++j) acc[j] = fmaf(a, wp[c * CN + j], acc[j]);
                    }
                }
            }
        } else {
            for (int kh = 0; kh < g.KH; ++kh) {
                const int iy = oy * g.stride + kh - g.pad;
                if ((unsigned)iy >= (unsigned)Hu) continue;
                for (int kw = 0; kw < g.KW; ++kw) {
                    const int ix = ox * g.stride + kw - g.pad;
                    if ((unsigned)ix >= (unsigned)Wu) continue;
                    float xv[CSM];
                    load_chan<CS>(x + ((size_t)((unsigned)n * g.Hs + (iy >> ush)) * g.Ws + (ix >> ush)) * Cs, xv, Cs);
                    const float* wp = w_s + (kh * g.KW + kw) * Cs * Cn;
#pragma unroll
                    for (int c = 0; c < CSM; ++c) {
                        if (CS || c < Cs) {
                            const float a = act1(xv[c], sc[c], sh[c], affine, relu);
#pragma unroll
                            for (int j = 0; j < CNM; ++j)
                                if (CN || j < Cn) acc[j] = fmaf(a, wp[c * Cn + j], acc[j]);
                        }
                    }
                }
            }
        }
        if (res) {
            float rv[CNM];
            load_chan<CN>(res + (size_t)m * Cn, rv, Cn);
#pragma unroll
            for (int j = 0; j < CNM; ++j) acc[j] += rv[j];
        }
        store_chan<CN>(y + (size_t)m * Cn, acc, Cn);
        if (partial) {
#pragma unroll
            for (int j = 0; j < CNM; ++j) {
                s1[j] += (double)acc[j];
                s2[j] += (double)acc[j] * (double)acc[j];
            }
        }
    }
    if (partial) {  // per-channel sums of the output = the next layer's BatchNorm statistics
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int j = 0; j < CNM; ++j) {
            const double a = wave_sum(s1[j]), b = wave_sum(s2[j]);
            if (lane == 0) {
                redf[wave][0][j] = a;
                redf[wave][1][j] = b;
            }
        }
        __syncthreads();
        if (threadIdx.x < 2 * SMALL_MAXC) {
            const int which = threadIdx.x / SMALL_MAXC, c = threadIdx.x % SMALL_MAXC;
            if (c < Cn)
                bn_stat_out(partial, which, CnPad, c, gridDim.x, blockIdx.x,
                            (redf[0][which][c] + redf[1][which][c]) + (redf[2][which][c] + redf[3][which][c]));
        }
    }
}

// channel-count specialisations: the MNIST (1 <-> 8, 1 -> 1, 1 -> 3) and RGB (3 <-> 8, 3 -> 3) image-side layers
#define SMALL_DISPATCH(KERNEL, ...)                                            \
    do {                                                                       \
        if (g.Cs == 1 && g.Cn == 8) KERNEL<1, 8, 0> __VA_ARGS__;               \
        else if (g.Cs == 8 && g.Cn == 1) KERNEL<8, 1, 0> __VA_ARGS__;          \
        else if (g.Cs == 1 && g.Cn == 1) KERNEL<1, 1, 0> __VA_ARGS__;          \
        else if (g.Cs == 1 && g.Cn == 3) KERNEL<1, 3, 0> __VA_ARGS__;          \
        else if (g.Cs == 3 && g.Cn == 8) KERNEL<3, 8, 0> __VA_ARGS__;          \
        else if (g.Cs == 8 && g.Cn == 3) KERNEL<8, 3, 0> __VA_ARGS__;          \
        else if (g.Cs == 3 && g.Cn == 3) KERNEL<3, 3, 0> __VA_ARGS__;          \
        else if (g.Cs == 3 && g.Cn == 16) KERNEL<3, 16, 0> __VA_ARGS__;        \
        else if (g.Cs == 16 && g.Cn == 3) KERNEL<16, 3, 0> __VA_ARGS__;        \
        else if (g.Cs == 3 && g.Cn == 9) KERNEL<3, 9, 0> __VA_ARGS__;          \
        else KERNEL<0, 0, 0> __VA_ARGS__;                                      \
    } while (0)
// (Cs, Cn, kernel size) of the layers the two configurations really have: tap loops unrolled, loads up front
#define SMALL_KS_CASES(X) \
    X(1, 8, 4) X(8, 1, 3) X(8, 1, 1) X(1, 1, 3) X(1, 1, 1) X(1, 3, 1) X(3, 3, 3) X(3, 3, 1) X(3, 16, 4) X(16, 3, 3) X(16, 3, 1) X(3, 9, 1)

int conv_small_fwd(const SmallGeom& g, int nblocks, const float* x, const float* scale, const float* shift, int relu,
                   const float* wT, const float* bias, const float* res, float* y, double* partial, int CnPad,
                   const BnFold& fold, hipStream_t st) {
#define X(CS_, CN_, KS_)                                                                                          \
    if (g.Cs == CS_ && g.Cn == CN_ && g.KH == KS_ && g.KW == KS_) {                                               \
        conv_small_fwd_kernel<CS_, CN_, KS_><<<nblocks, 256, 0, st>>>(g, x, scale, shift, relu, wT, bias, res, y, partial, \
                                                                       CnPad, fold);                              \
        return 0;                                                                                                 \
    }
    SMALL_KS_CASES(X)
#undef X
    SMALL_DISPATCH(conv_small_fwd_kernel, <<<nblocks, 256, 0, st>>>(g, x, scale, shift, relu, wT, bias, res, y, partial, CnPad, fold));
    return 0;
}

// ------------------------------------------------------------------------------------------------ data gradient
// one lane per SOURCE position (parent pixel when up == 2): all Cs channels, children summed in registers
// KSU = 0: generic tap loops.  KSU = KS * 16 + STRIDE * 4 + UP (all > 0): square kernel, stride and up-sampling factor
// known at compile time; the (children x parity-class taps) loads are issued up front from clamped addresses and masked.
template <int CS, int CN, int KSU>
__global__ __launch_bounds__(256) SMALL_OCC void conv_small_dgrad_kernel(SmallGeom g, const float* __restrict__ gy,
                                                               const float* __restrict__ wD, const float* __restrict__ x,
                                                               const float* __restrict__ scale, const float* __restrict__ shift,
                                                               int relu, const float* __restrict__ mean,
                                                               const float* __restrict__ invstd, float* __restrict__ gv,
                                                               double* __restrict__ partial, int CsPad) {
    constexpr int CSM = CS ? CS : SMALL_MAXC, CNM = CN ? CN : SMALL_MAXC;
    const int Cs = CS ? CS : g.Cs, Cn = CN ? CN : g.Cn;
    __shared__ __align__(16) float w_s[SMALL_MAXW];  // wD[t][co][c]
    __shared__ double red[4][2][SMALL_MAXC];
    const int T = g.KH * g.KW;
    for (int i = threadIdx.x; i < T * Cs * Cn; i += 256) w_s[i] = wD[i];
    __syncthreads();
    float sc[CSM], sh[CSM], mu[CSM], is[CSM];
#pragma unroll
    for (int c = 0; c < CSM; ++c) {
        const bool in = CS || c < Cs;
        sc[c] = (scale && in) ? scale[c] : 1.f;
        sh[c] = (scale && in) ? shift[c] : 0.f;
        mu[c] = (mean && in) ? mean[c] : 0.f;
        is[c] = (mean && in) ? invstd[c] : 0.f;
    }
    const unsigned P = (unsigned)g.N * g.Hs * g.Ws;
    const float inv_ws = 1.0f / (float)g.Ws, inv_hs = 1.0f / (float)g.Hs;
    const bool small = P < (1u << 22);
    const int nchild = g.up * g.up;
    const int smask = g.stride - 1, sshift = g.stride - 1;
    double s1[CSM], s2[CSM];
#pragma unroll
    for (int c = 0; c < CSM; ++c) s1[c] = s2[c] = 0.0;
    for (unsigned p = blockIdx.x * 256 + threadIdx.x; p < P; p += gridDim.x * 256) {
        const unsigned t0 = sdiv(p, g.Ws, inv_ws, small);
        const int sx = p - t0 * g.Ws;
        const int n = sdiv(t0, g.Hs, inv_hs, small);
        const int sy = t0 - (unsigned)n * g.Hs;
        float acc[CSM];
#pragma unroll
        for (int c = 0; c < CSM; ++c) acc[c] = 0.f;
        if constexpr (KSU > 0 && CS > 0 && CN > 0) {
            constexpr int KS = KSU >> 4, STRIDE = (KSU >> 2) & 3, UP = KSU & 3;
            constexpr int NT1 = (KS + STRIDE - 1) / STRIDE;  // taps of one parity class per dimension
            constexpr int NL = UP * UP * NT1 * NT1;
            constexpr int GS = (NL * CN <= 32) ? NL : NT1 * NT1;  // everything at once, or one child at a time
#pragma unroll 1  // a real loop over the children: unrolled, the address arithmetic of all NL taps is live at once
            for (int l0 = 0; l0 < NL; l0 += GS) {
                SMALL_REREAD_LDS();
                float gg[GS][CN];
                int tap[GS];  // kh * KS + kw, or -1 when the tap does not reach an output position
#pragma unroll
                for (int i = 0; i < GS; ++i) {
                    const int l = l0 + i;
                    const int ch = l / (NT1 * NT1), a = (l / NT1) % NT1, b = l % NT1;
                    const int iy = sy * UP + ch / UP, ix = sx * UP + ch % UP;
                    const int kh = ((iy + g.pad) & (STRIDE - 1)) + a * STRIDE, kw = ((ix + g.pad) & (STRIDE - 1)) + b * STRIDE;
                    const int ty = iy + g.pad - kh, tx = ix + g.pad - kw;
                    const int oy = ty >> (STRIDE - 1), ox = tx >> (STRIDE - 1);
                    const bool ok = kh < KS && kw < KS && ty >= 0 && tx >= 0 && oy < g.Ho && ox < g.Wo;
                    tap[i] = ok ? kh * KS + kw : -1;
                    const int cy = min(max(oy, 0), g.Ho - 1), cx = min(max(ox, 0), g.Wo - 1);
                    load_chan<CN>(gy + ((size_t)((unsigned)n * g.Ho + cy) * g.Wo + cx) * CN, gg[i], CN);
                }
#pragma unroll
                for (int i = 0; i < GS; ++i) {
                    const float* wp = w_s + max(tap[i], 0) * CN * CS;
#pragma unroll
                    for (int co = 0; co < CN; ++co) {
                        const float gval = __int_as_float(__float_as_int(gg[i][co]) & ~(tap[i] >> 31));
#pragma unroll
                        for (int c = 0; c < CS; ++c) acc[c] = fmaf(gval, wp[co * CS + c], acc[c]);
                    }
                }
            }
        } else
        for (int ch = 0; ch < nchild; ++ch) {
            const int iy = sy * g.up + (ch >> 1), ix = sx * g.up + (ch & 1);
            // stride 2: only the taps of this position's parity class contribute; start there and step by the stride
            for (int kh = (iy + g.pad) & smask; kh < g.KH; kh += g.stride) {
                const int ty = iy + g.pad - kh;
                if (ty < 0) break;  // kh only grows
                const int oy = ty >> sshift;
                if (oy >= g.Ho) continue;
                for (int kw = (ix + g.pad) & smask; kw < g.KW; kw += g.stride) {
                    const int tx = ix + g.pad - kw;
                    if (tx < 0) break;
                    const int ox = tx >> sshift;
                    if (ox >= g.Wo) continue;
                    float gg[CNM];
                    load_chan<CN>(gy + ((size_t)((unsigned)n * g.Ho + oy) * g.Wo + ox) * Cn, gg, Cn);
                    const float* wp = w_s + (kh * g.KW + kw) * Cn * Cs;
#pragma unroll
                    for (int co = 0; co < CNM; ++co) {
                        if (CN || co < Cn) {
#pragma unroll
                            for (int c = 0; c < CSM; ++c)
                                if (CS || c < Cs) acc[c] = fmaf(gg[co], wp[co * Cs + c], acc[c]);
                        }
                    }
                }
            }
        }
        const size_t o = (size_t)p * Cs;
        float xv[CSM];
        if (relu || mean) load_chan<CS>(x + o, xv, Cs);
#pragma unroll
        for (int c = 0; c < CSM; ++c) {
            if (CS || c < Cs) {
                float val = acc[c];
                if (relu) {
                    const float v = scale ? fmaf(xv[c], sc[c], sh[c]) : xv[c];
                    val = v > 0.f ? val : 0.f;
                }
                acc[c] = val;
                if (mean) {
                    s1[c] += (double)val;
                    s2[c] += (double)val * (double)((xv[c] - mu[c]) * is[c]);
                }
            }
        }
        store_chan<CS>(gv + o, acc, Cs);
    }
    if (mean) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int c = 0; c < CSM; ++c) {
            const double a = wave_sum(s1[c]), b = wave_sum(s2[c]);
            if (lane == 0) {
                red[wave][0][c] = a;
                red[wave][1][c] = b;
            }
        }
        __syncthreads();
        if (threadIdx.x < 2 * SMALL_MAXC) {
            const int which = threadIdx.x / SMALL_MAXC, c = threadIdx.x % SMALL_MAXC;
            if (c < Cs) {
                const double t = (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]);
                bn_stat_out(partial, which, CsPad, c, gridDim.x, blockIdx.x, t);
            }
        }
    }
}

int conv_small_dgrad(const SmallGeom& g, int nblocks, const float* gy, const float* wD, const float* x, const float* scale,
                     const float* shift, int relu, const float* mean, const float* invstd, float* gv, double* partial,
                     int CsPad, hipStream_t st) {
#define X(CS_, CN_, KS_)                                                                                             \
    if (g.Cs == CS_ && g.Cn == CN_ && g.KH == KS_ && g.KW == KS_ && g.stride <= 2 && g.up <= 2) {                    \
        constexpr int K16 = KS_ * 16;                                                                                \
        auto go = [&](auto tag) {                                                                                    \
            conv_small_dgrad_kernel<CS_, CN_, decltype(tag)::value><<<nblocks, 256, 0, st>>>(                        \
                g, gy, wD, x, scale, shift, relu, mean, invstd, gv, partial, CsPad);                                 \
        };                                                                                                           \
        if (g.stride == 1 && g.up == 1) go(std::integral_constant<int, K16 + 4 + 1>{});                              \
        else if (g.stride == 1 && g.up == 2) go(std::integral_constant<int, K16 + 4 + 2>{});                         \
        else if (g.stride == 2 && g.up == 1) go(std::integral_constant<int, K16 + 8 + 1>{});                         \
        else go(std::integral_constant<int, K16 + 8 + 2>{});                                                         \
        return 0;                                                                                                    \
    }
    SMALL_KS_CASES(X)
#undef X
    SMALL_DISPATCH(conv_small_dgrad_kernel,
                   <<<nblocks, 256, 0, st>>>(g, gy, wD, x, scale, shift, relu, mean, invstd, gv, partial, CsPad));
    return 0;
}

// ------------------------------------------------------------------------------------------------ weight gradient
// one lane per output pixel (grid-stride), every weight element accumulated in a register, then one fixed-order
// wave/LDS reduction per block -> partial[block][K+hasb][Cn] (reduced over blocks by wgrad_reduce_kernel)
template <int T, int CS, int CN>
__global__ __launch_bounds__(256) SMALL_OCC void conv_small_wgrad_kernel(SmallGeom g, const float* __restrict__ x,
                                                               const float* __restrict__ scale, const float* __restrict__ shift,
                                                               int relu, const float* __restrict__ gy,
                                                               float* __restrict__ partial, int has_bias, unsigned chunk) {
    constexpr int NA = T * CS * CN;
    __shared__ float red[4][NA + CN];
    const bool affine = scale != nullptr;
    float sc[CS], sh[CS];
#pragma unroll
    for (int c = 0; c < CS; ++c) {
        sc[c] = affine ? scale[c] : 1.f;
        sh[c] = affine ? shift[c] : 0.f;
    }
    float acc[T][CS][CN], accb[CN];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int c = 0; c < CS; ++c)
#pragma unroll
            for (int j = 0; j < CN; ++j) acc[t][c][j] = 0.f;
#pragma unroll
    for (int j = 0; j < CN; ++j) accb[j] = 0.f;

    const int Hu = g.Hs * g.up, Wu = g.Ws * g.up, ush = g.up - 1;
    const unsigned M = (unsigned)g.N * g.Ho * g.Wo;
    const float inv_wo = 1.0f / (float)g.Wo, inv_ho = 1.0f / (float)g.Ho;
    const bool small = M < (1u << 22);
    const unsigned mbeg = blockIdx.x * chunk, mend = min(M, mbeg + chunk);
    for (unsigned m = mbeg + threadIdx.x; m < mend; m += 256) {
        const unsigned t0 = sdiv(m, g.Wo, inv_wo, small);
        const int ox = m - t0 * g.Wo;
        const int n = sdiv(t0, g.Ho, inv_ho, small);
        const int oy = t0 - (unsigned)n * g.Ho;
        float gg[CN];
#pragma unroll
        for (int j = 0; j < CN; ++j) {
            gg[j] = gy[(size_t)m * CN + j];
            accb[j] += gg[j];
        }
        // the taps' loads are issued from clamped (always valid) addresses ahead of the arithmetic, in groups of <= 32
        // registers, and out-of-image taps are masked afterwards: a branch per tap exposes one load latency per tap
        constexpr int GS = (T * CS <= 32) ? T : (T % 3 == 0 ? 3 : (T % 4 == 0 ? 4 : 1));
#pragma unroll
        for (int t0g = 0; t0g < T; t0g += GS) {
            asm volatile("" ::: "memory");
            float xv[GS][CS];
            int ok[GS];  // all-ones / zero bit mask
#pragma unroll
            for (int i = 0; i < GS; ++i) {
                const int t = t0g + i;
                const int kh = t / g.KW, kw = t - kh * g.KW;
                const int iy = oy * g.stride + kh - g.pad, ix = ox * g.stride + kw - g.pad;
                ok[i] = ((unsigned)iy < (unsigned)Hu && (unsigned)ix < (unsigned)Wu) ? -1 : 0;
                const int cy = min(max(iy, 0), Hu - 1) >> ush, cx = min(max(ix, 0), Wu - 1) >> ush;
                load_chan<CS>(x + ((size_t)((unsigned)n * g.Hs + cy) * g.Ws + cx) * CS, xv[i], CS);
            }
#pragma unroll
            for (int i = 0; i < GS; ++i) {
#pragma unroll
                for (int c = 0; c < CS; ++c) {
                    const float a = __int_as_float(__float_as_int(act1(xv[i][c], sc[c], sh[c], affine, relu)) & ok[i]);
#pragma unroll
                    for (int j = 0; j < CN; ++j) acc[t0g + i][c][j] = fmaf(a, gg[j], acc[t0g + i][c][j]);
                }
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int c = 0; c < CS; ++c)
#pragma unroll
            for (int j = 0; j < CN; ++j) {
                const float s = wave_sum(acc[t][c][j]);
                if (lane == 0) red[wave][(t * CS + c) * CN + j] = s;
            }
#pragma unroll
    for (int j = 0; j < CN; ++j) {
        const float s = wave_sum(accb[j]);
        if (lane == 0) red[wave][NA + j] = s;
    }
    __syncthreads();
    const int Kp = T * CS + (has_bias ? 1 : 0);
    float* out = partial + (size_t)blockIdx.x * Kp * CN;
    for (int e = threadIdx.x; e < Kp * CN; e += 256) out[e] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
}

bool conv_small_wgrad_ok(const SmallGeom& g) {
    if (!conv_small_ok(g)) return false;
    const int T = g.KH * g.KW;
    return (T == 16 && g.Cs == 1 && g.Cn == 8) || (T == 9 && g.Cs == 8 && g.Cn == 1) || (T == 9 && g.Cs == 1 && g.Cn == 1) ||
           (T == 1 && g.Cs == 1 && g.Cn == 3) || (T == 1 && g.Cs == 1 && g.Cn == 1) || (T == 1 && g.Cs == 8 && g.Cn == 1) ||
           // RGB image side (every weight element in a register: T * Cs * Cn <= 81)
           (T == 9 && g.Cs == 3 && g.Cn == 3) || (T == 1 && g.Cs == 3 && g.Cn == 3) || (T == 1 && g.Cs == 3 && g.Cn == 9) ||
           (T == 1 && g.Cs == 16 && g.Cn == 3);
}

void conv_small_wgrad_plan(const SmallGeom& g, int& P, unsigned& chunk) {
    const unsigned M = (unsigned)g.N * g.Ho * g.Wo;
    P = imax(1, imin(512, (int)(M / 1024)));
    chunk = (M + P - 1) / P;
    P = cdiv(M, chunk);
}

int conv_small_wgrad(const SmallGeom& g, const float* x, const float* scale, const float* shift, int relu, const float* gy,
                     int has_bias, float* partial, hipStream_t st) {
    int P;
    unsigned chunk;
    conv_small_wgrad_plan(g, P, chunk);
    const int T = g.KH * g.KW;
#define SW(T_, CS_, CN_) \
    conv_small_wgrad_kernel<T_, CS_, CN_><<<P, 256, 0, st>>>(g, x, scale, shift, relu, gy, partial, has_bias, chunk)
    if (T == 16 && g.Cs == 1 && g.Cn == 8) SW(16, 1, 8);
    else if (T == 9 && g.Cs == 8 && g.Cn == 1) SW(9, 8, 1);
    else if (T == 9 && g.Cs == 1 && g.Cn == 1) SW(9, 1, 1);
    else if (T == 1 && g.Cs == 1 && g.Cn == 3) SW(1, 1, 3);
    else if (T == 1 && g.Cs == 1 && g.Cn == 1) SW(1, 1, 1);
    else if (T == 1 && g.Cs == 8 && g.Cn == 1) SW(1, 8, 1);
    else if (T == 9 && g.Cs == 3 && g.Cn == 3) SW(9, 3, 3);
    else if (T == 1 && g.Cs == 3 && g.Cn == 3) SW(1, 3, 3);
    else if (T == 1 && g.Cs == 3 && g.Cn == 9) SW(1, 3, 9);
    else if (T == 1 && g.Cs == 16 && g.Cn == 3) SW(1, 16, 3);
    else return -1;
#undef SW
    return P;
}
