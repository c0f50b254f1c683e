// ConvLayer kernels for the few layers whose channel counts are tiny (1, 3 or <= 8 on both sides): the image-side
// layers of the CNN (first encoder block: 1 -> 8 channels at 32x32; last decoder block: 8 -> 1, 1 -> 1, 1 -> 3 at 32x32).
// There an MFMA tile would be >= 75 % padding and the LDS-staged gather cannot use vector loads, so these layers --
// the highest-resolution, purely HBM-bound ones -- run as direct VALU convolutions: one lane per output position,
// weights broadcast from LDS, same fused prologue (BatchNorm-apply, ReLU, nearest up-sampling, zero padding) and
// epilogues (bias, residual; ReLU mask + BatchNorm-backward sums in fp64) as the implicit-GEMM kernels in conv.hip.
// Reference arithmetic: networks/cnn.py:183-192 and its autograd backward.
#include "common.h"
#include "conv_small.h"

#define SMALL_MAXC 8
#define SMALL_MAXW 1024

bool conv_small_ok(const SmallGeom& g) {
    if (g.Cs > SMALL_MAXC || g.Cn > SMALL_MAXC) return false;
    if (g.KH * g.KW * g.Cs * g.Cn > SMALL_MAXW) return false;
    if ((g.Cs % 4 == 0) && (g.Cn % 4 == 0)) return false;  // vectorised MFMA path handles these well
    return true;
}

__device__ __forceinline__ float act1(float v, float sc, float sh, bool affine, int relu) {
    if (affine) v = fmaf(v, sc, sh);
    if (relu) v = fmaxf(v, 0.f);
    return v;
}

// ------------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(256) void conv_small_fwd_kernel(SmallGeom g, const float* __restrict__ x,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             int relu, const float* __restrict__ wT,
                                                             const float* __restrict__ bias, const float* __restrict__ res,
                                                             float* __restrict__ y, double* __restrict__ partial, int CnPad) {
    __shared__ float w_s[SMALL_MAXW];
    __shared__ double redf[4][2][SMALL_MAXC];
    __shared__ float sc_s[SMALL_MAXC], sh_s[SMALL_MAXC], b_s[SMALL_MAXC];
    const int T = g.KH * g.KW;
    for (int i = threadIdx.x; i < T * g.Cs * g.Cn; i += 256) w_s[i] = wT[i];
    if (threadIdx.x < SMALL_MAXC) {
        const int c = threadIdx.x;
        sc_s[c] = (scale && c < g.Cs) ? scale[c] : 1.f;
        sh_s[c] = (scale && c < g.Cs) ? shift[c] : 0.f;
        b_s[c] = (bias && c < g.Cn) ? bias[c] : 0.f;
    }
    __syncthreads();
    const bool affine = scale != nullptr;
    const int Hu = g.Hs * g.up, Wu = g.Ws * g.up, ush = g.up - 1;
    const unsigned M = (unsigned)g.N * g.Ho * g.Wo;
    double s1[SMALL_MAXC], s2[SMALL_MAXC];
#pragma unroll
    for (int j = 0; j < SMALL_MAXC; ++j) s1[j] = s2[j] = 0.0;
    for (unsigned m = blockIdx.x * 256 + threadIdx.x; m < M; m += gridDim.x * 256) {
        const int ox = m % g.Wo;
        const unsigned t0 = m / g.Wo;
        const int oy = t0 % g.Ho;
        const int n = t0 / g.Ho;
        float acc[SMALL_MAXC];
#pragma unroll
        for (int j = 0; j < SMALL_MAXC; ++j) acc[j] = b_s[j];
        for (int kh = 0; kh < g.KH; ++kh) {
            const int iy = oy * g.stride + kh - g.pad;
            if ((unsigned)iy >= (unsigned)Hu) continue;
            for (int kw = 0; kw < g.KW; ++kw) {
                const int ix = ox * g.stride + kw - g.pad;
                if ((unsigned)ix >= (unsigned)Wu) continue;
                const float* xp = x + ((size_t)((unsigned)n * g.Hs + (iy >> ush)) * g.Ws + (ix >> ush)) * g.Cs;
                const float* wp = w_s + (kh * g.KW + kw) * g.Cs * g.Cn;
                for (int c = 0; c < g.Cs; ++c) {
                    const float a = act1(xp[c], sc_s[c], sh_s[c], affine, relu);
#pragma unroll
                    for (int j = 0; j < SMALL_MAXC; ++j)
                        if (j < g.Cn) acc[j] = fmaf(a, wp[c * g.Cn + j], acc[j]);
                }
            }
        }
        float* yp = y + (size_t)m * g.Cn;
        const float* rp = res ? res + (size_t)m * g.Cn : nullptr;
#pragma unroll
        for (int j = 0; j < SMALL_MAXC; ++j)
            if (j < g.Cn) {
                const float v = acc[j] + (rp ? rp[j] : 0.f);
                yp[j] = v;
                if (partial) {
                    s1[j] += (double)v;
                    s2[j] += (double)v * (double)v;
                }
            }
    }
    if (partial) {  // per-channel sums of the output = the next layer's BatchNorm statistics
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int j = 0; j < SMALL_MAXC; ++j) {
            const double a = wave_sum(s1[j]), b = wave_sum(s2[j]);
            if (lane == 0) {
                redf[wave][0][j] = a;
                redf[wave][1][j] = b;
            }
        }
        __syncthreads();
        if (threadIdx.x < 2 * SMALL_MAXC) {
            const int which = threadIdx.x / SMALL_MAXC, c = threadIdx.x % SMALL_MAXC;
            partial[((size_t)which * CnPad + c) * gridDim.x + blockIdx.x] =
                (redf[0][which][c] + redf[1][which][c]) + (redf[2][which][c] + redf[3][which][c]);
        }
    }
}

int conv_small_fwd(const SmallGeom& g, int nblocks, const float* x, const float* scale, const float* shift, int relu,
                   const float* wT, const float* bias, const float* res, float* y, double* partial, int CnPad,
                   hipStream_t st) {
    conv_small_fwd_kernel<<<nblocks, 256, 0, st>>>(g, x, scale, shift, relu, wT, bias, res, y, partial, CnPad);
    return 0;
}

// ------------------------------------------------------------------------------------------------ data gradient
// one lane per SOURCE position (parent pixel when up == 2): all Cs channels, children summed in registers
__global__ __launch_bounds__(256) void conv_small_dgrad_kernel(SmallGeom g, const float* __restrict__ gy,
                                                               const float* __restrict__ wD, const float* __restrict__ x,
                                                               const float* __restrict__ scale, const float* __restrict__ shift,
                                                               int relu, const float* __restrict__ mean,
                                                               const float* __restrict__ invstd, float* __restrict__ gv,
                                                               double* __restrict__ partial, int CsPad) {
    __shared__ float w_s[SMALL_MAXW];  // wD[t][co][c]
    __shared__ double red[4][2][SMALL_MAXC];
    const int T = g.KH * g.KW;
    for (int i = threadIdx.x; i < T * g.Cs * g.Cn; i += 256) w_s[i] = wD[i];
    __syncthreads();
    const unsigned P = (unsigned)g.N * g.Hs * g.Ws;
    const int nchild = g.up * g.up;
    const int smask = g.stride - 1, sshift = g.stride - 1;
    double s1[SMALL_MAXC], s2[SMALL_MAXC];
#pragma unroll
    for (int c = 0; c < SMALL_MAXC; ++c) s1[c] = s2[c] = 0.0;
    for (unsigned p = blockIdx.x * 256 + threadIdx.x; p < P; p += gridDim.x * 256) {
        const int sx = p % g.Ws;
        const unsigned t0 = p / g.Ws;
        const int sy = t0 % g.Hs;
        const int n = t0 / g.Hs;
        float acc[SMALL_MAXC];
#pragma unroll
        for (int c = 0; c < SMALL_MAXC; ++c) acc[c] = 0.f;
        for (int ch = 0; ch < nchild; ++ch) {
            const int iy = sy * g.up + (ch >> 1), ix = sx * g.up + (ch & 1);
            for (int kh = 0; kh < g.KH; ++kh) {
                const int ty = iy + g.pad - kh;
                if (ty < 0 || (ty & smask) != 0) continue;  // stride is 1 or 2
                const int oy = ty >> sshift;
                if (oy >= g.Ho) continue;
                for (int kw = 0; kw < g.KW; ++kw) {
                    const int tx = ix + g.pad - kw;
                    if (tx < 0 || (tx & smask) != 0) continue;
                    const int ox = tx >> sshift;
                    if (ox >= g.Wo) continue;
                    const float* gp = gy + ((size_t)((unsigned)n * g.Ho + oy) * g.Wo + ox) * g.Cn;
                    const float* wp = w_s + (kh * g.KW + kw) * g.Cn * g.Cs;
                    for (int co = 0; co < g.Cn; ++co) {
                        const float gvv = gp[co];
#pragma unroll
                        for (int c = 0; c < SMALL_MAXC; ++c)
                            if (c < g.Cs) acc[c] = fmaf(gvv, wp[co * g.Cs + c], acc[c]);
                    }
                }
            }
        }
        const size_t o = (size_t)p * g.Cs;
#pragma unroll
        for (int c = 0; c < SMALL_MAXC; ++c) {
            if (c < g.Cs) {
                float val = acc[c];
                float xv = 0.f;
                if (relu || mean) xv = x[o + c];
                if (relu) {
                    const float v = scale ? fmaf(xv, scale[c], shift[c]) : xv;
                    val = v > 0.f ? val : 0.f;
                }
                gv[o + c] = val;
                if (mean) {
                    s1[c] += (double)val;
                    s2[c] += (double)val * (double)((xv - mean[c]) * invstd[c]);
                }
            }
        }
    }
    if (mean) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int c = 0; c < SMALL_MAXC; ++c) {
            const double a = wave_sum(s1[c]), b = wave_sum(s2[c]);
            if (lane == 0) {
                red[wave][0][c] = a;
                red[wave][1][c] = b;
            }
        }
        __syncthreads();
        if (threadIdx.x < 2 * SMALL_MAXC) {
            const int which = threadIdx.x / SMALL_MAXC, c = threadIdx.x % SMALL_MAXC;
            const double t = (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]);
            partial[((size_t)which * CsPad + c) * gridDim.x + blockIdx.x] = t;
        }
        // columns [SMALL_MAXC, CsPad) of the partial rows are never read (finalize only touches c < Cs)
    }
}

int conv_small_dgrad(const SmallGeom& g, int nblocks, const float* gy, const float* wD, const float* x, const float* scale,
                     const float* shift, int relu, const float* mean, const float* invstd, float* gv, double* partial,
                     int CsPad, hipStream_t st) {
    conv_small_dgrad_kernel<<<nblocks, 256, 0, st>>>(g, gy, wD, x, scale, shift, relu, mean, invstd, gv, partial, CsPad);
    return 0;
}

// ------------------------------------------------------------------------------------------------ weight gradient
// one lane per output pixel (grid-stride), every weight element accumulated in a register, then one fixed-order
// wave/LDS reduction per block -> partial[block][K+hasb][Cn] (reduced over blocks by wgrad_reduce_kernel)
template <int T, int CS, int CN>
__global__ __launch_bounds__(256) void conv_small_wgrad_kernel(SmallGeom g, const float* __restrict__ x,
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
    const unsigned mbeg = blockIdx.x * chunk, mend = min(M, mbeg + chunk);
    for (unsigned m = mbeg + threadIdx.x; m < mend; m += 256) {
        const int ox = m % g.Wo;
        const unsigned t0 = m / g.Wo;
        const int oy = t0 % g.Ho;
        const int n = t0 / g.Ho;
        float gg[CN];
#pragma unroll
        for (int j = 0; j < CN; ++j) {
            gg[j] = gy[(size_t)m * CN + j];
            accb[j] += gg[j];
        }
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int kh = t / g.KW, kw = t - kh * g.KW;
            const int iy = oy * g.stride + kh - g.pad, ix = ox * g.stride + kw - g.pad;
            if ((unsigned)iy < (unsigned)Hu && (unsigned)ix < (unsigned)Wu) {
                const float* xp = x + ((size_t)((unsigned)n * g.Hs + (iy >> ush)) * g.Ws + (ix >> ush)) * CS;
#pragma unroll
                for (int c = 0; c < CS; ++c) {
                    const float a = act1(xp[c], sc[c], sh[c], affine, relu);
#pragma unroll
                    for (int j = 0; j < CN; ++j) acc[t][c][j] = fmaf(a, gg[j], acc[t][c][j]);
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
           (T == 1 && g.Cs == 1 && g.Cn == 3) || (T == 1 && g.Cs == 1 && g.Cn == 1) || (T == 1 && g.Cs == 8 && g.Cn == 1);
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
    else return -1;
#undef SW
    return P;
}
