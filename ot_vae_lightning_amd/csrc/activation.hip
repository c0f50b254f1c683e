// Non-ReLU activations of ConvLayer (reference networks/cnn.py:128-147: LeakyReLU(0.2), SELU, GELU, SiLU) and the scalar
// weight / bias multipliers of `equalized_lr` (cnn.py:114-118,186-188).
//
// The fused convolution kernels (conv*.hip) carry ReLU in their prologues and epilogues: that is what every BASELINE
// configuration trains with.  The other activations run UNFUSED around the same kernels,
//     a = act(x * scale + shift)                         otvae_bn_act_fwd       (scale / shift: the BatchNorm affine, or NULL)
//     y = conv(up(a), w * s) + b * m                     the fused kernels with neither BatchNorm nor activation
//     g_u = g_a * act'(x * scale + shift) (+ the BatchNorm-backward sums)   otvae_bn_act_bwd
// so that the hot ReLU path keeps its instruction streams and register budgets; one extra pass over the layer's input each way.
#include "common.h"

#define ACT_NONE 0
#define ACT_RELU 1
#define ACT_LEAKY 2   // LeakyReLU(0.2)
#define ACT_SELU 3
#define ACT_GELU 4    // exact (erf) form: nn.GELU() default
#define ACT_SILU 5

#define SELU_ALPHA 1.6732632423543772848170429916717f
#define SELU_SCALE 1.0507009873554804934193349852946f

__device__ __forceinline__ float act_fwd(float u, int kind) {
    switch (kind) {
        case ACT_RELU: return fmaxf(u, 0.f);
        case ACT_LEAKY: return u > 0.f ? u : 0.2f * u;
        case ACT_SELU: return SELU_SCALE * (u > 0.f ? u : SELU_ALPHA * expm1f(u));
        case ACT_GELU: return 0.5f * u * (1.f + erff(u * 0.70710678118654752440f));
        case ACT_SILU: return u / (1.f + expf(-u));
        default: return u;
    }
}

// d act / d u (torch's conventions at u == 0: ReLU 0, LeakyReLU the slope, SELU the exponential branch)
__device__ __forceinline__ float act_grad(float u, int kind) {
    switch (kind) {
        case ACT_RELU: return u > 0.f ? 1.f : 0.f;
        case ACT_LEAKY: return u > 0.f ? 1.f : 0.2f;
        case ACT_SELU: return u > 0.f ? SELU_SCALE : SELU_SCALE * SELU_ALPHA * expf(u);
        case ACT_GELU: {
            const float cdf = 0.5f * (1.f + erff(u * 0.70710678118654752440f));
            const float pdf = 0.39894228040143267794f * expf(-0.5f * u * u);
            return cdf + u * pdf;
        }
        case ACT_SILU: {
            const float s = 1.f / (1.f + expf(-u));
            return s * (1.f + u * (1.f - s));
        }
        default: return 1.f;
    }
}

__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, int kind, int64_t total, int C,
                                                         float* __restrict__ out) {
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        float u = x[i];
        if (scale) u = fmaf(u, scale[c], shift[c]);
        out[i] = act_fwd(u, kind);
    }
}

extern "C" int otvae_bn_act_fwd(const float* x, const float* scale, const float* shift, int kind, int64_t M, int C, float* out,
                                void* stream) {
    OTVAE_REQUIRE(x && out && M > 0 && C > 0, "otvae_bn_act_fwd: bad argument");
    OTVAE_REQUIRE((scale == nullptr) == (shift == nullptr), "otvae_bn_act_fwd: scale and shift come together");
    OTVAE_REQUIRE(kind >= ACT_NONE && kind <= ACT_SILU, "otvae_bn_act_fwd: unknown activation %d", kind);
    const int64_t total = M * C;
    bn_act_fwd_kernel<<<imin(cdiv(total, 256), 4096), 256, 0, (hipStream_t)stream>>>(x, scale, shift, kind, total, C, out);
    OTVAE_CHECK_LAUNCH("otvae_bn_act_fwd");
    return OTVAE_OK;
}

// partial sums per block: the layout otvae_bn_bwd_finalize reads, partial[2][C][P] (fp64)
#define BNACT_ROWS 512  // rows of [M][C] per block
extern "C" int otvae_bn_act_bwd_parts(int64_t M) { return (int)cdiv(M, BNACT_ROWS); }

// g_u = g_a * act'(u), u = x * scale + shift; with mean / invstd also s1[c] = sum g_u, s2[c] = sum g_u * xhat over the block's rows
__global__ __launch_bounds__(256) void bn_act_bwd_kernel(const float* __restrict__ ga, const float* __restrict__ x,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         const float* __restrict__ mean, const float* __restrict__ invstd, int kind,
                                                         int64_t M, int C, float* __restrict__ gv, double* __restrict__ partial, int P) {
    __shared__ double red[2][256];
    const int CW = C < 256 ? C : 256;          // channel lanes
    const int RW = 256 / CW;                   // rows in flight
    const int cq = threadIdx.x % CW, rq = threadIdx.x / CW;
    const int64_t r0 = (int64_t)blockIdx.x * BNACT_ROWS;
    const int64_t r1 = r0 + BNACT_ROWS < M ? r0 + BNACT_ROWS : M;
    for (int c0 = 0; c0 < C; c0 += CW) {        // one pass per channel group (C <= 256: a single one); uniform trip count
        const int c = c0 + cq;
        const bool live = c < C && rq < RW;
        double s1 = 0.0, s2 = 0.0;
        if (live) {
            const float sc = scale ? scale[c] : 1.f, sh = scale ? shift[c] : 0.f;
            const float mu = mean ? mean[c] : 0.f, is = mean ? invstd[c] : 0.f;
            for (int64_t r = r0 + rq; r < r1; r += RW) {
                const size_t o = (size_t)r * C + c;
                const float xv = x[o];
                const float g = ga[o] * act_grad(fmaf(xv, sc, sh), kind);
                gv[o] = g;
                s1 += (double)g;
                s2 += (double)g * (double)((xv - mu) * is);
            }
        }
        if (partial) {
            __syncthreads();
            red[0][threadIdx.x] = s1;
            red[1][threadIdx.x] = s2;
            __syncthreads();
            if (live && rq == 0) {
                for (int j = 1; j < RW; ++j) {   // fixed order over the row lanes
                    s1 += red[0][j * CW + cq];
                    s2 += red[1][j * CW + cq];
                }
                partial[((size_t)0 * C + c) * P + blockIdx.x] = s1;
                partial[((size_t)1 * C + c) * P + blockIdx.x] = s2;
            }
        }
    }
}

extern "C" int otvae_bn_act_bwd(const float* ga, const float* x, const float* scale, const float* shift, const float* mean,
                                const float* invstd, int kind, int64_t M, int C, float* gv, double* partial, void* stream) {
    OTVAE_REQUIRE(ga && x && gv && M > 0 && C > 0, "otvae_bn_act_bwd: bad argument");
    OTVAE_REQUIRE((scale == nullptr) == (shift == nullptr) && (mean == nullptr) == (invstd == nullptr),
                  "otvae_bn_act_bwd: scale/shift and mean/invstd come in pairs");
    OTVAE_REQUIRE((partial == nullptr) == (mean == nullptr), "otvae_bn_act_bwd: the partial sums go with mean / invstd");
    OTVAE_REQUIRE(kind >= ACT_NONE && kind <= ACT_SILU, "otvae_bn_act_bwd: unknown activation %d", kind);
    const int P = otvae_bn_act_bwd_parts(M);
    bn_act_bwd_kernel<<<P, 256, 0, (hipStream_t)stream>>>(ga, x, scale, shift, mean, invstd, kind, M, C, gv, partial, P);
    OTVAE_CHECK_LAUNCH("otvae_bn_act_bwd");
    return OTVAE_OK;
}

__global__ __launch_bounds__(256) void scale_f32_kernel(const float* __restrict__ src, float alpha, int64_t n, float* __restrict__ dst) {
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = alpha * src[i];
}

// dst = alpha * src over n contiguous floats (the weight / bias multipliers of equalized_lr and their gradients)
extern "C" int otvae_scale_f32(const float* src, float alpha, int64_t n, float* dst, void* stream) {
    OTVAE_REQUIRE(src && dst && n > 0, "otvae_scale_f32: bad argument");
    scale_f32_kernel<<<imin(cdiv(n, 256), 2048), 256, 0, (hipStream_t)stream>>>(src, alpha, n, dst);
    OTVAE_CHECK_LAUNCH("otvae_scale_f32");
    return OTVAE_OK;
}
