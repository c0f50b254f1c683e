// GroupNorm / InstanceNorm2d in front of a ConvLayer's activation (reference networks/cnn.py:121-125:
// nn.GroupNorm(div_sqrt(C // groups), C) and nn.InstanceNorm2d(C) -- the latter = one group per channel, no affine parameters):
// statistics per (sample, group) over the group's channels and all positions, so -- unlike training-mode BatchNorm -- no
// reduction over the batch: one workgroup per (sample, group) does statistics, normalisation, affine and activation in one launch,
// and the whole backward of that chain in another.  x, out, ga, dx: [N][HW][C] channels-last; the activation codes are those of
// csrc/activation.hip.  Used unfused around the convolution kernels (functional._conv_layer_general), like the non-ReLU activations.
#include "common.h"

#define SELU_ALPHA 1.6732632423543772848170429916717f
#define SELU_SCALE 1.0507009873554804934193349852946f

__device__ __forceinline__ float gn_act(float u, int kind) {
    switch (kind) {
        case 1: return fmaxf(u, 0.f);
        case 2: return u > 0.f ? u : 0.2f * u;
        case 3: return SELU_SCALE * (u > 0.f ? u : SELU_ALPHA * expm1f(u));
        case 4: return 0.5f * u * (1.f + erff(u * 0.70710678118654752440f));
        case 5: return u / (1.f + expf(-u));
        default: return u;
    }
}
__device__ __forceinline__ float gn_act_grad(float u, int kind) {
    switch (kind) {
        case 1: return u > 0.f ? 1.f : 0.f;
        case 2: return u > 0.f ? 1.f : 0.2f;
        case 3: return u > 0.f ? SELU_SCALE : SELU_SCALE * SELU_ALPHA * expf(u);
        case 4: return 0.5f * (1.f + erff(u * 0.70710678118654752440f)) + u * 0.39894228040143267794f * expf(-0.5f * u * u);
        case 5: {
            const float s = 1.f / (1.f + expf(-u));
            return s * (1.f + u * (1.f - s));
        }
        default: return 1.f;
    }
}

__device__ __forceinline__ double block_sum256(double v, double* red) {  // all 256 threads; result in every thread
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// grid (G, N).  element e of the group: position e / cg, channel g*cg + e % cg
__global__ __launch_bounds__(256) void group_norm_act_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, int HW, int C, int G, float eps, int kind,
                                                                 float* __restrict__ out, float* __restrict__ mean, float* __restrict__ rstd) {
    __shared__ double red[4];
    const int g = blockIdx.x, n = blockIdx.y, cg = C / G;
    const float* xb = x + (size_t)n * HW * C + g * cg;
    float* ob = out + (size_t)n * HW * C + g * cg;
    const int cnt = HW * cg;
    double s = 0.0;
    for (int e = threadIdx.x; e < cnt; e += 256) s += (double)xb[(size_t)(e / cg) * C + e % cg];
    const double mu = block_sum256(s, red) / (double)cnt;
    double q = 0.0;
    for (int e = threadIdx.x; e < cnt; e += 256) {
        const double d = (double)xb[(size_t)(e / cg) * C + e % cg] - mu;
        q += d * d;
    }
    const double var = block_sum256(q, red) / (double)cnt;   // biased, like nn.GroupNorm / nn.InstanceNorm2d
    const float fmu = (float)mu, frs = (float)(1.0 / sqrt(var + (double)eps));
    if (threadIdx.x == 0) {
        mean[(size_t)n * G + g] = fmu;
        rstd[(size_t)n * G + g] = frs;
    }
    for (int e = threadIdx.x; e < cnt; e += 256) {
        const int c = e % cg;
        const size_t o = (size_t)(e / cg) * C + c;
        float u = (xb[o] - fmu) * frs;
        if (gamma) u = fmaf(u, gamma[g * cg + c], beta[g * cg + c]);
        ob[o] = gn_act(u, kind);
    }
}

// backward of out = act(xhat * gamma + beta): gu = ga act'(u);  per (n, c): pg = sum_hw gu xhat, pb = sum_hw gu (partial parameter
// gradients, summed over n by the caller);  dx = rstd (gamma gu - mean_grp(gamma gu) - xhat mean_grp(gamma gu xhat))
__global__ __launch_bounds__(256) void group_norm_act_bwd_kernel(const float* __restrict__ ga, const float* __restrict__ x,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 const float* __restrict__ mean, const float* __restrict__ rstd, int HW, int C,
                                                                 int G, int kind, float* __restrict__ dx, float* __restrict__ pgamma,
                                                                 float* __restrict__ pbeta) {
    __shared__ double red[4];
    const int g = blockIdx.x, n = blockIdx.y, cg = C / G;
    const size_t base = (size_t)n * HW * C + g * cg;
    const int cnt = HW * cg;
    const float mu = mean[(size_t)n * G + g], rs = rstd[(size_t)n * G + g];
    double s1 = 0.0, s2 = 0.0;
    for (int e = threadIdx.x; e < cnt; e += 256) {
        const int c = e % cg;
        const size_t o = base + (size_t)(e / cg) * C + c;
        const float xh = (x[o] - mu) * rs;
        const float gm = gamma ? gamma[g * cg + c] : 1.f;
        const float u = gamma ? fmaf(xh, gm, beta[g * cg + c]) : xh;
        const float w = ga[o] * gn_act_grad(u, kind) * gm;
        s1 += (double)w;
        s2 += (double)w * (double)xh;
    }
    const float m1 = (float)(block_sum256(s1, red) / (double)cnt);
    const float m2 = (float)(block_sum256(s2, red) / (double)cnt);
    for (int e = threadIdx.x; e < cnt; e += 256) {
        const int c = e % cg;
        const size_t o = base + (size_t)(e / cg) * C + c;
        const float xh = (x[o] - mu) * rs;
        const float gm = gamma ? gamma[g * cg + c] : 1.f;
        const float u = gamma ? fmaf(xh, gm, beta[g * cg + c]) : xh;
        const float w = ga[o] * gn_act_grad(u, kind) * gm;
        dx[o] = rs * (w - m1 - xh * m2);
    }
    if (gamma) {  // per-channel parameter partials of this sample: channel c by the threads t == c (mod cg), fixed order over t
        for (int c = threadIdx.x; c < cg; c += 256) {
            double a = 0.0, b = 0.0;
            for (int p = 0; p < HW; ++p) {
                const size_t o = base + (size_t)p * C + c;
                const float xh = (x[o] - mu) * rs;
                const float gu = ga[o] * gn_act_grad(fmaf(xh, gamma[g * cg + c], beta[g * cg + c]), kind);
                a += (double)gu * (double)xh;
                b += (double)gu;
            }
            pgamma[(size_t)n * C + g * cg + c] = (float)a;
            pbeta[(size_t)n * C + g * cg + c] = (float)b;
        }
    }
}

extern "C" int otvae_group_norm_act_fwd(const float* x, const float* gamma, const float* beta, int N, int HW, int C, int G, float eps,
                                        int kind, float* out, float* mean, float* rstd, void* stream) {
    OTVAE_REQUIRE(x && out && mean && rstd && N > 0 && HW > 0 && C > 0 && G > 0 && C % G == 0, "otvae_group_norm_act_fwd: bad argument");
    OTVAE_REQUIRE((gamma == nullptr) == (beta == nullptr) && kind >= 0 && kind <= 5, "otvae_group_norm_act_fwd: bad argument");
    group_norm_act_fwd_kernel<<<dim3(G, N), 256, 0, (hipStream_t)stream>>>(x, gamma, beta, HW, C, G, eps, kind, out, mean, rstd);
    OTVAE_CHECK_LAUNCH("otvae_group_norm_act_fwd");
    return OTVAE_OK;
}

extern "C" int otvae_group_norm_act_bwd(const float* ga, const float* x, const float* gamma, const float* beta, const float* mean,
                                        const float* rstd, int N, int HW, int C, int G, int kind, float* dx, float* pgamma, float* pbeta,
                                        void* stream) {
    OTVAE_REQUIRE(ga && x && mean && rstd && dx && N > 0 && HW > 0 && C > 0 && G > 0 && C % G == 0, "otvae_group_norm_act_bwd: bad argument");
    OTVAE_REQUIRE((gamma == nullptr) == (beta == nullptr) && (gamma == nullptr || (pgamma && pbeta)) && kind >= 0 && kind <= 5,
                  "otvae_group_norm_act_bwd: bad argument");
    group_norm_act_bwd_kernel<<<dim3(G, N), 256, 0, (hipStream_t)stream>>>(ga, x, gamma, beta, mean, rstd, HW, C, G, kind, dx, pgamma, pbeta);
    OTVAE_CHECK_LAUNCH("otvae_group_norm_act_bwd");
    return OTVAE_OK;
}

// dst[c] = sum_r src[r][c] in a fixed order: 64 columns x 4 row lanes per block, a row lane adds rows q, q + 4, ... in fp64, the four
// lanes are combined in lane order through LDS.  (d gamma / d beta of the per-sample parameter partials above; the batch sum behind
// a parameter that was broadcast over the batch: the ViT's position embeddings and learned tokens.)
__global__ __launch_bounds__(256) void colsum_f32_kernel(const float* __restrict__ src, int R, int C, float* __restrict__ dst) {
    __shared__ double part[4][64];
    const int cl = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    double s = 0.0;
    if (c < C) {
#pragma unroll 8
        for (int r = q; r < R; r += 4) s += (double)src[(size_t)r * C + c];
    }
    part[q][cl] = s;
    __syncthreads();
    if (q == 0 && c < C) dst[c] = (float)((part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]));
}

// gw[k][:] = sum_{b : idx[b] == k} g[b][:], rows added in increasing b (fixed order): the backward pass of an embedding lookup
// (nn.Embedding: the ViT's class token, networks/vit.py:167,203; ConditionalGaussianPrior's class rows, prior/conditional_gaussian.py:76-77)
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const float* __restrict__ g, const int64_t* __restrict__ idx, int B, int d,
                                                            float* __restrict__ gw) {
    // 64 columns x 4 sample lanes per block; a lane visits samples q, q + 4, ... (loads issued unconditionally, selected afterwards, so
    // that they pipeline), the four lanes are combined in lane order: a fixed order of additions per (row, column)
    __shared__ float part[4][64];
    const int k = blockIdx.x, cl = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int j = blockIdx.y * 64 + cl;
    float acc = 0.f;
    if (j < d) {
#pragma unroll 8
        for (int b = q; b < B; b += 4) {
            const float v = g[(size_t)b * d + j];
            acc += idx[b] == (int64_t)k ? v : 0.f;
        }
    }
    part[q][cl] = acc;
    __syncthreads();
    if (q == 0 && j < d) gw[(size_t)k * d + j] = (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
}

extern "C" int otvae_embedding_bwd(const float* g, const int64_t* idx, int B, int K, int d, float* gw, void* stream) {
    OTVAE_REQUIRE(g && idx && gw && B > 0 && K > 0 && d > 0, "otvae_embedding_bwd: bad argument");
    embedding_bwd_kernel<<<dim3(K, cdiv(d, 64)), 256, 0, (hipStream_t)stream>>>(g, idx, B, d, gw);
    OTVAE_CHECK_LAUNCH("otvae_embedding_bwd");
    return OTVAE_OK;
}

extern "C" int otvae_colsum_f32(const float* src, int R, int C, float* dst, void* stream) {
    OTVAE_REQUIRE(src && dst && R > 0 && C > 0, "otvae_colsum_f32: bad argument");
    colsum_f32_kernel<<<cdiv(C, 64), 256, 0, (hipStream_t)stream>>>(src, R, C, dst);
    OTVAE_CHECK_LAUNCH("otvae_colsum_f32");
    return OTVAE_OK;
}
