// Two more options of the reference's ConvLayer (networks/cnn.py:112-118,160-192), unfused around the convolution kernels like
// the non-ReLU activations (functional._conv_layer_general):
//   * FiLM conditioning (`additional_embed`): x * scale[n][c] + bias[n][c] between the normalisation and the activation, scale /
//     bias = two Linear projections of the activated embedding (those run on the library GEMM);
//   * nn.Dropout2d(p) after the convolution: whole (sample, channel) maps are dropped; the mask is a hash of
//     (call key, sample, channel) that the backward recomputes (dropout_hash.h), never stored.
// x, out, g: [N][HW][C] channels-last.
#include "common.h"
#include "dropout_hash.h"

__global__ __launch_bounds__(256) void film_fwd_kernel(const float* __restrict__ x, const float* __restrict__ s, const float* __restrict__ b,
                                                       int64_t total, int HWC, int C, float* __restrict__ out) {
    for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t n = i / HWC;
        const int c = (int)(i % C);
        out[i] = fmaf(x[i], s[n * C + c], b[n * C + c]);
    }
}

// one workgroup per sample: gx = g * s;  gs[n][c] = sum_hw g x;  gb[n][c] = sum_hw g  (fixed order over hw)
__global__ __launch_bounds__(256) void film_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x, const float* __restrict__ s,
                                                       int HW, int C, float* __restrict__ gx, float* __restrict__ gs, float* __restrict__ gb) {
    const int n = blockIdx.x;
    const size_t base = (size_t)n * HW * C;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float sc = s[(size_t)n * C + c];
        double a = 0.0, bsum = 0.0;
        for (int p = 0; p < HW; ++p) {
            const size_t o = base + (size_t)p * C + c;
            const float gg = g[o];
            gx[o] = gg * sc;
            a += (double)gg * (double)x[o];
            bsum += (double)gg;
        }
        gs[(size_t)n * C + c] = (float)a;
        gb[(size_t)n * C + c] = (float)bsum;
    }
}

extern "C" int otvae_film_fwd(const float* x, const float* scale, const float* bias, int N, int HW, int C, float* out, void* stream) {
    OTVAE_REQUIRE(x && scale && bias && out && N > 0 && HW > 0 && C > 0, "otvae_film_fwd: bad argument");
    const int64_t total = (int64_t)N * HW * C;
    film_fwd_kernel<<<imin(cdiv(total, 256), 4096), 256, 0, (hipStream_t)stream>>>(x, scale, bias, total, HW * C, C, out);
    OTVAE_CHECK_LAUNCH("otvae_film_fwd");
    return OTVAE_OK;
}

extern "C" int otvae_film_bwd(const float* g, const float* x, const float* scale, int N, int HW, int C, float* gx, float* gscale,
                              float* gbias, void* stream) {
    OTVAE_REQUIRE(g && x && scale && gx && gscale && gbias && N > 0 && HW > 0 && C > 0, "otvae_film_bwd: bad argument");
    film_bwd_kernel<<<N, 256, 0, (hipStream_t)stream>>>(g, x, scale, HW, C, gx, gscale, gbias);
    OTVAE_CHECK_LAUNCH("otvae_film_bwd");
    return OTVAE_OK;
}

// keep(n, c) = hash(call key, n, c) >= thresh;  y = keep ? x / (1 - p) : 0.  BWD: the key is the one the forward left in `used`.
template <bool BWD>
__global__ __launch_bounds__(256) void dropout2d_kernel(const float* __restrict__ x, int64_t total, int HWC, int C, uint32_t thresh,
                                                        float inv_keep, const int64_t* __restrict__ key, int stream_id,
                                                        int64_t* __restrict__ used, float* __restrict__ out) {
    uint64_t ck;
    if constexpr (BWD) {
        ck = (uint64_t)key[0];
    } else {
        ck = call_key(key, stream_id);
        if (blockIdx.x == 0 && threadIdx.x == 0) used[0] = (int64_t)ck;
    }
    for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const uint32_t n = (uint32_t)(i / HWC);
        const int c = (int)(i % C);
        out[i] = keep_pair(row_hash(ck, n), c, thresh) ? x[i] * inv_keep : 0.f;
    }
}

__global__ __launch_bounds__(256) void dropout2d_mask_kernel(int N, int C, uint32_t thresh, const int64_t* __restrict__ used,
                                                             uint8_t* __restrict__ keep) {
    const uint64_t ck = (uint64_t)used[0];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < N * C) keep[i] = keep_pair(row_hash(ck, (uint32_t)(i / C)), i % C, thresh) ? 1 : 0;
}

extern "C" int otvae_dropout2d_fwd(const float* x, int N, int HW, int C, float p, const int64_t* key, int stream_id, float* y,
                                   int64_t* used, void* stream) {
    OTVAE_REQUIRE(x && y && key && used && N > 0 && HW > 0 && C > 0 && p >= 0.f && p < 1.f && stream_id >= 0 && stream_id < 4095,
                  "otvae_dropout2d_fwd: bad argument");
    const int64_t total = (int64_t)N * HW * C;
    dropout2d_kernel<false><<<imin(cdiv(total, 256), 4096), 256, 0, (hipStream_t)stream>>>(x, total, HW * C, C, dropout_threshold(p),
                                                                                         1.f / (1.f - p), key, stream_id, used, y);
    OTVAE_CHECK_LAUNCH("otvae_dropout2d_fwd");
    return OTVAE_OK;
}

extern "C" int otvae_dropout2d_bwd(const float* gy, int N, int HW, int C, float p, const int64_t* used, float* gx, void* stream) {
    OTVAE_REQUIRE(gy && gx && used && N > 0 && HW > 0 && C > 0 && p >= 0.f && p < 1.f, "otvae_dropout2d_bwd: bad argument");
    const int64_t total = (int64_t)N * HW * C;
    dropout2d_kernel<true><<<imin(cdiv(total, 256), 4096), 256, 0, (hipStream_t)stream>>>(gy, total, HW * C, C, dropout_threshold(p),
                                                                                        1.f / (1.f - p), used, 0, nullptr, gx);
    OTVAE_CHECK_LAUNCH("otvae_dropout2d_bwd");
    return OTVAE_OK;
}

extern "C" int otvae_dropout2d_mask(int N, int C, float p, const int64_t* used, uint8_t* keep, void* stream) {
    OTVAE_REQUIRE(used && keep && N > 0 && C > 0 && p >= 0.f && p < 1.f, "otvae_dropout2d_mask: bad argument");
    dropout2d_mask_kernel<<<cdiv(N * C, 256), 256, 0, (hipStream_t)stream>>>(N, C, dropout_threshold(p), used, keep);
    OTVAE_CHECK_LAUNCH("otvae_dropout2d_mask");
    return OTVAE_OK;
}
