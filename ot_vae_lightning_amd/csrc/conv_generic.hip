// Generic convolution: ANY stride and any square footprint up to 32 x 32 taps -- the fallback behind the tuned kernels, which take
// strides 1 / 2 and footprints up to 7 x 7.  It exists so that every configuration the reference's ConvLayer / CNN can be given runs:
// `down_sample = s` makes a (2 s) x (2 s) kernel with stride s (networks/cnn.py:98-101), e.g. CNN(scaling_factor=4) -> 8 x 8, stride 4
// (get_block_scaling, cnn.py:605-621), which rounds 1-3 refused.  Plain direct convolutions, one thread per result element, fp32
// accumulation in tap-major order (kh, kw, ci) like the tuned kernels; no BatchNorm / activation fusion (the unfused route of
// functional._conv_layer_general runs them around it).  NHWC activations, HWIO weights.  Correct first: these layers are rare, so no
// LDS tiling and no MFMA here.
#include "common.h"

namespace {

struct GG {
    int N, Hs, Ws, Cs, Ho, Wo, Cn, KH, KW, stride, pad;
};

// y[n][oy][ox][co] = bias[co] + sum_{kh,kw,ci} x[n][oy*s + kh - p][ox*s + kw - p][ci] * w[kh][kw][ci][co]
__global__ __launch_bounds__(256) void generic_fwd_kernel(GG g, const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y) {
    const size_t total = (size_t)g.N * g.Ho * g.Wo * g.Cn;
    for (size_t e = blockIdx.x * (size_t)256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int co = (int)(e % g.Cn);
        size_t r = e / g.Cn;
        const int ox = (int)(r % g.Wo);
        r /= g.Wo;
        const int oy = (int)(r % g.Ho);
        const int n = (int)(r / g.Ho);
        float acc = 0.f;
        for (int kh = 0; kh < g.KH; ++kh) {
            const int iy = oy * g.stride + kh - g.pad;
            if (iy < 0 || iy >= g.Hs) continue;
            for (int kw = 0; kw < g.KW; ++kw) {
                const int ix = ox * g.stride + kw - g.pad;
                if (ix < 0 || ix >= g.Ws) continue;
                const float* xp = x + (((size_t)n * g.Hs + iy) * g.Ws + ix) * g.Cs;
                const float* wp = w + ((size_t)(kh * g.KW + kw) * g.Cs) * g.Cn + co;
                for (int ci = 0; ci < g.Cs; ++ci) acc = fmaf(xp[ci], wp[(size_t)ci * g.Cn], acc);
            }
        }
        y[e] = bias ? acc + bias[co] : acc;
    }
}

// gx[n][iy][ix][ci] = sum_{kh,kw : (iy + p - kh) % s == 0} sum_co gy[n][(iy+p-kh)/s][(ix+p-kw)/s][co] * w[kh][kw][ci][co]
__global__ __launch_bounds__(256) void generic_dgrad_kernel(GG g, const float* __restrict__ gy, const float* __restrict__ w,
                                                            float* __restrict__ gx) {
    const size_t total = (size_t)g.N * g.Hs * g.Ws * g.Cs;
    for (size_t e = blockIdx.x * (size_t)256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int ci = (int)(e % g.Cs);
        size_t r = e / g.Cs;
        const int ix = (int)(r % g.Ws);
        r /= g.Ws;
        const int iy = (int)(r % g.Hs);
        const int n = (int)(r / g.Hs);
        float acc = 0.f;
        for (int kh = 0; kh < g.KH; ++kh) {
            const int ty = iy + g.pad - kh;
            if (ty < 0 || ty % g.stride != 0) continue;
            const int oy = ty / g.stride;
            if (oy >= g.Ho) continue;
            for (int kw = 0; kw < g.KW; ++kw) {
                const int tx = ix + g.pad - kw;
                if (tx < 0 || tx % g.stride != 0) continue;
                const int ox = tx / g.stride;
                if (ox >= g.Wo) continue;
                const float* gp = gy + (((size_t)n * g.Ho + oy) * g.Wo + ox) * g.Cn;
                const float* wp = w + ((size_t)(kh * g.KW + kw) * g.Cs + ci) * g.Cn;
                for (int co = 0; co < g.Cn; ++co) acc = fmaf(gp[co], wp[co], acc);
            }
        }
        gx[e] = acc;
    }
}

// partial[p][k][co], k = (kh*KW + kw)*Cs + ci (+ one extra row K for the bias: sum of gy): block (k, p) sums the output positions
// of chunk p in increasing order; threads run along co.  Reduced over p by generic_reduce_kernel in a fixed order.
__global__ __launch_bounds__(256) void generic_wgrad_kernel(GG g, const float* __restrict__ x, const float* __restrict__ gy, int has_bias,
                                                            int P, float* __restrict__ partial) {
    const int K = g.KH * g.KW * g.Cs, Kp = K + (has_bias ? 1 : 0);
    const int k = blockIdx.x, p = blockIdx.y;
    const size_t M = (size_t)g.N * g.Ho * g.Wo;
    const size_t per = (M + P - 1) / P, m0 = (size_t)p * per, m1 = m0 + per < M ? m0 + per : M;
    const bool is_bias = k == K;
    int kh = 0, kw = 0, ci = 0;
    if (!is_bias) {
        ci = k % g.Cs;
        const int t = k / g.Cs;
        kw = t % g.KW;
        kh = t / g.KW;
    }
    for (int co = threadIdx.x; co < g.Cn; co += 256) {
        float acc = 0.f;
        for (size_t m = m0; m < m1; ++m) {
            const int ox = (int)(m % g.Wo);
            const size_t r = m / g.Wo;
            const int oy = (int)(r % g.Ho), n = (int)(r / g.Ho);
            float xv = 1.f;
            if (!is_bias) {
                const int iy = oy * g.stride + kh - g.pad, ix = ox * g.stride + kw - g.pad;
                if (iy < 0 || iy >= g.Hs || ix < 0 || ix >= g.Ws) continue;
                xv = x[(((size_t)n * g.Hs + iy) * g.Ws + ix) * g.Cs + ci];
            }
            acc = fmaf(xv, gy[m * g.Cn + co], acc);
        }
        partial[((size_t)p * Kp + k) * g.Cn + co] = acc;
    }
}

__global__ __launch_bounds__(256) void generic_reduce_kernel(const float* __restrict__ partial, int P, int K, int Kp, int Cn,
                                                             float* __restrict__ gw, float* __restrict__ gb) {
    const size_t total = (size_t)Kp * Cn;
    for (size_t e = blockIdx.x * (size_t)256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        double s = 0.0;
        for (int p = 0; p < P; ++p) s += (double)partial[(size_t)p * total + e];
        const int k = (int)(e / Cn);
        if (k < K) gw[e] = (float)s;
        else if (gb) gb[e - (size_t)K * Cn] = (float)s;
    }
}

int check_geom(const char* who, const otvae_conv_geom* g, GG* out) {
    OTVAE_REQUIRE(g, "%s: NULL geometry", who);
    OTVAE_REQUIRE(g->N > 0 && g->Hs > 0 && g->Ws > 0 && g->Cs > 0 && g->Cn > 0 && g->Ho > 0 && g->Wo > 0, "%s: bad sizes", who);
    OTVAE_REQUIRE(g->up == 1, "%s: the generic convolution takes no fused up-sampling (up must be 1)", who);
    OTVAE_REQUIRE(g->KH >= 1 && g->KH <= 32 && g->KW >= 1 && g->KW <= 32, "%s: footprint %d x %d outside 1..32", who, g->KH, g->KW);
    OTVAE_REQUIRE(g->stride >= 1 && g->pad >= 0, "%s: bad stride / padding", who);
    OTVAE_REQUIRE(g->Ho == (g->Hs + 2 * g->pad - g->KH) / g->stride + 1 && g->Wo == (g->Ws + 2 * g->pad - g->KW) / g->stride + 1,
                  "%s: output size %d x %d does not match the geometry", who, g->Ho, g->Wo);
    *out = GG{g->N, g->Hs, g->Ws, g->Cs, g->Ho, g->Wo, g->Cn, g->KH, g->KW, g->stride, g->pad};
    return OTVAE_OK;
}

}  // namespace

extern "C" int otvae_conv_generic_fwd(const otvae_conv_geom* geom, const float* x, const float* w_hwio, const float* bias, float* y,
                                      void* stream) {
    GG g;
    if (int rc = check_geom("otvae_conv_generic_fwd", geom, &g)) return rc;
    OTVAE_REQUIRE(x && w_hwio && y, "otvae_conv_generic_fwd: NULL tensor");
    const size_t total = (size_t)g.N * g.Ho * g.Wo * g.Cn;
    generic_fwd_kernel<<<imin(cdiv((int64_t)total, 256), 8192), 256, 0, (hipStream_t)stream>>>(g, x, w_hwio, bias, y);
    OTVAE_CHECK_LAUNCH("otvae_conv_generic_fwd");
    return OTVAE_OK;
}

extern "C" int otvae_conv_generic_bwd_data(const otvae_conv_geom* geom, const float* gy, const float* w_hwio, float* gx, void* stream) {
    GG g;
    if (int rc = check_geom("otvae_conv_generic_bwd_data", geom, &g)) return rc;
    OTVAE_REQUIRE(gy && w_hwio && gx, "otvae_conv_generic_bwd_data: NULL tensor");
    const size_t total = (size_t)g.N * g.Hs * g.Ws * g.Cs;
    generic_dgrad_kernel<<<imin(cdiv((int64_t)total, 256), 8192), 256, 0, (hipStream_t)stream>>>(g, gy, w_hwio, gx);
    OTVAE_CHECK_LAUNCH("otvae_conv_generic_bwd_data");
    return OTVAE_OK;
}

// workspace floats for otvae_conv_generic_bwd_weight
extern "C" int64_t otvae_conv_generic_bwd_weight_ws(const otvae_conv_geom* geom, int has_bias) {
    GG g;
    if (check_geom("otvae_conv_generic_bwd_weight_ws", geom, &g)) return -1;
    const int64_t Kp = (int64_t)g.KH * g.KW * g.Cs + (has_bias ? 1 : 0);
    const int64_t M = (int64_t)g.N * g.Ho * g.Wo;
    const int P = (int)(M < 64 ? M : 64);
    return (int64_t)P * Kp * g.Cn;
}

extern "C" int otvae_conv_generic_bwd_weight(const otvae_conv_geom* geom, const float* x, const float* gy, int has_bias, float* ws,
                                             float* gw_hwio, float* gb, void* stream) {
    GG g;
    if (int rc = check_geom("otvae_conv_generic_bwd_weight", geom, &g)) return rc;
    OTVAE_REQUIRE(x && gy && ws && gw_hwio && (!has_bias || gb), "otvae_conv_generic_bwd_weight: NULL tensor");
    const int K = g.KH * g.KW * g.Cs, Kp = K + (has_bias ? 1 : 0);
    const int64_t M = (int64_t)g.N * g.Ho * g.Wo;
    const int P = (int)(M < 64 ? M : 64);
    OTVAE_REQUIRE(Kp <= 65535 * 16, "otvae_conv_generic_bwd_weight: too many weight rows");
    hipStream_t st = (hipStream_t)stream;
    generic_wgrad_kernel<<<dim3(Kp, P), 256, 0, st>>>(g, x, gy, has_bias, P, ws);
    OTVAE_CHECK_LAUNCH("otvae_conv_generic_bwd_weight");
    generic_reduce_kernel<<<imin(cdiv((int64_t)Kp * g.Cn, 256), 4096), 256, 0, st>>>(ws, P, K, Kp, g.Cn, gw_hwio, gb);
    OTVAE_CHECK_LAUNCH("otvae_conv_generic_bwd_weight(reduce)");
    return OTVAE_OK;
}
