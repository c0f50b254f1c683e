// LayerNorm over the last dimension, forward and backward, for the token streams of the ViT (reference networks/vit.py:
// PositionalEmbedding.LayerNorm :38,54 and the two norms of every nn.TransformerEncoderLayer :169-172; arithmetic of
// torch.nn.functional.layer_norm: biased variance, eps inside the square root).
//   y = (x + res - mean) * rstd * gamma + beta        (res: optional residual summed in, the post-norm block's "x + sublayer")
// One wave per row (D <= 4096): the row lives in registers (D/64 values per lane), mean and centred variance are two
// wave reductions.  Backward: dx = rstd * (g*gamma - mean(g*gamma) - xhat * mean(g*gamma*xhat)); the parameter gradients are
// column sums over all rows -- per-block partials in fixed order, then one fixed-order reduction (deterministic).
#include "common.h"

#define LN_MAXV 32  // values per lane (template parameter NV = 1, 2, 4 ... 32): D <= 2048
// values per lane as a compile-time constant (the row must stay in registers: a runtime-indexed array would go to scratch)
#define LN_NV_SWITCH(D_, MACRO)          \
    {                                    \
        const int nv_ = ((D_) + 63) >> 6; \
        if (nv_ <= 1) MACRO(1);           \
        else if (nv_ <= 2) MACRO(2);      \
        else if (nv_ <= 4) MACRO(4);      \
        else if (nv_ <= 8) MACRO(8);      \
        else if (nv_ <= 16) MACRO(16);    \
        else MACRO(32);                   \
    }


template <int NV>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, int M,
                                                            int D, float eps, float* __restrict__ sum_out, float* __restrict__ y,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (size_t)row * D;
    const float* rr = res ? res + (size_t)row * D : nullptr;
    float v[NV];
    float s = 0.f;
    constexpr int nv = NV;
#pragma unroll
    for (int k = 0; k < nv; ++k) {
        const int c = lane + 64 * k;
        float t = c < D ? xr[c] : 0.f;
        if (rr && c < D) t += rr[c];
        v[k] = t;
        s += t;
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < nv; ++k) {
        const float d = (lane + 64 * k < D) ? v[k] - mean : 0.f;
        q += d * d;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int k = 0; k < nv; ++k) {
        const int c = lane + 64 * k;
        if (c < D) {
            if (sum_out) sum_out[(size_t)row * D + c] = v[k];  // x + res: what backward normalises again
            y[(size_t)row * D + c] = (v[k] - mean) * rstd * gamma[c] + beta[c];
        }
    }
    if (lane == 0) {
        mean_out[row] = mean;
        rstd_out[row] = rstd;
    }
}

extern "C" int otvae_layernorm_fwd(const float* x, const float* res, const float* gamma, const float* beta, int M, int D, float eps,
                                   float* sum_out, float* y, float* mean, float* rstd, void* stream) {
    OTVAE_REQUIRE(x && gamma && beta && y && mean && rstd && M > 0 && D > 0, "otvae_layernorm_fwd: bad argument");
    if (D > 64 * LN_MAXV) {
        otvae_set_error("otvae_layernorm_fwd: D = %d unsupported (D <= %d)", D, 64 * LN_MAXV);
        return OTVAE_EUNSUPPORTED;
    }
    OTVAE_REQUIRE(!res || sum_out, "otvae_layernorm_fwd: a residual needs sum_out (x + res is what backward reads)");
    hipStream_t st = (hipStream_t)stream;
#define LN_FWD(NV_) layernorm_fwd_kernel<NV_><<<cdiv(M, 4), 256, 0, st>>>(x, res, gamma, beta, M, D, eps, sum_out, y, mean, rstd)
    LN_NV_SWITCH(D, LN_FWD)
#undef LN_FWD
    OTVAE_CHECK_LAUNCH("otvae_layernorm_fwd");
    return OTVAE_OK;
}

#define LN_ROWS_PER_BLOCK 16  // rows a block of the backward kernel walks (4 per wave): enough blocks to fill the chip at ~5k rows

// partial[block][2][D]: column sums of g * xhat (d gamma) and of g (d beta) over the block's rows, waves combined in order
template <int NV>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ xs, const float* __restrict__ gy,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, int M, int D, float* __restrict__ gx,
                                                            float* __restrict__ partial) {
    extern __shared__ float ln_red[];  // [4 waves][2][D]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int nv = NV;
    float dg[NV], db[NV];
#pragma unroll
    for (int k = 0; k < nv; ++k) dg[k] = db[k] = 0.f;
    const int row0 = blockIdx.x * LN_ROWS_PER_BLOCK;
    for (int i = wave; i < LN_ROWS_PER_BLOCK; i += 4) {  // fixed row order per wave
        const int row = row0 + i;
        if (row >= M) break;
        const float mu = mean[row], rs = rstd[row];
        float gg[NV], xh[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < nv; ++k) {
            const int c = lane + 64 * k;
            const bool in = c < D;
            const float g = in ? gy[(size_t)row * D + c] : 0.f;
            xh[k] = in ? (xs[(size_t)row * D + c] - mu) * rs : 0.f;
            gg[k] = in ? g * gamma[c] : 0.f;
            s1 += gg[k];
            s2 += gg[k] * xh[k];
            dg[k] += g * xh[k];
            db[k] += g;
        }
        s1 = wave_sum(s1) / (float)D;
        s2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int k = 0; k < nv; ++k) {
            const int c = lane + 64 * k;
            if (c < D) gx[(size_t)row * D + c] = rs * (gg[k] - s1 - xh[k] * s2);
        }
    }
#pragma unroll
    for (int k = 0; k < nv; ++k) {
        const int c = lane + 64 * k;
        if (c < D) {
            ln_red[(wave * 2 + 0) * D + c] = dg[k];
            ln_red[(wave * 2 + 1) * D + c] = db[k];
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * D; e += 256) {
        const int which = e / D, c = e - which * D;
        partial[((size_t)blockIdx.x * 2 + which) * D + c] =
            (ln_red[(0 * 2 + which) * D + c] + ln_red[(1 * 2 + which) * D + c]) +
            (ln_red[(2 * 2 + which) * D + c] + ln_red[(3 * 2 + which) * D + c]);
    }
}

// dgamma / dbeta = column sums of the per-block partials.  A block owns 16 columns of one of the two vectors; its 16
// partial-lanes each sum every 16th partial in increasing order, then lane 0 adds the 16 lane sums in order: a fixed
// summation tree (deterministic) with P/16 dependent loads instead of P.
__global__ __launch_bounds__(256) void layernorm_param_reduce_kernel(const float* __restrict__ partial, int P, int D,
                                                                     float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float red[16][17];
    const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + cl;  // column index over [dgamma | dbeta]
    float s = 0.f;
    if (e < 2 * D) {
        const int which = e / D, c = e - which * D;
        for (int p = pl; p < P; p += 16) s += partial[((size_t)p * 2 + which) * D + c];
    }
    red[pl][cl] = s;
    __syncthreads();
    if (pl == 0 && e < 2 * D) {
        float t = red[0][cl];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += red[k][cl];
        const int which = e / D, c = e - which * D;
        (which == 0 ? dgamma : dbeta)[c] = t;
    }
}

extern "C" int otvae_layernorm_bwd_ws(int M, int D) {
    if (M <= 0 || D <= 0) return -1;
    return cdiv(M, LN_ROWS_PER_BLOCK) * 2 * D;  // floats
}

extern "C" int otvae_layernorm_bwd(const float* xs, const float* gy, const float* gamma, const float* mean, const float* rstd, int M,
                                   int D, float* gx, float* dgamma, float* dbeta, float* ws, void* stream) {
    OTVAE_REQUIRE(xs && gy && gamma && mean && rstd && gx && dgamma && dbeta && ws && M > 0 && D > 0,
                  "otvae_layernorm_bwd: bad argument");
    if (D > 64 * LN_MAXV || (size_t)8 * D * sizeof(float) > 64 * 1024) {
        otvae_set_error("otvae_layernorm_bwd: D = %d unsupported (D <= 2048)", D);
        return OTVAE_EUNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    const int P = cdiv(M, LN_ROWS_PER_BLOCK);
#define LN_BWD(NV_) layernorm_bwd_kernel<NV_><<<P, 256, (size_t)8 * D * sizeof(float), st>>>(xs, gy, gamma, mean, rstd, M, D, gx, ws)
    LN_NV_SWITCH(D, LN_BWD)
#undef LN_BWD
    OTVAE_CHECK_LAUNCH("otvae_layernorm_bwd");
    layernorm_param_reduce_kernel<<<cdiv(2 * D, 16), 256, 0, st>>>(ws, P, D, dgamma, dbeta);
    OTVAE_CHECK_LAUNCH("otvae_layernorm_bwd(reduce)");
    return OTVAE_OK;
}
