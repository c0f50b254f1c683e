// LayerNorm over the last dimension, forward and backward, for the token streams of the ViT (reference networks/vit.py:
// PositionalEmbedding.LayerNorm :38,54 and the two norms of every nn.TransformerEncoderLayer :169-172; arithmetic of
// torch.nn.functional.layer_norm: biased variance, eps inside the square root).
//   y = (x + res - mean) * rstd * gamma + beta        (res: optional residual summed in, the post-norm block's "x + sublayer")
// One wave per row (D <= 4096): the row lives in registers (D/64 values per lane), mean and centred variance are two
// wave reductions.  Backward: dx = rstd * (g*gamma - mean(g*gamma) - xhat * mean(g*gamma*xhat)); the parameter gradients are
// column sums over all rows -- per-block partials in fixed order, then one fixed-order reduction (deterministic).
#include "common.h"
#include "dropout_hash.h"

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


// DROP: the sublayer output x is thinned before the residual is summed in, s = res + x o keep / (1-p) -- the
// "x + dropout(sublayer(x))" of a training-mode nn.TransformerEncoderLayer; keep(row, col) is the hash of dropout_hash.h
struct LnDrop {
    uint32_t thresh;
    float inv_keep;
    const int64_t* key;  // forward: {seed, counter}; backward: the call key the forward left in `used`
    int stream_id;
    int64_t* used;
};

template <int NV, bool DROP>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, int M,
                                                            int D, float eps, float* __restrict__ sum_out, float* __restrict__ y,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out, LnDrop dr) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    uint32_t rh = 0;
    if constexpr (DROP) {
        const uint64_t ck = call_key(dr.key, dr.stream_id);
        if (blockIdx.x == 0 && threadIdx.x == 0) dr.used[0] = (int64_t)ck;
        rh = row_hash(ck, (uint32_t)row);
    }
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
        if constexpr (DROP) t = keep_pair(rh, c, dr.thresh) ? t * dr.inv_keep : 0.f;
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
#define LN_FWD(NV_) layernorm_fwd_kernel<NV_, false><<<cdiv(M, 4), 256, 0, st>>>(x, res, gamma, beta, M, D, eps, sum_out, y, mean, rstd, LnDrop{})
    LN_NV_SWITCH(D, LN_FWD)
#undef LN_FWD
    OTVAE_CHECK_LAUNCH("otvae_layernorm_fwd");
    return OTVAE_OK;
}

extern "C" int otvae_layernorm_dropout_fwd(const float* x, const float* res, const float* gamma, const float* beta, int M, int D,
                                           float eps, float p, const int64_t* key, int stream_id, float* sum_out, float* y, float* mean,
                                           float* rstd, int64_t* used, void* stream) {
    OTVAE_REQUIRE(x && res && gamma && beta && sum_out && y && mean && rstd && key && used && M > 0 && D > 0,
                  "otvae_layernorm_dropout_fwd: bad argument (the residual and sum_out are required)");
    OTVAE_REQUIRE(p >= 0.f && p < 1.f && stream_id >= 0 && stream_id < 4095, "otvae_layernorm_dropout_fwd: bad p or stream_id");
    if (D > 64 * LN_MAXV) {
        otvae_set_error("otvae_layernorm_dropout_fwd: D = %d unsupported (D <= %d)", D, 64 * LN_MAXV);
        return OTVAE_EUNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    const LnDrop dr = {dropout_threshold(p), 1.f / (1.f - p), key, stream_id, used};
#define LN_FWD(NV_) layernorm_fwd_kernel<NV_, true><<<cdiv(M, 4), 256, 0, st>>>(x, res, gamma, beta, M, D, eps, sum_out, y, mean, rstd, dr)
    LN_NV_SWITCH(D, LN_FWD)
#undef LN_FWD
    OTVAE_CHECK_LAUNCH("otvae_layernorm_dropout_fwd");
    return OTVAE_OK;
}

#define LN_ROWS_PER_BLOCK 16  // rows a block of the backward kernel walks (4 per wave): enough blocks to fill the chip at ~5k rows

// partial[block][2][D]: column sums of g * xhat (d gamma) and of g (d beta) over the block's rows, waves combined in order
// DROP: gx is the gradient of the summed row (= of the residual); gxd = gx o keep / (1-p) is that of the thinned operand
template <int NV, bool DROP>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ xs, const float* __restrict__ gy,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, int M, int D, float* __restrict__ gx,
                                                            double* __restrict__ partial, float* __restrict__ gxd, LnDrop dr) {
    // The parameter gradients are sums over ALL rows of products that largely cancel (measured at the vit.yaml shape, round 4: an fp32
    // accumulation sat 4.5e-3 in relative L2 from the float64 truth where the reference's own fp32 arithmetic is at 2.9e-4): they are
    // accumulated in fp64 from the first product to the final column sum (fp32 products, fp64 adds; 2 NV more registers a lane).
    extern __shared__ double ln_red[];  // [2][D], the waves add into it in wave order
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int nv = NV;
    double dg[NV], db[NV];
#pragma unroll
    for (int k = 0; k < nv; ++k) dg[k] = db[k] = 0.0;
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
            dg[k] += (double)(g * xh[k]);
            db[k] += (double)g;
        }
        s1 = wave_sum(s1) / (float)D;
        s2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int k = 0; k < nv; ++k) {
            const int c = lane + 64 * k;
            if (c < D) {
                const float d = rs * (gg[k] - s1 - xh[k] * s2);
                gx[(size_t)row * D + c] = d;
                if constexpr (DROP)
                    gxd[(size_t)row * D + c] =
                        keep_pair(row_hash((uint64_t)dr.key[0], (uint32_t)row), c, dr.thresh) ? d * dr.inv_keep : 0.f;
            }
        }
    }
    for (int w = 0; w < 4; ++w) {   // waves 0..3 in turn: a fixed summation order
        if (wave == w) {
#pragma unroll
            for (int k = 0; k < nv; ++k) {
                const int c = lane + 64 * k;
                if (c < D) {
                    ln_red[c] = (w == 0 ? 0.0 : ln_red[c]) + dg[k];
                    ln_red[D + c] = (w == 0 ? 0.0 : ln_red[D + c]) + db[k];
                }
            }
        }
        __syncthreads();
    }
    for (int e = threadIdx.x; e < 2 * D; e += 256) partial[(size_t)blockIdx.x * 2 * D + e] = ln_red[e];
}

// dgamma / dbeta = column sums of the per-block partials.  A block owns 16 columns of one of the two vectors; its 16
// partial-lanes each sum every 16th partial in increasing order, then lane 0 adds the 16 lane sums in order: a fixed
// summation tree (deterministic) with P/16 dependent loads instead of P.
__global__ __launch_bounds__(256) void layernorm_param_reduce_kernel(const double* __restrict__ partial, int P, int D,
                                                                     float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ double red[16][17];
    const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + cl;  // column index over [dgamma | dbeta]
    double s = 0.0;
    if (e < 2 * D) {
        const int which = e / D, c = e - which * D;
        for (int p = pl; p < P; p += 16) s += partial[((size_t)p * 2 + which) * D + c];
    }
    red[pl][cl] = s;
    __syncthreads();
    if (pl == 0 && e < 2 * D) {
        double t = red[0][cl];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += red[k][cl];
        const int which = e / D, c = e - which * D;
        (which == 0 ? dgamma : dbeta)[c] = (float)t;
    }
}

extern "C" int otvae_layernorm_bwd_ws(int M, int D) {
    if (M <= 0 || D <= 0) return -1;
    return cdiv(M, LN_ROWS_PER_BLOCK) * 2 * D * 2;  // floats (the partials are doubles; a torch fp32 allocation is 8-byte aligned)
}

extern "C" int otvae_layernorm_bwd(const float* xs, const float* gy, const float* gamma, const float* mean, const float* rstd, int M,
                                   int D, float* gx, float* dgamma, float* dbeta, float* ws, void* stream) {
    OTVAE_REQUIRE(xs && gy && gamma && mean && rstd && gx && dgamma && dbeta && ws && M > 0 && D > 0,
                  "otvae_layernorm_bwd: bad argument");
    if (D > 64 * LN_MAXV || (size_t)2 * D * sizeof(double) > 64 * 1024) {
        otvae_set_error("otvae_layernorm_bwd: D = %d unsupported (D <= 2048)", D);
        return OTVAE_EUNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    const int P = cdiv(M, LN_ROWS_PER_BLOCK);
#define LN_BWD(NV_) \
    layernorm_bwd_kernel<NV_, false><<<P, 256, (size_t)2 * D * sizeof(double), st>>>(xs, gy, gamma, mean, rstd, M, D, gx, (double*)ws, nullptr, LnDrop{})
    LN_NV_SWITCH(D, LN_BWD)
#undef LN_BWD
    OTVAE_CHECK_LAUNCH("otvae_layernorm_bwd");
    layernorm_param_reduce_kernel<<<cdiv(2 * D, 16), 256, 0, st>>>((const double*)ws, P, D, dgamma, dbeta);
    OTVAE_CHECK_LAUNCH("otvae_layernorm_bwd(reduce)");
    return OTVAE_OK;
}

extern "C" int otvae_layernorm_dropout_bwd(const float* xs, const float* gy, const float* gamma, const float* mean, const float* rstd,
                                           int M, int D, float p, const int64_t* used, float* gx, float* gx_dropped, float* dgamma,
                                           float* dbeta, float* ws, void* stream) {
    OTVAE_REQUIRE(xs && gy && gamma && mean && rstd && used && gx && gx_dropped && dgamma && dbeta && ws && M > 0 && D > 0,
                  "otvae_layernorm_dropout_bwd: bad argument");
    OTVAE_REQUIRE(p >= 0.f && p < 1.f, "otvae_layernorm_dropout_bwd: dropout probability must be in [0, 1)");
    if (D > 64 * LN_MAXV || (size_t)2 * D * sizeof(double) > 64 * 1024) {
        otvae_set_error("otvae_layernorm_dropout_bwd: D = %d unsupported (D <= 2048)", D);
        return OTVAE_EUNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    const int P = cdiv(M, LN_ROWS_PER_BLOCK);
    const LnDrop dr = {dropout_threshold(p), 1.f / (1.f - p), used, 0, nullptr};
#define LN_BWD(NV_) \
    layernorm_bwd_kernel<NV_, true><<<P, 256, (size_t)2 * D * sizeof(double), st>>>(xs, gy, gamma, mean, rstd, M, D, gx, (double*)ws, gx_dropped, dr)
    LN_NV_SWITCH(D, LN_BWD)
#undef LN_BWD
    OTVAE_CHECK_LAUNCH("otvae_layernorm_dropout_bwd");
    layernorm_param_reduce_kernel<<<cdiv(2 * D, 16), 256, 0, st>>>((const double*)ws, P, D, dgamma, dbeta);
    OTVAE_CHECK_LAUNCH("otvae_layernorm_dropout_bwd(reduce)");
    return OTVAE_OK;
}

// keep mask of a otvae_layernorm_dropout_fwd call as uint8 [M][D] (test aid)
__global__ __launch_bounds__(256) void layernorm_dropout_mask_kernel(int M, int D, uint32_t thresh, const int64_t* __restrict__ used,
                                                                     uint8_t* __restrict__ keep) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)M * D) return;
    const int row = (int)(e / D);
    keep[e] = keep_pair(row_hash((uint64_t)used[0], (uint32_t)row), (int)(e - (long)row * D), thresh) ? 1 : 0;
}

extern "C" int otvae_layernorm_dropout_mask(int M, int D, float p, const int64_t* used, uint8_t* keep, void* stream) {
    OTVAE_REQUIRE(used && keep && M > 0 && D > 0 && p >= 0.f && p < 1.f, "otvae_layernorm_dropout_mask: bad argument");
    layernorm_dropout_mask_kernel<<<(int)cdiv((int64_t)M * D, 256), 256, 0, (hipStream_t)stream>>>(M, D, dropout_threshold(p), used, keep);
    OTVAE_CHECK_LAUNCH("otvae_layernorm_dropout_mask");
    return OTVAE_OK;
}
