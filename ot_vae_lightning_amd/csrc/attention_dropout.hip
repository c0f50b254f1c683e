// Self-attention with dropout on the attention probabilities: what nn.MultiheadAttention(dropout=p) computes in training
// mode, which is how the reference's ViT builds its layers (networks/vit.py:157-172 pass `dropout` to every
// TransformerEncoderLayer; configs/vae/vit.yaml trains with 0.1).
//
//   P = softmax(q k^T * scale);  P' = P o keep / (1 - p);  out = P' v          keep[t][s] ~ Bernoulli(1 - p)
// and, with `causal`, the softmax restricted to s <= t (the ViT's `causal_mask`, networks/vit.py:215-217; p may be 0).
//
// The T x T mask is never stored: keep[t][s] is a counter-based hash of (call key, slice, t, s), recomputed by the
// backward pass.  The call key is derived on the device from `key` = {seed, call counter} (int64[2] in device memory,
// so that a captured hipGraph draws fresh masks on every replay: the host bumps the counter with a captured add) and a
// per-call-site `stream_id`; the forward kernel leaves the key it used in `used[0]` for its backward.
//
// Cross-attention (nn.MultiheadAttention(query, memory, memory) inside the nn.TransformerDecoderLayer of the ViT's
// `preprocess_depth` variant, networks/vit.py:171-181,244) is the same arithmetic with Tq queries against Tk keys of another
// token set: the kernels address q, k, v through explicit image / row strides (AttnIn), and the self-attention entry points
// are the special case q | k | v = thirds of one [N][T][3*H*C] tensor.
//
// Same layouts as attention.hip: qkv [N][T][3*H*C] (q | k | v, head-major), out [N][T][H*C], lse [N][H][T] (natural log
// of the un-dropped row sums).  One thread per (slice, token); a slice = one (image, head), its records staged in LDS.
// Written for the ViT shapes (T <= 256 tokens, head width up to 32); these are not the CNN's hot attention kernels and
// carry none of their specialisations.
#include "common.h"
#include "dropout_hash.h"

#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f
#define EXP2(x) __builtin_amdgcn_exp2f(x)
#define ADROP_LDS_FLOATS 16384  // 64 KiB of dynamic LDS

template <int C>
__device__ __forceinline__ float dotr(const float (&a)[C], const float* __restrict__ b) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) s = fmaf(a[c], b[c], s);
    return s;
}

// where q, k, v live: element (image n, token t, head h, channel c) of q is q[n * q_img + t * q_row + h * C + c], likewise k / v
struct AttnIn {
    const float *q, *k, *v;
    long q_img, kv_img;
    int q_row, kv_row;
};
struct AttnGrad {
    float *gq, *gk, *gv;
    long q_img, kv_img;
    int q_row, kv_row;
};

// {k[C], v[C]} records of the block's slices -> sm[sl][t][RS], TR records per slice
template <int C, int RS>
__device__ __forceinline__ void stage_keys(float* __restrict__ sm, const AttnIn& in, long slice0, int nsl, int Tk, int TR, int H) {
    const int per = Tk * 2 * C;
    for (int e = threadIdx.x; e < nsl * per; e += 256) {
        const int sl = e / per, r = e - sl * per;
        const int t = r / (2 * C), j = r - t * 2 * C;
        const long s = slice0 + sl, n = s / H;
        const int h = (int)(s - n * H);
        const float* src = (j < C ? in.k : in.v) + n * in.kv_img + (long)t * in.kv_row + h * C + (j % C);
        sm[(sl * TR + t) * RS + j] = *src;
    }
}

template <int C>
__global__ __launch_bounds__(256) void attn_drop_fwd_kernel(AttnIn in, int N, int Tq, int Tk, int H, int SPB, float scale,
                                                            uint32_t thresh, float inv_keep, const int64_t* __restrict__ key,
                                                            int stream_id, int64_t* __restrict__ used, float* __restrict__ out,
                                                            float* __restrict__ lse, int causal) {
    extern __shared__ __align__(16) float sm[];
    constexpr int RS = 2 * C;
    const int HC = H * C, TM = Tq > Tk ? Tq : Tk;
    const long total = (long)N * H, slice0 = (long)blockIdx.x * SPB;
    const int nsl = (int)min((long)SPB, total - slice0);
    stage_keys<C, RS>(sm, in, slice0, nsl, Tk, Tk, H);
    const uint64_t ck = key ? call_key(key, stream_id) : 0ull;  // no key: no dropout asked for (thresh == 0 keeps every pair)
    if (used && blockIdx.x == 0 && threadIdx.x == 0) used[0] = (int64_t)ck;
    __syncthreads();
    const int sl = threadIdx.x / TM, t = threadIdx.x - sl * TM;
    if (sl >= nsl || t >= Tq) return;  // no barrier below
    const long s = slice0 + sl, n = s / H;
    const int h = (int)(s - n * H);
    const float* kv = sm + (size_t)sl * Tk * RS;
    float q[C], acc[C];
    const float qs = scale * LOG2E;  // scores in the log2 domain: exp is a bare v_exp_f32
    const float* qp = in.q + n * in.q_img + (long)t * in.q_row + h * C;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        q[c] = qp[c] * qs;
        acc[c] = 0.f;
    }
    const int jend = causal ? t + 1 : Tk;  // causal: token t attends to tokens 0..t (the -inf upper triangle of
    float mx = -INFINITY;                  // nn.Transformer.generate_square_subsequent_mask)
    for (int j = 0; j < jend; ++j) mx = fmaxf(mx, dotr<C>(q, kv + j * RS));
    const uint32_t rh = row_hash(ck, (uint32_t)(s * Tq + t));
    float l = 0.f;
    for (int j = 0; j < jend; ++j) {
        const float* r = kv + j * RS;
        const float p = EXP2(dotr<C>(q, r) - mx);
        l += p;  // the softmax normaliser sees every key; only the value sum is thinned
        const float w = keep_pair(rh, j, thresh) ? p : 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = fmaf(w, r[C + c], acc[c]);
    }
    const float norm = inv_keep / l;
#pragma unroll
    for (int c = 0; c < C; ++c) out[(n * Tq + t) * HC + h * C + c] = acc[c] * norm;
    lse[s * Tq + t] = mx * LN2 + __logf(l);
}

// Phase A (thread = query t): dq[t] = scale * sum_s dS[t][s] k[s], dS = P o (dP - delta), dP[t][s] = keep/(1-p) * gout[t].v[s],
//   delta[t] = sum_s P dP = gout[t].out[t]  (P' and dP carry the same mask, so the identity of the un-dropped case holds).
// Phase B (thread = key s): the block's LDS is re-filled with the query-side records {q, gout, lse, delta, row hash} straight
//   from the registers of phase A;  dv[s] = sum_t P'[t][s] gout[t],  dk[s] = scale * sum_t dS[t][s] q[t].
template <int C>
__global__ __launch_bounds__(256) void attn_drop_bwd_kernel(AttnIn in, const float* __restrict__ out, const float* __restrict__ lse,
                                                            const float* __restrict__ gout, int N, int Tq, int Tk, int H, int SPB,
                                                            float scale, uint32_t thresh, float inv_keep,
                                                            const int64_t* __restrict__ used, AttnGrad gr, int causal) {
    extern __shared__ __align__(16) float sm[];
    constexpr int RS = 2 * C + 3;
    const int HC = H * C, TM = Tq > Tk ? Tq : Tk;
    const long total = (long)N * H, slice0 = (long)blockIdx.x * SPB;
    const int nsl = (int)min((long)SPB, total - slice0);
    stage_keys<C, RS>(sm, in, slice0, nsl, Tk, TM, H);
    const uint64_t ck = used ? (uint64_t)used[0] : 0ull;
    __syncthreads();
    const int sl = threadIdx.x / TM, t = threadIdx.x - sl * TM;
    const bool in_slice = sl < nsl;
    const bool active = in_slice && t < Tq;
    const long s = slice0 + (in_slice ? sl : 0), n = s / H;
    const int h = (int)(s - n * H);
    float* rec = sm + (size_t)(in_slice ? sl : 0) * TM * RS;
    const float qs = scale * LOG2E;
    float q[C], g[C];
    float L2 = 0.f, delta = 0.f;
    uint32_t rh = 0;
    if (active) {
        float dq[C];
        const float* qp = in.q + n * in.q_img + (long)t * in.q_row + h * C;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            q[c] = qp[c];
            g[c] = gout[(n * Tq + t) * HC + h * C + c];
            delta = fmaf(g[c], out[(n * Tq + t) * HC + h * C + c], delta);
            dq[c] = 0.f;
        }
        L2 = lse[s * Tq + t] * LOG2E;
        rh = row_hash(ck, (uint32_t)(s * Tq + t));
        const int jend = causal ? t + 1 : Tk;
        // consistent delta (round 4, as in csrc/attention.hip phase A): delta = gout . out comes from the forward pass's output, whose
        // rounding is independent of this pass's p and dP; on a peaked row dP* - delta then carries an absolute error eps |dP| where
        // the true value is (1 - p*) x something, the same sign for every key.  r = sum p (dP - delta) (zero for a consistent delta),
        // sum p and sum p k ride along; delta' = delta + r / sum p goes into the query record for phase B (which recomputes p and dP
        // with the same operands in the same order: bit-identical), dq is corrected by - (r / sum p) sum p k.
        float rsum = 0.f, psum = 0.f, bk[C];
#pragma unroll
        for (int c = 0; c < C; ++c) bk[c] = 0.f;
        for (int j = 0; j < jend; ++j) {
            const float* r = rec + j * RS;
            const float p = EXP2(fmaf(dotr<C>(q, r), qs, -L2));
            const float dp = keep_pair(rh, j, thresh) ? dotr<C>(g, r + C) * inv_keep : 0.f;
            const float ds = p * (dp - delta);
            rsum += ds;
            psum += p;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                dq[c] = fmaf(ds, r[c], dq[c]);
                bk[c] = fmaf(p, r[c], bk[c]);
            }
        }
        const float corr = psum > 0.f ? rsum / psum : 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) dq[c] = fmaf(-corr, bk[c], dq[c]);
        delta += corr;
        float* gqp = gr.gq + n * gr.q_img + (long)t * gr.q_row + h * C;
#pragma unroll
        for (int c = 0; c < C; ++c) gqp[c] = dq[c] * scale;
    }
    __syncthreads();  // every thread of the block: the key records are dead from here on
    if (active) {
        float* mine = rec + t * RS;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            mine[c] = q[c];
            mine[C + c] = g[c];
        }
        mine[2 * C] = L2;
        mine[2 * C + 1] = delta;
        mine[2 * C + 2] = __uint_as_float(rh);
    }
    __syncthreads();
    if (!in_slice || t >= Tk) return;
    float k[C], v[C], dk[C], dv[C];
    const float* kp = in.k + n * in.kv_img + (long)t * in.kv_row + h * C;
    const float* vp = in.v + n * in.kv_img + (long)t * in.kv_row + h * C;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        k[c] = kp[c];
        v[c] = vp[c];
        dk[c] = 0.f;
        dv[c] = 0.f;
    }
    for (int i = causal ? t : 0; i < Tq; ++i) {  // causal: key t is seen by queries t..T-1
        const float* r = rec + i * RS;
        const float p = EXP2(fmaf(dotr<C>(k, r), qs, -r[2 * C]));
        const bool keep = keep_pair(__float_as_uint(r[2 * C + 2]), t, thresh);
        const float w = keep ? p * inv_keep : 0.f;
        const float dp = keep ? dotr<C>(v, r + C) * inv_keep : 0.f;
        const float ds = p * (dp - r[2 * C + 1]);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            dv[c] = fmaf(w, r[C + c], dv[c]);
            dk[c] = fmaf(ds, r[c], dk[c]);
        }
    }
    float* gkp = gr.gk + n * gr.kv_img + (long)t * gr.kv_row + h * C;
    float* gvp = gr.gv + n * gr.kv_img + (long)t * gr.kv_row + h * C;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        gkp[c] = dk[c] * scale;
        gvp[c] = dv[c];
    }
}

__global__ __launch_bounds__(256) void attn_drop_mask_kernel(long rows, int T, uint32_t thresh, const int64_t* __restrict__ used,
                                                             uint8_t* __restrict__ keep) {
    const uint64_t ck = (uint64_t)used[0];
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= rows * T) return;
    const long row = e / T;
    keep[e] = keep_pair(row_hash(ck, (uint32_t)row), (int)(e - row * T), thresh) ? 1 : 0;
}

static int adrop_plan(const char* who, int N, int Tq, int Tk, int H, int C, float p, int rs, int* spb, uint32_t* thresh) {
    OTVAE_REQUIRE(N > 0 && Tq > 0 && Tk > 0 && H > 0 && C > 0, "%s: bad sizes", who);
    OTVAE_REQUIRE(p >= 0.f && p < 1.f, "%s: dropout probability must be in [0, 1)", who);
    OTVAE_REQUIRE((int64_t)N * H * Tq < ((int64_t)1 << 32), "%s: N*H*T must stay below 2^32 (the mask hash counts rows in 32 bits)", who);
    const int T = Tq > Tk ? Tq : Tk;
    if (T > 256 || T * rs > ADROP_LDS_FLOATS) {
        otvae_set_error("%s: a slice of T=%d tokens, head width %d does not fit (T <= 256 and T*(2C+3) <= %d floats of LDS)", who, T,
                        C, ADROP_LDS_FLOATS);
        return OTVAE_EUNSUPPORTED;
    }
    int s = 256 / T, cap = ADROP_LDS_FLOATS / (T * rs);
    s = s < cap ? s : cap;
    while (s > 1 && cdiv((int64_t)N * H, s) < 512) s >>= 1;  // few slices: rather thinner blocks than idle CUs (256 of them)
    *spb = s;
    *thresh = dropout_threshold(p);
    return OTVAE_OK;
}

#define ADROP_C_SWITCH(C_, MACRO)                                                                                      \
    switch (C_) {                                                                                                      \
        case 1: MACRO(1); break;                                                                                       \
        case 2: MACRO(2); break;                                                                                       \
        case 4: MACRO(4); break;                                                                                       \
        case 8: MACRO(8); break;                                                                                       \
        case 16: MACRO(16); break;                                                                                     \
        case 32: MACRO(32); break;                                                                                     \
        default:                                                                                                       \
            otvae_set_error("attention with dropout: head width C=%d not instantiated (1,2,4,8,16,32)", C_);           \
            return OTVAE_EUNSUPPORTED;                                                                                 \
    }

static int attn_general_fwd(const char* who, const AttnIn& in, int N, int Tq, int Tk, int H, int C, float scale, float p, int causal,
                            const int64_t* key, int stream_id, float* out, float* lse, int64_t* used, void* stream) {
    OTVAE_REQUIRE(scale > 0.f && stream_id >= 0 && stream_id < 4095, "%s: bad scale or stream_id", who);
    OTVAE_REQUIRE(!causal || Tq == Tk, "%s: the causal mask needs as many queries as keys", who);
    OTVAE_REQUIRE(p == 0.f || (key && used), "%s: dropout needs `key` and `used`", who);
    int spb;
    uint32_t thresh;
    int rc = adrop_plan(who, N, Tq, Tk, H, C, p, 2 * C + 3, &spb, &thresh);  // the backward's plan: same slices
    if (rc) return rc;
    const int grid = (int)cdiv((int64_t)N * H, spb);
    const size_t lds = (size_t)spb * Tk * 2 * C * sizeof(float);
    const float inv_keep = 1.f / (1.f - p);
#define FWD_K(CC) \
    attn_drop_fwd_kernel<CC><<<grid, 256, lds, (hipStream_t)stream>>>(in, N, Tq, Tk, H, spb, scale, thresh, inv_keep, key, stream_id, used, out, lse, causal)
    ADROP_C_SWITCH(C, FWD_K)
#undef FWD_K
    OTVAE_CHECK_LAUNCH(who);
    return OTVAE_OK;
}

static int attn_general_bwd(const char* who, const AttnIn& in, const float* out, const float* lse, const float* gout, int N, int Tq,
                            int Tk, int H, int C, float scale, float p, int causal, const int64_t* used, const AttnGrad& gr,
                            void* stream) {
    OTVAE_REQUIRE(scale > 0.f, "%s: scale must be positive", who);
    OTVAE_REQUIRE(!causal || Tq == Tk, "%s: the causal mask needs as many queries as keys", who);
    OTVAE_REQUIRE(p == 0.f || used, "%s: dropout needs the forward pass's `used`", who);
    int spb;
    uint32_t thresh;
    int rc = adrop_plan(who, N, Tq, Tk, H, C, p, 2 * C + 3, &spb, &thresh);
    if (rc) return rc;
    const int grid = (int)cdiv((int64_t)N * H, spb);
    const size_t lds = (size_t)spb * (Tq > Tk ? Tq : Tk) * (2 * C + 3) * sizeof(float);
    const float inv_keep = 1.f / (1.f - p);
#define BWD_K(CC) \
    attn_drop_bwd_kernel<CC><<<grid, 256, lds, (hipStream_t)stream>>>(in, out, lse, gout, N, Tq, Tk, H, spb, scale, thresh, inv_keep, used, gr, causal)
    ADROP_C_SWITCH(C, BWD_K)
#undef BWD_K
    OTVAE_CHECK_LAUNCH(who);
    return OTVAE_OK;
}

extern "C" int otvae_attn_dropout_fwd(const float* qkv, int N, int T, int H, int C, float scale, float p, int causal,
                                      const int64_t* key, int stream_id, float* out, float* lse, int64_t* used, void* stream) {
    OTVAE_REQUIRE(qkv && out && lse && key && used, "otvae_attn_dropout_fwd: NULL tensor");
    const int HC = H * C, W3 = 3 * HC;
    const AttnIn in = {qkv, qkv + HC, qkv + 2 * HC, (long)T * W3, (long)T * W3, W3, W3};
    return attn_general_fwd("otvae_attn_dropout_fwd", in, N, T, T, H, C, scale, p, causal, key, stream_id, out, lse, used, stream);
}

extern "C" int otvae_attn_dropout_bwd(const float* qkv, const float* out, const float* lse, const float* gout, int N, int T, int H,
                                      int C, float scale, float p, int causal, const int64_t* used, float* gqkv, void* stream) {
    OTVAE_REQUIRE(qkv && out && lse && gout && used && gqkv, "otvae_attn_dropout_bwd: NULL tensor");
    const int HC = H * C, W3 = 3 * HC;
    const AttnIn in = {qkv, qkv + HC, qkv + 2 * HC, (long)T * W3, (long)T * W3, W3, W3};
    const AttnGrad gr = {gqkv, gqkv + HC, gqkv + 2 * HC, (long)T * W3, (long)T * W3, W3, W3};
    return attn_general_bwd("otvae_attn_dropout_bwd", in, out, lse, gout, N, T, T, H, C, scale, p, causal, used, gr, stream);
}

// Cross-attention: Tq queries of one token set against Tk keys / values of another.  q / k / v are addressed as
// ptr[n * img_stride + t * row_stride + h * C + c] (floats), so a caller may pass thirds of one in-projected tensor or three
// separate ones; out [N][Tq][H*C], lse [N][H][Tq].  p == 0: key and used may be NULL.
extern "C" int otvae_attn_cross_fwd(const float* q, int64_t q_img_stride, int q_row_stride, const float* k, const float* v,
                                    int64_t kv_img_stride, int kv_row_stride, int N, int Tq, int Tk, int H, int C, float scale,
                                    float p, const int64_t* key, int stream_id, float* out, float* lse, int64_t* used, void* stream) {
    OTVAE_REQUIRE(q && k && v && out && lse, "otvae_attn_cross_fwd: NULL tensor");
    OTVAE_REQUIRE(q_row_stride >= H * C && kv_row_stride >= H * C && q_img_stride >= (int64_t)Tq * q_row_stride &&
                      kv_img_stride >= (int64_t)Tk * kv_row_stride,
                  "otvae_attn_cross_fwd: strides smaller than the rows / images they step over");
    const AttnIn in = {q, k, v, (long)q_img_stride, (long)kv_img_stride, q_row_stride, kv_row_stride};
    return attn_general_fwd("otvae_attn_cross_fwd", in, N, Tq, Tk, H, C, scale, p, 0, key, stream_id, out, lse, used, stream);
}

extern "C" int otvae_attn_cross_bwd(const float* q, int64_t q_img_stride, int q_row_stride, const float* k, const float* v,
                                    int64_t kv_img_stride, int kv_row_stride, const float* out, const float* lse, const float* gout,
                                    int N, int Tq, int Tk, int H, int C, float scale, float p, const int64_t* used, float* gq,
                                    int64_t gq_img_stride, int gq_row_stride, float* gk, float* gv, int64_t gkv_img_stride,
                                    int gkv_row_stride, void* stream) {
    OTVAE_REQUIRE(q && k && v && out && lse && gout && gq && gk && gv, "otvae_attn_cross_bwd: NULL tensor");
    OTVAE_REQUIRE(q_row_stride >= H * C && kv_row_stride >= H * C && gq_row_stride >= H * C && gkv_row_stride >= H * C &&
                      q_img_stride >= (int64_t)Tq * q_row_stride && kv_img_stride >= (int64_t)Tk * kv_row_stride &&
                      gq_img_stride >= (int64_t)Tq * gq_row_stride && gkv_img_stride >= (int64_t)Tk * gkv_row_stride,
                  "otvae_attn_cross_bwd: strides smaller than the rows / images they step over");
    const AttnIn in = {q, k, v, (long)q_img_stride, (long)kv_img_stride, q_row_stride, kv_row_stride};
    const AttnGrad gr = {gq, gk, gv, (long)gq_img_stride, (long)gkv_img_stride, gq_row_stride, gkv_row_stride};
    return attn_general_bwd("otvae_attn_cross_bwd", in, out, lse, gout, N, Tq, Tk, H, C, scale, p, 0, used, gr, stream);
}

extern "C" int otvae_attn_dropout_mask(int N, int T, int H, float p, const int64_t* used, uint8_t* keep, void* stream) {
    OTVAE_REQUIRE(used && keep, "otvae_attn_dropout_mask: NULL tensor");
    int spb;
    uint32_t thresh;
    int rc = adrop_plan("otvae_attn_dropout_mask", N, T, T, H, 1, p, 5, &spb, &thresh);
    if (rc) return rc;
    const long rows = (long)N * H * T;
    attn_drop_mask_kernel<<<(int)cdiv((int64_t)rows * T, 256), 256, 0, (hipStream_t)stream>>>(rows, T, thresh, used, keep);
    OTVAE_CHECK_LAUNCH("otvae_attn_dropout_mask");
    return OTVAE_OK;
}

extern "C" int otvae_attn_cross_mask(int N, int Tq, int Tk, int H, float p, const int64_t* used, uint8_t* keep, void* stream) {
    OTVAE_REQUIRE(used && keep, "otvae_attn_cross_mask: NULL tensor");
    int spb;
    uint32_t thresh;
    int rc = adrop_plan("otvae_attn_cross_mask", N, Tq, Tk, H, 1, p, 5, &spb, &thresh);
    if (rc) return rc;
    const long rows = (long)N * H * Tq;
    attn_drop_mask_kernel<<<(int)cdiv((int64_t)rows * Tk, 256), 256, 0, (hipStream_t)stream>>>(rows, Tk, thresh, used, keep);
    OTVAE_CHECK_LAUNCH("otvae_attn_cross_mask");
    return OTVAE_OK;
}
