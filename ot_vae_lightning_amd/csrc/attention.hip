// Fused QKV self-attention for the CNN's AttentionBlock (reference networks/nets_utils.py:63-82, called from
// networks/cnn.py:235-240): qkv [N][T][3*H*C] channels-last -> out [N][T][H*C].
//   w[t][s] = sum_c (q[t][c]*C^-1/2) (k[s][c]*C^-1/2);  out[t][c] = sum_s softmax_s(w)[t][s] v[s][c]
// The reference materialises the TxT matrix (6 MiB/image at T=1024); here it never leaves registers.
//
// Head width C is 1..16 and T is 1..1024 in every configuration, so the products are far too thin for MFMA tiles
// (K = C): one VALU lane owns one query row (forward, dQ) or one key row (dK, dV) and the other side is read as a
// wave-uniform LDS broadcast.  A workgroup stages the whole qkv slab of its image(s) in LDS with coalesced loads.
#include "common.h"

#define ATTN_MAX_LDS_FLOATS 36864  // 144 KiB

template <int C>
__device__ __forceinline__ float dotc(const float (&a)[C], const float* __restrict__ b) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) s = fmaf(a[c], b[c], s);
    return s;
}

template <int C>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ qkv, int N, int T, int H, int NB,
                                                       float* __restrict__ out, float* __restrict__ lse) {
    extern __shared__ __align__(16) float sm[];
    const int HC = H * C, W3 = 3 * HC;
    const int n_base = blockIdx.x * NB;
    const int nb = min(NB, N - n_base);
    const int slab = nb * T * W3;
    const float* src = qkv + (size_t)n_base * T * W3;
    for (int i = threadIdx.x; i < slab; i += 256) sm[i] = src[i];
    __syncthreads();
    const float inv_c = 1.f / (float)C;
    const int items = nb * H * T;
    for (int it = threadIdx.x; it < items; it += 256) {
        const int t = it % T;
        const int h = (it / T) % H;
        const int nl = it / (T * H);
        const float* base = sm + (size_t)nl * T * W3;
        float q[C], acc[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            q[c] = base[t * W3 + h * C + c] * inv_c;
            acc[c] = 0.f;
        }
        const float* kp = base + HC + h * C;
        const float* vp = base + 2 * HC + h * C;
        float mx = -INFINITY;
        for (int s = 0; s < T; ++s) mx = fmaxf(mx, dotc<C>(q, kp + s * W3));
        float l = 0.f;
        for (int s = 0; s < T; ++s) {
            const float p = __expf(dotc<C>(q, kp + s * W3) - mx);
            l += p;
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = fmaf(p, vp[s * W3 + c], acc[c]);
        }
        const float rl = 1.f / l;
        float* o = out + ((size_t)(n_base + nl) * T + t) * HC + h * C;
#pragma unroll
        for (int c = 0; c < C; ++c) o[c] = acc[c] * rl;
        lse[((size_t)(n_base + nl) * H + h) * T + t] = mx + __logf(l);
    }
}

// Backward.  LDS: qkv slab | gout slab [nb][T][HC] | lse [nb][H][T] | delta [nb][H][T]
template <int C>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ out,
                                                       const float* __restrict__ lse_g, const float* __restrict__ gout,
                                                       int N, int T, int H, int NB, float* __restrict__ gqkv) {
    extern __shared__ __align__(16) float sm[];
    const int HC = H * C, W3 = 3 * HC;
    const int n_base = blockIdx.x * NB;
    const int nb = min(NB, N - n_base);
    float* s_qkv = sm;
    float* s_go = s_qkv + (size_t)NB * T * W3;
    float* s_lse = s_go + (size_t)NB * T * HC;
    float* s_del = s_lse + (size_t)NB * H * T;
    {
        const int slab = nb * T * W3;
        const float* src = qkv + (size_t)n_base * T * W3;
        for (int i = threadIdx.x; i < slab; i += 256) s_qkv[i] = src[i];
        const int slab2 = nb * T * HC;
        const float* gsrc = gout + (size_t)n_base * T * HC;
        for (int i = threadIdx.x; i < slab2; i += 256) s_go[i] = gsrc[i];
        const int nl_items = nb * H * T;
        const float* lsrc = lse_g + (size_t)n_base * H * T;
        for (int i = threadIdx.x; i < nl_items; i += 256) s_lse[i] = lsrc[i];
    }
    __syncthreads();
    const int items = nb * H * T;
    // delta[nl][h][t] = sum_c gout * out
    for (int it = threadIdx.x; it < items; it += 256) {
        const int t = it % T;
        const int h = (it / T) % H;
        const int nl = it / (T * H);
        const float* o = out + ((size_t)(n_base + nl) * T + t) * HC + h * C;
        const float* g = s_go + ((size_t)nl * T + t) * HC + h * C;
        float d = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) d = fmaf(g[c], o[c], d);
        s_del[((size_t)nl * H + h) * T + t] = d;
    }
    __syncthreads();
    const float inv_c = 1.f / (float)C;
    // phase A: one lane per query row -> dQ
    for (int it = threadIdx.x; it < items; it += 256) {
        const int t = it % T;
        const int h = (it / T) % H;
        const int nl = it / (T * H);
        const float* base = s_qkv + (size_t)nl * T * W3;
        const float* kp = base + HC + h * C;
        const float* vp = base + 2 * HC + h * C;
        float q[C], go[C], dq[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            q[c] = base[t * W3 + h * C + c] * inv_c;
            go[c] = s_go[((size_t)nl * T + t) * HC + h * C + c];
            dq[c] = 0.f;
        }
        const float ls = s_lse[((size_t)nl * H + h) * T + t];
        const float dl = s_del[((size_t)nl * H + h) * T + t];
        for (int s = 0; s < T; ++s) {
            const float p = __expf(dotc<C>(q, kp + s * W3) - ls);
            const float dp = dotc<C>(go, vp + s * W3);
            const float ds = p * (dp - dl);
#pragma unroll
            for (int c = 0; c < C; ++c) dq[c] = fmaf(ds, kp[s * W3 + c], dq[c]);
        }
        float* o = gqkv + ((size_t)(n_base + nl) * T + t) * W3 + h * C;
#pragma unroll
        for (int c = 0; c < C; ++c) o[c] = dq[c] * inv_c;
    }
    // phase B: one lane per key row -> dK, dV
    for (int it = threadIdx.x; it < items; it += 256) {
        const int s = it % T;
        const int h = (it / T) % H;
        const int nl = it / (T * H);
        const float* base = s_qkv + (size_t)nl * T * W3;
        const float* qp = base + h * C;
        const float* gp = s_go + (size_t)nl * T * HC + h * C;
        const float* lp = s_lse + ((size_t)nl * H + h) * T;
        const float* dlp = s_del + ((size_t)nl * H + h) * T;
        float k[C], v[C], dk[C], dv[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            k[c] = base[s * W3 + HC + h * C + c] * inv_c;  // fold the 1/C of the score into k here
            v[c] = base[s * W3 + 2 * HC + h * C + c];
            dk[c] = dv[c] = 0.f;
        }
        for (int t = 0; t < T; ++t) {
            const float p = __expf(dotc<C>(k, qp + t * W3) - lp[t]);
            const float dp = dotc<C>(v, gp + t * HC);
            const float ds = p * (dp - dlp[t]);
#pragma unroll
            for (int c = 0; c < C; ++c) {
                dv[c] = fmaf(p, gp[t * HC + c], dv[c]);
                dk[c] = fmaf(ds, qp[t * W3 + c], dk[c]);
            }
        }
        float* o = gqkv + ((size_t)(n_base + nl) * T + s) * W3 + h * C;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            o[HC + c] = dk[c] * inv_c;
            o[2 * HC + c] = dv[c];
        }
    }
}

// Dynamic LDS above the 64 KiB default needs a one-time function attribute; done once per kernel instantiation and
// only ever raised, outside of any captured region in practice (the first eager call of a shape does it).
static size_t fwd_lds_set[33] = {0}, bwd_lds_set[33] = {0};
static bool ensure_lds(const void* fn, size_t* cur, size_t want) {
    if (want <= 65536 || want <= *cur) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(ATTN_MAX_LDS_FLOATS * sizeof(float))) !=
        hipSuccess) {
        otvae_set_error("attention: cannot raise dynamic LDS limit");
        return false;
    }
    *cur = ATTN_MAX_LDS_FLOATS * sizeof(float);
    return true;
}

static int attn_nb(int T, int H, int C, int per_image_floats) {
    int nb = 256 / (H * T);
    if (nb < 1) nb = 1;
    int cap = ATTN_MAX_LDS_FLOATS / per_image_floats;
    if (nb > cap) nb = cap;
    return nb;
}

#define ATTN_DISPATCH(C_, KERNEL, ...)                 \
    switch (C_) {                                      \
        case 1: KERNEL(1, __VA_ARGS__); break;         \
        case 2: KERNEL(2, __VA_ARGS__); break;         \
        case 3: KERNEL(3, __VA_ARGS__); break;         \
        case 4: KERNEL(4, __VA_ARGS__); break;         \
        case 6: KERNEL(6, __VA_ARGS__); break;         \
        case 8: KERNEL(8, __VA_ARGS__); break;         \
        case 12: KERNEL(12, __VA_ARGS__); break;       \
        case 16: KERNEL(16, __VA_ARGS__); break;       \
        case 32: KERNEL(32, __VA_ARGS__); break;       \
        default:                                       \
            otvae_set_error("attention: head width C=%d not instantiated (1,2,3,4,6,8,12,16,32)", C_); \
            return OTVAE_EUNSUPPORTED;                 \
    }

extern "C" int otvae_attn_fwd(const float* qkv, int N, int T, int H, int C, float* out, float* lse, void* stream) {
    OTVAE_REQUIRE(qkv && out && lse && N > 0 && T > 0 && H > 0 && C > 0, "otvae_attn_fwd: bad argument");
    const int per_img = T * 3 * H * C;
    if (per_img > ATTN_MAX_LDS_FLOATS) {
        otvae_set_error("otvae_attn_fwd: T*3*H*C = %d floats exceeds the LDS slab (%d)", per_img, ATTN_MAX_LDS_FLOATS);
        return OTVAE_EUNSUPPORTED;
    }
    const int NB = attn_nb(T, H, C, per_img);
    const size_t lds = (size_t)NB * per_img * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    const int grid = cdiv(N, NB);
#define FWD_K(CC, ...)                                                                    \
    do {                                                                                  \
        if (!ensure_lds((const void*)attn_fwd_kernel<CC>, &fwd_lds_set[CC], lds)) return OTVAE_ELAUNCH; \
        attn_fwd_kernel<CC><<<grid, 256, lds, st>>>(qkv, N, T, H, NB, out, lse);          \
    } while (0)
    ATTN_DISPATCH(C, FWD_K, 0)
#undef FWD_K
    OTVAE_CHECK_LAUNCH("otvae_attn_fwd");
    return OTVAE_OK;
}

extern "C" int otvae_attn_bwd(const float* qkv, const float* out, const float* lse, const float* gout, int N, int T, int H,
                              int C, float* gqkv, void* stream) {
    OTVAE_REQUIRE(qkv && out && lse && gout && gqkv && N > 0 && T > 0 && H > 0 && C > 0, "otvae_attn_bwd: bad argument");
    const int per_img = T * (3 * H * C + H * C + 2 * H);
    if (per_img > ATTN_MAX_LDS_FLOATS) {
        otvae_set_error("otvae_attn_bwd: per-image LDS slab of %d floats exceeds %d", per_img, ATTN_MAX_LDS_FLOATS);
        return OTVAE_EUNSUPPORTED;
    }
    const int NB = attn_nb(T, H, C, per_img);
    const size_t lds = (size_t)NB * per_img * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    const int grid = cdiv(N, NB);
#define BWD_K(CC, ...)                                                                    \
    do {                                                                                  \
        if (!ensure_lds((const void*)attn_bwd_kernel<CC>, &bwd_lds_set[CC], lds)) return OTVAE_ELAUNCH; \
        attn_bwd_kernel<CC><<<grid, 256, lds, st>>>(qkv, out, lse, gout, N, T, H, NB, gqkv); \
    } while (0)
    ATTN_DISPATCH(C, BWD_K, 0)
#undef BWD_K
    OTVAE_CHECK_LAUNCH("otvae_attn_bwd");
    return OTVAE_OK;
}
