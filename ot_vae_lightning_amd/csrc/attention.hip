// Fused QKV self-attention for the CNN's AttentionBlock (reference networks/nets_utils.py:63-82, called from
// networks/cnn.py:235-240): qkv [N][T][3*H*C] channels-last -> out [N][T][H*C].
//   w[t][s] = sum_c (q[t][c]*C^-1/2) (k[s][c]*C^-1/2);  out[t][c] = sum_s softmax_s(w)[t][s] v[s][c]
// The reference materialises the TxT matrix (6 MiB/image at T=1024); here it never leaves registers.
//
// Head width C is 1..16 and T is 1..1024 in every configuration, so the products are far too thin for MFMA tiles
// (K = C; fp32 MFMA runs at the VALU rate anyway): the kernels are exp/VALU-bound and organised to minimise issue slots
// per (query, key) pair:
//   * a "slice" = one (image, head).  Its keys are staged in LDS as records {k[C], v[C]} (and, for backward, its
//     queries as {q/C [C], gout[C], lse, delta}) so that ONE wave-uniform LDS read (b64/b128 broadcast) feeds a pair;
//   * every lane owns QPT (=4 when T >= 256) consecutive queries (or keys) of one slice, so each record read is
//     amortised over QPT pairs and QPT independent exp/FMA chains hide each other's latency;
//   * for C == 1 the row maximum is closed-form (q * max_s k or q * min_s k), so the forward needs a single pass.
#include "common.h"
#include <type_traits>

#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f
#define EXP2(x) __builtin_amdgcn_exp2f(x)
#define ATTN_LDS_FLOATS 16384  // 64 KiB dynamic LDS budget (the default limit: no function attribute needed)
// OTVAE_ATTN_PACK=0 (compile time, together with -fno-slp-vectorize) spells the pair arithmetic as plain v_fma_f32 instead of
// v_pk_fma_f32 over two queries.  Measured on the MI355X at 4 waves per SIMD (profiles/r03_attn_ab.txt): plain is 19-29 % SLOWER
// (C = 1: 153 -> 182 / 154 -> 198 us, C = 2 backward 77 -> 100 us) -- a v_fma_f32 costs ~3.7 issue cycles there, a v_pk_fma_f32
// ~5.8 for two FMAs; the cycle table's 2-cycle v_fma_f32 was not reproduced in these loops.  Packed stays the default.
#ifndef OTVAE_ATTN_PACK
#define OTVAE_ATTN_PACK 1
#endif
// largest distance (log2 units) between the Cauchy-Schwarz bound |q| max|k| and the smallest possible row maximum for which
// the forward pass skips its row-maximum pass over the keys: every p = 2^(score - bound) then lies in [2^-64, 1]
#define ATTN_BOUND_GAP 64.0f

template <int C>
__device__ __forceinline__ float dotc(const float (&a)[C], const float* __restrict__ b) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) s = fmaf(a[c], b[c], s);
    return s;
}

template <int C>
struct Rec {
    static constexpr int KV = 2 * C;                     // {k[C], v[C]}
    static constexpr int QG = (2 * C + 2 + 3) & ~3;      // {q/C [C], gout[C], lse, delta} padded to 16 bytes
};

// floor(k / d) by one multiply for 0 <= k < 2^22 (exact, see fast_div in conv.hip); the staging loops decode
// (slice, token, channel) per element and a 64-bit udiv there costs more than the element's share of the main loop
__device__ __forceinline__ int adiv(int k, int d, float inv_d) {
    return k < (1 << 22) ? (int)(((float)k + 0.5f) * inv_d) : k / d;
}

// stage {k, v} records of the block's slices: sm[sl][t][STRIDE], the first 2C floats of each record (STRIDE > 2C leaves
// room for the per-key products of the AUX forward kernel)
template <int C, int STRIDE = 2 * C>
__device__ __forceinline__ void stage_kv(float* __restrict__ sm, const float* __restrict__ qkv, long slice0, int nsl, int T,
                                         int H) {
    const int HC = H * C, W3 = 3 * HC;
    const int per = T * 2 * C;
    const float inv_per = 1.0f / (float)per, inv_h = 1.0f / (float)H;
    const int n0 = (int)(slice0 / H), h0 = (int)(slice0 - (long)n0 * H);  // once per block
    // four elements per thread and trip, the loads issued together (a trip per element exposed one global round trip each:
    // 16 dependent trips at T = 256, C = 2 -- a fifth of the backward kernel's time)
    const int total = nsl * per, NT = blockDim.x;
    for (int e0 = threadIdx.x; e0 < total; e0 += 4 * NT) {
        float val[4];
        int dst[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * NT;
            dst[u] = -1;
            val[u] = 0.f;
            if (e < total) {
                const int sl = adiv(e, per, inv_per), r = e - sl * per;
                const int t = r / (2 * C), j = r - t * 2 * C;  // compile-time divisor
                const int which = j / C, c = j - which * C;
                const int hs = h0 + sl;                          // < H + slices per block
                const int dn = adiv(hs, H, inv_h);
                const long n = n0 + dn;
                const int h = hs - dn * H;
                dst[u] = STRIDE == 2 * C ? e : (sl * T + t) * STRIDE + j;
                val[u] = qkv[(n * T + t) * W3 + (1 + which) * HC + h * C + c];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (dst[u] >= 0) sm[dst[u]] = val[u];
    }
}

// Combine one value per lane over the TPS consecutive lanes of a slice (OP 0 sum, 1 max, 2 min).  All threads of the
// block call it together; TPS must be a multiple of 64 or a power of two below 64 (slice_reduce_ok).
__device__ __forceinline__ bool slice_reduce_ok(int TPS) { return TPS % 64 == 0 || (TPS < 64 && (TPS & (TPS - 1)) == 0); }
template <int OP>
__device__ __forceinline__ float slice_combine(float a, float b) {
    return OP == 0 ? a + b : OP == 1 ? fmaxf(a, b) : fminf(a, b);
}
template <int OP>
__device__ __forceinline__ float slice_reduce(float v, int TPS, float* red /* 4 floats of LDS */) {
    if (TPS % 64 == 0) {
        v = OP == 0 ? wave_sum(v) : OP == 1 ? wave_max(v) : wave_min(v);
        if (TPS > 64) {
            __syncthreads();  // the previous call's readers are done with red
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
            __syncthreads();
            const int wps = TPS >> 6, w0 = (int)(threadIdx.x >> 6) / wps * wps;  // the waves of this lane's slice
            v = red[w0];
            for (int w = 1; w < wps; ++w) v = slice_combine<OP>(v, red[w0 + w]);
        }
    } else {
        for (int m = 1; m < TPS; m <<= 1) v = slice_combine<OP>(v, __shfl_xor(v, m, 64));
    }
    return v;
}

// AUX (head widths 1 and 2, training): besides out the forward pass also accumulates, per query, the covariance of values
// and keys under its attention weights, D[c'][c] = sum_s p_s (v_s[c'] - out[c']) k_s[c].  The query gradient is
//   dq[c] = sum_s p_s (go . v_s - go . out) k_s[c] = sum_c' go[c'] D[c'][c],
// so the backward pass needs no pass over the keys for it (it was 47 % of the backward kernel).  D is accumulated as
// shifted-data moments around the plain mean vbar of the slice's values, D = sum p (v - vbar) k - (out - vbar) sum p k,
// which keeps the cancellation error proportional to the spread of v rather than to its magnitude (a cheaper pivot, the
// slice's first value, was 10x less accurate on the whole-network gradient of the ill-conditioned residual=None
// configuration).  The per-key products u[c'][c] = (v[c'] - vbar[c']) k[c] are formed once per key and stored in its LDS
// record {k[C], v[C], u[C*C]} (padded to 16 bytes), so that a (query, key) pair costs C*C + C extra FMAs and nothing else.
// aux[n][h][t][C*C] = D row-major.
template <int C, bool AUX>
struct FwdRec {
    static constexpr int KV = AUX ? ((2 * C + C * C + 3) & ~3) : 2 * C;
};

// The AttentionBlock's two 1x1 convolutions (reference networks/cnn.py:212-240: qkv = Conv1x1(BN(x)), proj_out(attention) [+ skip])
// for the FUSED forward kernel: a block owns whole images (SPB % H == 0), forms q / k / v of its tokens from the normalised input while
// it stages them, and applies the output projection, the residual sum and the next BatchNorm's partial sums to its tokens at the end.
struct AttnStage {
    const float* x;         // [N][T][HC] the block's input, channels-last
    const float* scale;     // [HC] BatchNorm affine in front of the qkv convolution, or NULL
    const float* shift;
    const float* wqkv;      // [HC][3 HC]  (HWIO of the 1x1 kernel)
    const float* wproj;     // [HC][HC]
    const float* residual;  // [N][T][HC] or NULL
    float* qkv;             // [N][T][3 HC] written for a three-launch backward pass, or NULL
    float* y;               // [N][T][HC]
    double* stat_partial;   // [2][HC][gridDim.x] per-channel sum / sum of squares of y, or NULL (bit 0 set: statistic slots, common.h)
    BnFold fold;            // fold.slots != NULL: the BatchNorm in front of the qkv convolution folded into this launch (scale / shift unused)
};

// n floats global -> LDS, 16 bytes per lane when the source allows; eight loads of a thread in flight at once (a load -> store loop
// pays one memory round trip per trip: 24 of them for the 48 KiB of a 64-wide block's qkv weights on 128 threads)
__device__ __forceinline__ void stage_weights(float* __restrict__ dst, const float* __restrict__ src, int n) {
    if ((n & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        const int step = blockDim.x * 4;
        int i0 = threadIdx.x * 4;
        for (; i0 + 7 * step < n; i0 += 8 * step) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(src + i0 + u * step);
#pragma unroll
            for (int u = 0; u < 8; ++u) *reinterpret_cast<float4*>(dst + i0 + u * step) = v[u];
        }
        for (; i0 < n; i0 += step) *reinterpret_cast<float4*>(dst + i0) = *reinterpret_cast<const float4*>(src + i0);
    } else {
        for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
    }
}

// the block's n = ntok * HC input values (contiguous in [N][T][HC]) -> LDS, through the BatchNorm affine when there is one
__device__ __forceinline__ void stage_input_tile(float* __restrict__ xs, const float* __restrict__ xg, int nx, int HC,
                                                 const float* __restrict__ scale, const float* __restrict__ shift) {
    if ((HC & 3) == 0) {
        for (int i = threadIdx.x * 4; i < nx; i += blockDim.x * 4) {
            float4 v = *reinterpret_cast<const float4*>(xg + i);
            if (scale) {
                const int c0 = i % HC;
                const float4 a = *reinterpret_cast<const float4*>(scale + c0), b = *reinterpret_cast<const float4*>(shift + c0);
                v.x = fmaf(v.x, a.x, b.x), v.y = fmaf(v.y, a.y, b.y), v.z = fmaf(v.z, a.z, b.z), v.w = fmaf(v.w, a.w, b.w);
            }
            *reinterpret_cast<float4*>(xs + i) = v;
        }
    } else {
        for (int i = threadIdx.x; i < nx; i += blockDim.x) {
            float v = xg[i];
            if (scale) v = fmaf(v, scale[i % HC], shift[i % HC]);
            xs[i] = v;
        }
    }
}

// out(tok, col) = sum_ci in[tok][ci] * W[ci][col] over the block's ntok tokens (LDS, row stride Cin, ntok % 4 == 0) with W [Cin][ncols]
// in LDS as well: a work item is 4 consecutive tokens x NC adjacent columns, columns fastest (the lanes of a wave read W rows
// conflict-free and one token row as a broadcast).  store(tok0, col0, a[4][NC]) receives the sums.
// The token rows may also lie in global memory (a wave's lanes read the same row: one L1 line per load); sc / sh != NULL: row values
// pass through x * sc[ci] + sh[ci] first when AFF (the BatchNorm affine the LDS tile would have applied).
template <int NC, bool AFF = false, class Store>
__device__ __forceinline__ void token_gemm(const float* __restrict__ in, int ntok, int Cin, const float* __restrict__ W, int ncols,
                                           Store&& store, const float* __restrict__ sc = nullptr, const float* __restrict__ sh = nullptr) {
    const int cg = ncols / NC;
    const int items = (ntok >> 2) * cg;
    const float inv_cg = 1.0f / (float)cg;
    for (int e = threadIdx.x; e < items; e += blockDim.x) {
        const int tg = adiv(e, cg, inv_cg), col = (e - tg * cg) * NC;
        const float* r = in + (size_t)tg * 4 * Cin;
        const float* w = W + col;
        float a[4][NC];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < NC; ++j) a[t][j] = 0.f;
        if ((Cin & 3) == 0) {
#pragma unroll 2
            for (int ci = 0; ci < Cin; ci += 4) {
                float wv[4][NC];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < NC; ++j) wv[u][j] = w[(ci + u) * ncols + j];
                float4 av = make_float4(1.f, 1.f, 1.f, 1.f), bv = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (AFF) av = *reinterpret_cast<const float4*>(sc + ci), bv = *reinterpret_cast<const float4*>(sh + ci);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    float4 xv = *reinterpret_cast<const float4*>(r + t * Cin + ci);
                    if constexpr (AFF) xv.x = fmaf(xv.x, av.x, bv.x), xv.y = fmaf(xv.y, av.y, bv.y), xv.z = fmaf(xv.z, av.z, bv.z), xv.w = fmaf(xv.w, av.w, bv.w);
#pragma unroll
                    for (int j = 0; j < NC; ++j) {
                        a[t][j] = fmaf(xv.x, wv[0][j], a[t][j]);
                        a[t][j] = fmaf(xv.y, wv[1][j], a[t][j]);
                        a[t][j] = fmaf(xv.z, wv[2][j], a[t][j]);
                        a[t][j] = fmaf(xv.w, wv[3][j], a[t][j]);
                    }
                }
            }
        } else {
            for (int ci = 0; ci < Cin; ++ci) {
                const float a1 = AFF ? sc[ci] : 1.f, b1 = AFF ? sh[ci] : 0.f;
                float xr[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) xr[t] = AFF ? fmaf(r[t * Cin + ci], a1, b1) : r[t * Cin + ci];
#pragma unroll
                for (int j = 0; j < NC; ++j) {
                    const float wv = w[ci * ncols + j];
#pragma unroll
                    for (int t = 0; t < 4; ++t) a[t][j] = fmaf(xr[t], wv, a[t][j]);
                }
            }
        }
        store(tg * 4, col, a);
    }
}

// q / k / v of the block's tokens from the normalised input tile: q (times qmul) -> qd[(slice-major token) * qstride + c] when QREC (the
// backward kernel's query records) or qd [tok][HC] otherwise, k / v -> the slices' records, all three -> global when gq != NULL
template <int C, int KVS, int NC, bool QREC, bool AFF = false>
__device__ __forceinline__ void stage_qkv(float* __restrict__ kvrec, float* __restrict__ qd, int qstride, float qmul,
                                          const float* __restrict__ xs, const float* __restrict__ wl, int ntok, int T, int H,
                                          float* __restrict__ gq, const float* __restrict__ sc = nullptr,
                                          const float* __restrict__ sh = nullptr) {
    const int HC = H * C, W3 = 3 * HC;
    token_gemm<NC, AFF>(xs, ntok, HC, wl, W3, [&](int t0_, int col0, const float (&a)[4][NC]) {
        const int img = t0_ / T, tt = t0_ - img * T;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int col = col0 + j;
            const int part = col / HC, hc = col - part * HC;
            const int h_ = hc / C, c_ = hc - h_ * C;
            float* dst;
            int stride;
            float mul = 1.f;
            if (part == 0) {
                if constexpr (QREC) {
                    dst = qd + ((size_t)(img * H + h_) * T + tt) * qstride + c_;
                    stride = qstride;
                } else {
                    dst = qd + (size_t)t0_ * HC + hc;
                    stride = HC;
                }
                mul = qmul;
            } else {
                dst = kvrec + ((size_t)(img * H + h_) * T + tt) * KVS + (part - 1) * C + c_;
                stride = KVS;
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) dst[t * stride] = a[t][j] * mul;
        }
        if (gq) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float* g = gq + (size_t)(t0_ + t) * W3 + col0;
                if constexpr (NC == 2) *reinterpret_cast<float2*>(g) = make_float2(a[t][0], a[t][1]);
                else g[0] = a[t][0];
            }
        }
    }, sc, sh);
}

// Per-block partial of per-column double sums carried by threads whose NC columns are fixed (column group = thread % (HC / NC)):
// lanes of a wave (shuffles over the lane bits above HC / NC), waves (LDS, red [waves][2][HC] doubles), then partial [2][HC][gridDim.x].
// All threads of the block call it.
template <int NC>
__device__ __forceinline__ void column_sums_to_partial(double (&s1)[NC], double (&s2)[NC], double* __restrict__ red, int HC,
                                                       double* __restrict__ partial) {
    const int cg = HC / NC;
    for (int m = cg; m < 64; m <<= 1) {
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            s1[j] += __shfl_xor(s1[j], m, 64);
            s2[j] += __shfl_xor(s2[j], m, 64);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (lane < cg) {
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            red[(wave * 2 + 0) * HC + lane * NC + j] = s1[j];
            red[(wave * 2 + 1) * HC + lane * NC + j] = s2[j];
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < 2 * HC) {
        const int which = threadIdx.x / HC, cc = threadIdx.x - which * HC;
        double t = red[which * HC + cc];
        for (int w = 1; w < nw; ++w) t += red[(w * 2 + which) * HC + cc];
        bn_stat_out(partial, which, HC, cc, gridDim.x, blockIdx.x, t);
    }
}

// output projection + residual of the block's tokens (attention output in xs, wproj in wl); a thread's NC columns are fixed
// (blockDim.x % (HC / NC) == 0), so it also carries their sums for the next BatchNorm: lanes of a wave (shuffles over the lane bits
// above HC / NC), waves (LDS, red [waves][2][HC] doubles), one partial per block
template <int NC>
__device__ __forceinline__ void stage_project(const float* __restrict__ xs, const float* __restrict__ wl, double* __restrict__ red,
                                              int ntok, int HC, float* __restrict__ yg, const float* __restrict__ rg,
                                              double* __restrict__ stat_partial) {
    double s1[NC], s2[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) s1[j] = s2[j] = 0.;
    token_gemm<NC>(xs, ntok, HC, wl, HC, [&](int t0_, int col0, const float (&a)[4][NC]) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const size_t o = (size_t)(t0_ + t) * HC + col0;
            float v[NC];
#pragma unroll
            for (int j = 0; j < NC; ++j) v[j] = a[t][j];
            if (rg) {
                if constexpr (NC == 2) {
                    const float2 rv = *reinterpret_cast<const float2*>(rg + o);
                    v[0] += rv.x, v[1] += rv.y;
                } else v[0] += rg[o];
            }
            if constexpr (NC == 2) *reinterpret_cast<float2*>(yg + o) = make_float2(v[0], v[1]);
            else yg[o] = v[0];
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                s1[j] += (double)v[j];
                s2[j] += (double)v[j] * (double)v[j];
            }
        }
    });
    if (stat_partial) column_sums_to_partial<NC>(s1, s2, red, HC, stat_partial);
}

template <int C, int QPT, bool AUX, bool FUSED>
__device__ __forceinline__ void attn_fwd_body(float* __restrict__ sm, const float* __restrict__ qkv, int N, int T, int H, int SPB,
                                              float* __restrict__ out, float* __restrict__ lse, float* __restrict__ aux,
                                              float qk_scale, const AttnStage& sg) {
    constexpr int KVS = FwdRec<C, AUX>::KV;
    const int HC = H * C, W3 = 3 * HC;
    const int TPS = T / QPT;  // threads per slice
    const long total = (long)N * H;
    const long slice0 = (long)blockIdx.x * SPB;
    const int nsl = (int)min((long)SPB, total - slice0);
    // FUSED: LDS = {k, v (, u)} records [SPB][T][KVS] | q [SPB / H][T][HC] | normalised input, later the attention output [SPB / H][T][HC]
    //              | the two 1x1 kernels' weights
    float* qs = sm + (size_t)SPB * T * KVS;
    float* xs = qs + (size_t)SPB * T * C;
    float* wl = xs + (size_t)SPB * T * C;  // [HC][3 HC] qkv weights
    float* wpl = wl + 3 * HC * HC;         // [HC][HC] projection weights (all weight loads fly with the input tile's)
    const int ntok = FUSED ? nsl / H * T : 0;                 // slice0 % H == 0 and (N H) % H == 0: whole images
    const size_t tok0 = FUSED ? (size_t)(slice0 / H) * T : 0;  // first token of the block in [N][T]
    if constexpr (FUSED) {
        if (sg.fold.slots) {   // every block turns the statistic slots into the affine itself (common.h); HC <= 32 here
            __shared__ __align__(16) float bn_tab[2][32];
            bn_fold_prologue(sg.fold, HC, bn_tab[0], bn_tab[1], blockIdx.x == 0);
            stage_input_tile(xs, sg.x + tok0 * HC, ntok * HC, HC, bn_tab[0], bn_tab[1]);
        } else {
            stage_input_tile(xs, sg.x + tok0 * HC, ntok * HC, HC, sg.scale, sg.shift);
        }
        stage_weights(wl, sg.wqkv, HC * W3);
        stage_weights(wpl, sg.wproj, HC * HC);
        __syncthreads();
        float* gq = sg.qkv ? sg.qkv + tok0 * W3 : nullptr;  // (NULL: the backward kernel forms q / k / v again from x)
        if (HC & 1) stage_qkv<C, KVS, 1, false>(sm, qs, 0, 1.f, xs, wl, ntok, T, H, gq);
        else stage_qkv<C, KVS, 2, false>(sm, qs, 0, 1.f, xs, wl, ntok, T, H, gq);
    } else {
        stage_kv<C, KVS>(sm, qkv, slice0, nsl, T, H);
    }
    __syncthreads();
    const int sl = adiv((int)threadIdx.x, TPS, 1.0f / (float)TPS);
    const bool active = sl < nsl;
    const int t0 = (threadIdx.x - sl * TPS) * QPT;
    long n;
    int h;
    {
        const int bn0 = (int)(slice0 / H), bh0 = (int)(slice0 - (long)bn0 * H);  // block-uniform (scalar unit)
        const int hs = bh0 + sl;
        const int dn = adiv(hs, H, 1.0f / (float)H);
        n = bn0 + dn;
        h = hs - dn * H;
    }
    // scores are kept in the log2 domain (q pre-multiplied by log2(e)/C) so that exp is a bare v_exp_f32
    const float inv_c = LOG2E * qk_scale;  // qk_scale = 1/C for the CNN's QKVAttention, 1/sqrt(C) for nn.MultiheadAttention
    float* kv = sm + (size_t)(active ? sl : 0) * T * KVS;

    constexpr int NA = AUX ? C * C + C : 1;
    float q[QPT][C], acc[QPT][C], mx[QPT], l[QPT], am[QPT][NA];
    // Slice-wide quantities, each from one value per lane combined over the slice's lanes (every lane walks all keys only
    // for slice sizes slice_reduce cannot split): vbar (AUX) = plain mean of the slice's values, the pivot of the
    // shifted-data covariance; kmax / kmin (C == 1) = the extreme keys, which give the row maxima in closed form.
    __shared__ float red[4];
    const bool fast = slice_reduce_ok(TPS);
    float vbar[C], kmax = -INFINITY, kmin = INFINITY, knorm = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) vbar[c] = 0.f;
    if (fast) {
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            const float* r = kv + (size_t)(t0 + i) * KVS;
            if constexpr (AUX) {
#pragma unroll
                for (int c = 0; c < C; ++c) vbar[c] += r[C + c];
            }
            if constexpr (C == 1) {
                kmax = fmaxf(kmax, r[0]);
                kmin = fminf(kmin, r[0]);
            } else {
                float s2 = 0.f;
#pragma unroll
                for (int c = 0; c < C; ++c) s2 = fmaf(r[c], r[c], s2);
                knorm = fmaxf(knorm, s2);
            }
        }
        if constexpr (C > 1) knorm = __builtin_sqrtf(slice_reduce<1>(knorm, TPS, red));
        if constexpr (AUX) {
#pragma unroll
            for (int c = 0; c < C; ++c) vbar[c] = slice_reduce<0>(vbar[c], TPS, red);
        }
        if constexpr (C == 1) {
            kmax = slice_reduce<1>(kmax, TPS, red);
            kmin = slice_reduce<2>(kmin, TPS, red);
        }
    } else if (AUX || C == 1) {
        for (int s = 0; s < T; ++s) {
            const float* r = kv + (size_t)s * KVS;
            if constexpr (AUX) {
#pragma unroll
                for (int c = 0; c < C; ++c) vbar[c] += r[C + c];
            }
            if constexpr (C == 1) {
                kmax = fmaxf(kmax, r[0]);
                kmin = fminf(kmin, r[0]);
            }
        }
    }
    if constexpr (AUX) {
        const float inv_t = 1.f / (float)T;
#pragma unroll
        for (int c = 0; c < C; ++c) vbar[c] *= inv_t;
        // the per-key products of this lane's own tokens
        if (active) {
#pragma unroll
            for (int i = 0; i < QPT; ++i) {
                float* r = kv + (size_t)(t0 + i) * KVS;
#pragma unroll
                for (int c2 = 0; c2 < C; ++c2)
#pragma unroll
                    for (int c = 0; c < C; ++c) r[2 * C + c2 * C + c] = (r[C + c2] - vbar[c2]) * r[c];
            }
        }
        __syncthreads();
    }
    if constexpr (!FUSED) {
        if (!active) return;  // idle lanes stayed for the block barriers above; they read slice 0 and wrote nothing
    }
    const int img_l = FUSED ? (int)(n - slice0 / H) : 0;  // the lane's image within the block
    if (!FUSED || active) {
#pragma unroll
    for (int i = 0; i < QPT; ++i) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            if constexpr (FUSED) q[i][c] = qs[((size_t)img_l * T + t0 + i) * HC + h * C + c] * inv_c;
            else q[i][c] = qkv[(n * T + t0 + i) * W3 + h * C + c] * inv_c;
            acc[i][c] = 0.f;
        }
#pragma unroll
        for (int a = 0; a < NA; ++a) am[i][a] = 0.f;
        l[i] = 0.f;
        mx[i] = -INFINITY;
    }
    // scores of two queries at a time as one packed FMA (the compiler does not pair them by itself: it folds the
    // subtraction of the maximum into a source-negation modifier, which v_pk_fma_f32 would have to apply to both halves)
    typedef float f2 __attribute__((ext_vector_type(2)));
    constexpr int QP = OTVAE_ATTN_PACK ? QPT / 2 : 0;
    f2 q2[QP > 0 ? QP : 1][C], nm2[QP > 0 ? QP : 1];
#pragma unroll
    for (int j = 0; j < QP; ++j)
#pragma unroll
        for (int c = 0; c < C; ++c) q2[j][c] = (f2){q[2 * j][c], q[2 * j + 1][c]};
    // C >= 2: |q . k| <= |q| max_s |k_s|.  When twice that bound stays below ATTN_BOUND_GAP for every query of the wave the bound
    // itself serves as the row "maximum" (any m >= max_s score with sum_s 2^(score - m) > 0 gives the same softmax and the same
    // LSE) and the pass over the keys that computes the exact one is skipped
    bool bounded = false;
    if constexpr (C > 1) {
        if (fast) {
            float qn2 = 0.f;
#pragma unroll
            for (int i = 0; i < QPT; ++i) {
                float s2 = 0.f;
#pragma unroll
                for (int c = 0; c < C; ++c) s2 = fmaf(q[i][c], q[i][c], s2);
                mx[i] = __builtin_sqrtf(s2) * knorm * 1.0001f;
                qn2 = fmaxf(qn2, mx[i]);
            }
            bounded = __all(2.f * qn2 < ATTN_BOUND_GAP);
        }
    }
    if constexpr (C == 1) {
#pragma unroll
        for (int i = 0; i < QPT; ++i) mx[i] = fmaxf(q[i][0] * kmax, q[i][0] * kmin);
    } else if (bounded) {
        // mx holds the bound
    } else if constexpr (QP > 0) {
#pragma unroll
        for (int i = 0; i < QPT; ++i) mx[i] = -INFINITY;
#pragma unroll 2
        for (int s = 0; s < T; ++s) {
            const float* r = kv + s * KVS;
#pragma unroll
            for (int j = 0; j < QP; ++j) {
                f2 d = q2[j][0] * (f2){r[0], r[0]};
#pragma unroll
                for (int c = 1; c < C; ++c) d = __builtin_elementwise_fma(q2[j][c], (f2){r[c], r[c]}, d);
                mx[2 * j] = fmaxf(mx[2 * j], d.x);
                mx[2 * j + 1] = fmaxf(mx[2 * j + 1], d.y);
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < QPT; ++i) mx[i] = -INFINITY;
#pragma unroll 2
        for (int s = 0; s < T; ++s) {
            const float* r = kv + s * KVS;
#pragma unroll
            for (int i = 0; i < QPT; ++i) mx[i] = fmaxf(mx[i], dotc<C>(q[i], r));
        }
    }
#pragma unroll
    for (int j = 0; j < QP; ++j) nm2[j] = (f2){-mx[2 * j], -mx[2 * j + 1]};
#pragma unroll 4
    for (int s = 0; s < T; ++s) {
        const float* r = kv + s * KVS;
        float kk[C], vv[C], uu[AUX ? C * C : 1];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            kk[c] = r[c];
            vv[c] = r[C + c];
        }
        if constexpr (AUX) {
#pragma unroll
            for (int a = 0; a < C * C; ++a) uu[a] = r[2 * C + a];
        }
        float scs[QPT];
        if constexpr (QP > 0) {
#pragma unroll
            for (int j = 0; j < QP; ++j) {
                f2 sc = nm2[j];
#pragma unroll
                for (int c = 0; c < C; ++c) sc = __builtin_elementwise_fma(q2[j][c], (f2){kk[c], kk[c]}, sc);
                scs[2 * j] = sc.x;
                scs[2 * j + 1] = sc.y;
            }
        } else {
#pragma unroll
            for (int i = 0; i < QPT; ++i) {
                float sc = -mx[i];
#pragma unroll
                for (int c = 0; c < C; ++c) sc = fmaf(q[i][c], kk[c], sc);
                scs[i] = sc;
            }
        }
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            const float p = EXP2(scs[i]);
            l[i] += p;
#pragma unroll
            for (int c = 0; c < C; ++c) acc[i][c] = fmaf(p, vv[c], acc[i][c]);
            if constexpr (AUX) {
#pragma unroll
                for (int c = 0; c < C; ++c) am[i][C * C + c] = fmaf(p, kk[c], am[i][C * C + c]);
#pragma unroll
                for (int a = 0; a < C * C; ++a) am[i][a] = fmaf(p, uu[a], am[i][a]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < QPT; ++i) {
        const float rl = 1.f / l[i];
        float* o = out + (n * T + t0 + i) * HC + h * C;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            o[c] = acc[i][c] * rl;
            if constexpr (FUSED) xs[((size_t)img_l * T + t0 + i) * HC + h * C + c] = acc[i][c] * rl;  // (the input tile is dead by now)
        }
        lse[(n * H + h) * T + t0 + i] = mx[i] * LN2 + __logf(l[i]);  // natural-log LSE
        if constexpr (AUX) {
            float* ao = aux + (((size_t)n * H + h) * T + t0 + i) * (C * C);
#pragma unroll
            for (int c2 = 0; c2 < C; ++c2)
#pragma unroll
                for (int c = 0; c < C; ++c)
                    ao[c2 * C + c] = (am[i][c2 * C + c] - (acc[i][c2] * rl - vbar[c2]) * am[i][C * C + c]) * rl;
        }
    }
    }  // active
    if constexpr (FUSED) {
        __syncthreads();
        float* yg = sg.y + tok0 * HC;
        const float* rg = sg.residual ? sg.residual + tok0 * HC : nullptr;
        double* red = reinterpret_cast<double*>(qs);  // (q has been in registers since the barrier behind the staging)
        if (HC & 1) stage_project<1>(xs, wpl, red, ntok, HC, yg, rg, sg.stat_partial);
        else stage_project<2>(xs, wpl, red, ntok, HC, yg, rg, sg.stat_partial);
    }
}

template <int C, int QPT, bool AUX>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ qkv, int N, int T, int H, int SPB,
                                                       float* __restrict__ out, float* __restrict__ lse,
                                                       float* __restrict__ aux, float qk_scale) {
    extern __shared__ __align__(16) float sm[];
    attn_fwd_body<C, QPT, AUX, false>(sm, qkv, N, T, H, SPB, out, lse, aux, qk_scale, AttnStage{});
}

template <int C, int QPT, bool AUX>
__global__ __launch_bounds__(256) void attn_stage_fwd_kernel(AttnStage sg, int N, int T, int H, int SPB, float* __restrict__ out,
                                                             float* __restrict__ lse, float* __restrict__ aux, float qk_scale) {
    extern __shared__ __align__(16) float sm[];
    attn_fwd_body<C, QPT, AUX, true>(sm, nullptr, N, T, H, SPB, out, lse, aux, qk_scale, sg);
}

// The backward pass of the AttentionBlock's three stages for the FUSED backward kernel: the attention output's gradient is formed from
// the block output's gradient (gout = gy . wproj^T) while the records are staged, and the gradient of the normalised input
// (gv = dqkv . wqkv^T) with its BatchNorm-backward sums leaves in the epilogue; dqkv is still written (the qkv weight gradient reads it).
struct AttnStageBwd {
    const float* gy;      // [N][T][HC] gradient of the block's output
    const float* wproj;   // [HC][HC]
    const float* wqkv;    // [HC][3 HC]
    const float* x;       // [N][T][HC] the block's input (read for the BatchNorm sums only)
    const float* mean;    // [HC] batch statistics of x, or NULL (no BatchNorm in front of the qkv convolution)
    const float* invstd;
    float* gv;            // [N][T][HC] gradient of the qkv convolution's (normalised) input
    double* bn_partial;   // [2][HC][gridDim.x]: sum gv, sum gv * xhat per channel, or NULL
    const float* scale;   // qkv == NULL (the forward kernel did not write it): the BatchNorm affine [HC] (or NULL) with which q / k / v
    const float* shift;   // are formed again from x
    int direct;           // != 0: no LDS tiles of gy and x, the token products read their rows from global memory (the records alone then
                          // leave room for a third workgroup per CU at 256 tokens x 4 heads: 88 -> 94 us for the pair loops otherwise)
};

// n = rows * cols floats global [rows][cols] -> LDS [cols][rows]
__device__ __forceinline__ void stage_weights_t(float* __restrict__ dst, const float* __restrict__ src, int rows, int cols) {
    const int n = rows * cols;
    const float inv_cols = 1.0f / (float)cols;
    for (int i0 = threadIdx.x; i0 < n; i0 += 8 * blockDim.x) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * blockDim.x;
            v[u] = i < n ? src[i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * blockDim.x;
            if (i < n) {
                const int r = adiv(i, cols, inv_cols), c = i - r * cols;
                dst[c * rows + r] = v[u];
            }
        }
    }
}

template <int C, int QPT, bool AUX, bool FUSED>
__device__ __forceinline__ void attn_bwd_body(float* __restrict__ sm, const float* __restrict__ qkv, const float* __restrict__ out,
                                              const float* __restrict__ lse_g, const float* __restrict__ gout,
                                              const float* __restrict__ aux, int N, int T, int H, int SPB,
                                              float* __restrict__ gqkv, float qk_scale, const AttnStageBwd& sg) {
    constexpr int RKV = Rec<C>::KV, RQG = Rec<C>::QG;
    const int HC = H * C, W3 = 3 * HC;
    const int TPS = T / QPT;
    const long total = (long)N * H;
    const long slice0 = (long)blockIdx.x * SPB;
    const int nsl = (int)min((long)SPB, total - slice0);
    float* s_kv = sm;
    float* s_qg = sm + (size_t)SPB * T * RKV;
    // FUSED: behind the records the 1x1 kernels' weights, the reduction scratch and (unless sg.direct) the gy and x tiles; the records'
    // place is reused for the dqkv tile [SPB / H][T][3 HC] of the epilogue
    float* wl = s_qg + (size_t)SPB * T * RQG;  // wproj^T [HC][HC]
    float* wql = wl + HC * HC;                 // wqkv^T [3 HC][HC]
    float* wq3 = wql + 3 * HC * HC;            // wqkv [HC][3 HC] (for q / k / v formed again here: qkv == NULL)
    float* s_red = wl + ((7 * HC * HC + 3) & ~3);  // [4 waves][2][HC] doubles of the BatchNorm sums' reduction
    float* s_gy = s_red + 16 * HC;             // gy tile and normalised input tile [SPB / H][T][HC] each -- not allocated when sg.direct
    float* s_x = s_gy + (size_t)SPB * T * C;
    const bool recompute = FUSED && qkv == nullptr;
    const bool direct = FUSED && sg.direct != 0;
    const int ntok = FUSED ? nsl / H * T : 0;
    const size_t tok0 = FUSED ? (size_t)(slice0 / H) * T : 0;
    const float inv_c = qk_scale;
    if (!recompute) stage_kv<C>(s_kv, qkv, slice0, nsl, T, H);
    if constexpr (FUSED) {
        if (recompute) {
            if (!direct) stage_input_tile(s_x, sg.x + tok0 * HC, ntok * HC, HC, sg.scale, sg.shift);
            stage_weights(wq3, sg.wqkv, HC * W3);
        }
        if (!direct) stage_weights(s_gy, sg.gy + tok0 * HC, ntok * HC);
        stage_weights_t(wl, sg.wproj, HC, HC);  // wl[co][ci] = wproj[ci][co]
        stage_weights_t(wql, sg.wqkv, HC, W3);  // wql[col][ci] = wqkv[ci][col] for the epilogue
        __syncthreads();
        // gout[tok][ci] = sum_co gy[tok][co] wproj[ci][co] -> the gout slot of the query records
        auto put = [&](int t0_, int col0, const auto& a) {
            constexpr int NC_ = sizeof(a[0]) / sizeof(float);
            const int img = t0_ / T, tt = t0_ - img * T;
#pragma unroll
            for (int j = 0; j < NC_; ++j) {
                const int hc = col0 + j, h_ = hc / C, c_ = hc - h_ * C;
                float* dst = s_qg + ((size_t)(img * H + h_) * T + tt) * RQG + C + c_;
#pragma unroll
                for (int t = 0; t < 4; ++t) dst[t * RQG] = a[t][j];
            }
        };
        const float* gy_rows = direct ? sg.gy + tok0 * HC : s_gy;
        if (HC & 1) token_gemm<1>(gy_rows, ntok, HC, wl, HC, put);
        else token_gemm<2>(gy_rows, ntok, HC, wl, HC, put);
        if (recompute) {  // q * scale -> the q slot of the query records, k / v -> the key records
            const float* x_rows = direct ? sg.x + tok0 * HC : s_x;
            if (direct && sg.scale) {
                if (HC & 1) stage_qkv<C, RKV, 1, true, true>(s_kv, s_qg, RQG, inv_c, x_rows, wq3, ntok, T, H, nullptr, sg.scale, sg.shift);
                else stage_qkv<C, RKV, 2, true, true>(s_kv, s_qg, RQG, inv_c, x_rows, wq3, ntok, T, H, nullptr, sg.scale, sg.shift);
            } else {
                if (HC & 1) stage_qkv<C, RKV, 1, true>(s_kv, s_qg, RQG, inv_c, x_rows, wq3, ntok, T, H, nullptr);
                else stage_qkv<C, RKV, 2, true>(s_kv, s_qg, RQG, inv_c, x_rows, wq3, ntok, T, H, nullptr);
            }
        }
        __syncthreads();
    }
    // query records {q/C, gout, lse, delta = sum_c gout*out}
    const float inv_t = 1.0f / (float)T, inv_hq = 1.0f / (float)H;
    const int qn0 = (int)(slice0 / H), qh0 = (int)(slice0 - (long)qn0 * H);
#pragma unroll 2
    for (int it = threadIdx.x; it < nsl * T; it += blockDim.x) {
        const int sl = adiv(it, T, inv_t), t = it - sl * T;
        const int hs = qh0 + sl;
        const int dn = adiv(hs, H, inv_hq);
        const long n = qn0 + dn;
        const int h = hs - dn * H;
        float* r = s_qg + (size_t)it * RQG;
        const float* qp = qkv + (n * T + t) * W3 + h * C;
        const float* gp = FUSED ? r + C : gout + (n * T + t) * HC + h * C;
        const float* op = out + (n * T + t) * HC + h * C;
        float d = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float g = gp[c];
            if (!recompute) r[c] = qp[c] * inv_c;
            if constexpr (!FUSED) r[C + c] = g;
            d = fmaf(g, op[c], d);
        }
        r[2 * C] = lse_g[(n * H + h) * T + t] * LOG2E;  // log2-domain LSE
        r[2 * C + 1] = d;
    }
    __syncthreads();
    const int sl = adiv((int)threadIdx.x, TPS, 1.0f / (float)TPS);
    const bool active = sl < nsl;
    if constexpr (!FUSED) {
        if (!active) return;
    }
    const int t0 = (threadIdx.x - sl * TPS) * QPT;
    float res[FUSED ? QPT : 1][3][C];  // FUSED: the lane's dq | dk | dv rows, kept past the barrier that frees the records
    long n;
    int h;
    {
        const int bn0 = (int)(slice0 / H), bh0 = (int)(slice0 - (long)bn0 * H);  // block-uniform (scalar unit)
        const int hs = bh0 + sl;
        const int dn = adiv(hs, H, 1.0f / (float)H);
        n = bn0 + dn;
        h = hs - dn * H;
    }
    // (declared ahead of the `active` sections: phase A and phase B are separate sections since round 4, with a block barrier between
    // them that EVERY thread reaches; an idle lane's pointers are never dereferenced)
    const float* kv = s_kv + (size_t)sl * T * RKV;
    const float* qg = s_qg + (size_t)sl * T * RQG;
    float dqv[QPT][C];
    constexpr bool PHB = !AUX;   // the pair passes with the consistent delta (phase A, non-AUX branch): bare and one-launch kernels alike
    float dl_new[PHB ? QPT : 1];
    if (!FUSED || active) {

    // ---- phase A: dQ.  With the forward pass's key moments (AUX) it is a few FMAs per query; otherwise this lane's QPT
    // queries against every key
    // (kept in registers: this lane's queries are also its keys of phase B, so the three gradients of a token are
    // stored together at the end -- whole q|k|v rows instead of three strided partial-line passes)
    if constexpr (AUX) {
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            const float* r = qg + (size_t)(t0 + i) * RQG;
            const float* dm = aux + (((size_t)n * H + h) * T + t0 + i) * (C * C);
#pragma unroll
            for (int c = 0; c < C; ++c) {
                float s = 0.f;
#pragma unroll
                for (int c2 = 0; c2 < C; ++c2) s = fmaf(r[C + c2], dm[c2 * C + c], s);
                dqv[i][c] = s * inv_c;
            }
        }
    } else if constexpr (OTVAE_ATTN_PACK && QPT % 2 == 0) {
        // two queries at a time in packed registers (as the forward pass and phase B)
        typedef float f2 __attribute__((ext_vector_type(2)));
        constexpr int QP = QPT / 2;
        f2 q2[QP][C], go2[QP][C], dq2[QP][C], nls2[QP], ndl2[QP];
#pragma unroll
        for (int j = 0; j < QP; ++j) {
            const float* r0 = qg + (size_t)(t0 + 2 * j) * RQG;
            const float* r1 = r0 + RQG;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                q2[j][c] = (f2){r0[c] * LOG2E, r1[c] * LOG2E};
                go2[j][c] = (f2){r0[C + c], r1[C + c]};
                dq2[j][c] = (f2){0.f, 0.f};
            }
            nls2[j] = (f2){-r0[2 * C], -r1[2 * C]};
            ndl2[j] = (f2){-r0[2 * C + 1], -r1[2 * C + 1]};
        }
#pragma unroll 4
        for (int s = 0; s < T; ++s) {
            const float* r = kv + s * RKV;
            f2 kk[C], vv[C];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                kk[c] = (f2){r[c], r[c]};
                vv[c] = (f2){r[C + c], r[C + c]};
            }
#pragma unroll
            for (int j = 0; j < QP; ++j) {
                f2 sc = nls2[j], dp = ndl2[j];
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    sc = __builtin_elementwise_fma(q2[j][c], kk[c], sc);
                    dp = __builtin_elementwise_fma(go2[j][c], vv[c], dp);
                }
                const f2 ds = (f2){EXP2(sc.x), EXP2(sc.y)} * dp;
#pragma unroll
                for (int c = 0; c < C; ++c) dq2[j][c] = __builtin_elementwise_fma(ds, kk[c], dq2[j][c]);
            }
        }
#pragma unroll
        for (int j = 0; j < QP; ++j)
#pragma unroll
            for (int c = 0; c < C; ++c) {
                dqv[2 * j][c] = dq2[j][c].x * inv_c;
                dqv[2 * j + 1][c] = dq2[j][c].y * inv_c;
            }
    } else {
        float q[QPT][C], go[QPT][C], dq[QPT][C], ls[QPT], dl[QPT];
        // CONSISTENT delta (round 4; the bare kernel only, see below).  dS = p (dP - delta) with delta = gout . out taken from the
        // FORWARD pass's output: when a row's softmax is peaked (p* -> 1) the true dP* - delta is (1 - p*) x something, while the
        // rounding of `out` (accumulated by another kernel) and of this pass's dP* are independent, so the difference carries an
        // absolute error eps |dP| instead of eps (1 - p*) |dP| -- and it is the same sign for every key of the row, i.e. an error
        // along sum_s p k_s in dq (and along q_t in every dk).  Measured on the vit.yaml step (tools/diag/vit_ln_grad.py): dq 3.4e-4
        // from the float64 truth where torch's fp32 softmax backward (which subtracts sum_s p dP formed from the SAME p and dP) is
        // at 7.6e-5, and -- the errors being aligned -- a LayerNorm parameter gradient downstream 4.5e-3 off against 2.9e-4.
        // Here the residual r = sum_s p (dP - delta) (zero for a consistent delta), sum_s p and sum_s p k ride along the pass:
        // delta' = delta + r / sum p is what this pass's own p and dP define, dq is corrected by - (r / sum p) sum_s p k (the two
        // terms share their rounding, so the correction cancels the error instead of adding one), and the record's delta is
        // replaced by delta' for phase B.
        constexpr bool FIXD = PHB;   // (= true in this branch)
        float rs[FIXD ? QPT : 1], sp[FIXD ? QPT : 1], bk[FIXD ? QPT : 1][FIXD ? C : 1];
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            const float* r = qg + (size_t)(t0 + i) * RQG;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                q[i][c] = r[c] * LOG2E;
                go[i][c] = r[C + c];
                dq[i][c] = 0.f;
                if constexpr (FIXD) bk[i][c] = 0.f;
            }
            ls[i] = r[2 * C];
            dl[i] = r[2 * C + 1];
            if constexpr (FIXD) rs[i] = sp[i] = 0.f;
        }
#pragma unroll 4
        for (int s = 0; s < T; ++s) {
            const float* r = kv + s * RKV;
            float kk[C], vv[C];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                kk[c] = r[c];
                vv[c] = r[C + c];
            }
#pragma unroll
            for (int i = 0; i < QPT; ++i) {
                // FIXD: the score and dP chains are spelled exactly as phase B spells them (same operands, same order, dP started at
                // zero and delta subtracted afterwards), so that both phases hold bit-identical p and dP: delta' is only
                // consistent with the values it was formed from
                float sc = -ls[i], dp = FIXD ? 0.f : -dl[i];
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    sc = fmaf(q[i][c], kk[c], sc);
                    dp = fmaf(go[i][c], vv[c], dp);
                }
                if constexpr (FIXD) dp -= dl[i];
                const float pp = EXP2(sc);
                const float ds = pp * dp;
#pragma unroll
                for (int c = 0; c < C; ++c) dq[i][c] = fmaf(ds, kk[c], dq[i][c]);
                if constexpr (FIXD) {
                    rs[i] += ds;
                    sp[i] += pp;
#pragma unroll
                    for (int c = 0; c < C; ++c) bk[i][c] = fmaf(pp, kk[c], bk[i][c]);
                }
            }
        }
        if constexpr (FIXD) {
#pragma unroll
            for (int i = 0; i < QPT; ++i) {
                const float corr = rs[i] / sp[i];
#pragma unroll
                for (int c = 0; c < C; ++c) dq[i][c] = fmaf(-corr, bk[i][c], dq[i][c]);
                dl_new[i] = dl[i] + corr;
            }
        }
#pragma unroll
        for (int i = 0; i < QPT; ++i)
#pragma unroll
            for (int c = 0; c < C; ++c) dqv[i][c] = dq[i][c] * inv_c;
    }
    }   // (end of the phase A section)
    if constexpr (PHB) {
        __syncthreads();   // every lane has read its queries' delta
        if (!FUSED || active) {
#pragma unroll
            for (int i = 0; i < QPT; ++i) s_qg[((size_t)sl * T + t0 + i) * RQG + 2 * C + 1] = dl_new[i];
        }
        __syncthreads();   // phase B reads the corrected records
    }
    if (!FUSED || active) {
    // ---- phase B: this lane's QPT keys against every query -> dK, dV.  Two keys at a time in packed registers (spelled
    // out: left to itself the compiler folds the "- lse" / "- delta" into source-negation modifiers of scalar FMAs and
    // packs only the accumulations)
    if constexpr (OTVAE_ATTN_PACK && QPT % 2 == 0) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        constexpr int QP = QPT / 2;
        f2 k2[QP][C], v2[QP][C], dk2[QP][C], dv2[QP][C];
#pragma unroll
        for (int j = 0; j < QP; ++j) {
            const float* r0 = kv + (size_t)(t0 + 2 * j) * RKV;
            const float* r1 = r0 + RKV;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                // (PHB: the bare kernel with a consistent delta keeps the keys unscaled and scales the QUERY by log2 e, as phase A does)
                k2[j][c] = PHB ? (f2){r0[c], r1[c]} : (f2){r0[c] * LOG2E, r1[c] * LOG2E};
                v2[j][c] = (f2){r0[C + c], r1[C + c]};
                dk2[j][c] = dv2[j][c] = (f2){0.f, 0.f};
            }
        }
#pragma unroll 4
        for (int t = 0; t < T; ++t) {
            const float* r = qg + (size_t)t * RQG;
            f2 qq[C], gg[C], ql[PHB ? C : 1];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                qq[c] = (f2){r[c], r[c]};
                gg[c] = (f2){r[C + c], r[C + c]};
                if constexpr (PHB) ql[c] = (f2){r[c] * LOG2E, r[c] * LOG2E};
            }
            const float nls = -r[2 * C], ndl = -r[2 * C + 1];
#pragma unroll
            for (int j = 0; j < QP; ++j) {
                f2 sc = (f2){nls, nls}, dp = PHB ? (f2){0.f, 0.f} : (f2){ndl, ndl};
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    sc = __builtin_elementwise_fma(PHB ? ql[c] : qq[c], k2[j][c], sc);
                    dp = __builtin_elementwise_fma(gg[c], v2[j][c], dp);
                }
                if constexpr (PHB) dp += (f2){ndl, ndl};
                const f2 p = (f2){EXP2(sc.x), EXP2(sc.y)};
                const f2 ds = p * dp;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    dv2[j][c] = __builtin_elementwise_fma(p, gg[c], dv2[j][c]);
                    dk2[j][c] = __builtin_elementwise_fma(ds, qq[c], dk2[j][c]);
                }
            }
        }
        if constexpr (C == 1 && QPT == 4 && !FUSED) {
            // one head of one channel, the whole workgroup on one image (T = 1024): the lane's 4 tokens are 12 consecutive
            // floats q|k|v q|k|v ..., a wave's 3 KiB are contiguous.  They pass through LDS (the staging area is free now)
            // so that every store instruction writes 1 KiB of consecutive addresses, 16 bytes per lane.
            if (H == 1 && T == 1024 && (reinterpret_cast<uintptr_t>(gqkv) & 15) == 0) {
                __syncthreads();  // every wave is done with the key / query records
                float* w = sm + (threadIdx.x >> 6) * (64 * 12);
                float4* wl = reinterpret_cast<float4*>(w + (threadIdx.x & 63) * 12);
                wl[0] = make_float4(dqv[0][0], dk2[0][0].x, dv2[0][0].x, dqv[1][0]);
                wl[1] = make_float4(dk2[0][0].y, dv2[0][0].y, dqv[2][0], dk2[1][0].x);
                wl[2] = make_float4(dv2[1][0].x, dqv[3][0], dk2[1][0].y, dv2[1][0].y);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                float4* o = reinterpret_cast<float4*>(gqkv + (n * T + (t0 - (int)(threadIdx.x & 63) * 4)) * 3);
                const float4* r4 = reinterpret_cast<const float4*>(w);
#pragma unroll
                for (int k = 0; k < 3; ++k) o[k * 64 + (threadIdx.x & 63)] = r4[k * 64 + (threadIdx.x & 63)];
                return;
            }
        }
        if constexpr (FUSED) {
#pragma unroll
            for (int j = 0; j < QP; ++j)
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    res[2 * j][0][c] = dqv[2 * j][c], res[2 * j + 1][0][c] = dqv[2 * j + 1][c];
                    res[2 * j][1][c] = dk2[j][c].x, res[2 * j + 1][1][c] = dk2[j][c].y;
                    res[2 * j][2][c] = dv2[j][c].x, res[2 * j + 1][2][c] = dv2[j][c].y;
                }
        } else {
#pragma unroll
        for (int j = 0; j < QP; ++j) {
            float* o0 = gqkv + (n * T + t0 + 2 * j) * W3 + h * C;
            float* o1 = o0 + W3;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                o0[c] = dqv[2 * j][c];
                o1[c] = dqv[2 * j + 1][c];
                o0[HC + c] = dk2[j][c].x;
                o1[HC + c] = dk2[j][c].y;
                o0[2 * HC + c] = dv2[j][c].x;
                o1[2 * HC + c] = dv2[j][c].y;
            }
        }
        }
    } else {
        float k[QPT][C], v[QPT][C], dk[QPT][C], dv[QPT][C];
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            const float* r = kv + (size_t)(t0 + i) * RKV;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                k[i][c] = PHB ? r[c] : r[c] * LOG2E;   // (PHB: phase A's spelling of the score, see there)
                v[i][c] = r[C + c];
                dk[i][c] = dv[i][c] = 0.f;
            }
        }
#pragma unroll 4
        for (int t = 0; t < T; ++t) {
            const float* r = qg + (size_t)t * RQG;
            float qq[C], gg[C], ql[PHB ? C : 1];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                qq[c] = r[c];
                gg[c] = r[C + c];
                if constexpr (PHB) ql[c] = r[c] * LOG2E;
            }
            const float ls = r[2 * C], dl = r[2 * C + 1];
#pragma unroll
            for (int i = 0; i < QPT; ++i) {
                float sc = -ls, dp = PHB ? 0.f : -dl;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    sc = fmaf(PHB ? ql[c] : qq[c], k[i][c], sc);
                    dp = fmaf(gg[c], v[i][c], dp);
                }
                if constexpr (PHB) dp -= dl;
                const float p = EXP2(sc);
                const float ds = p * dp;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    dv[i][c] = fmaf(p, gg[c], dv[i][c]);
                    dk[i][c] = fmaf(ds, qq[c], dk[i][c]);
                }
            }
        }
        if constexpr (C == 1 && QPT == 4 && !FUSED) {
            // one head of one channel, the whole workgroup on one image (T = 1024): see the packed branch above
            if (H == 1 && T == 1024 && (reinterpret_cast<uintptr_t>(gqkv) & 15) == 0) {
                __syncthreads();  // every wave is done with the key / query records
                float* w = sm + (threadIdx.x >> 6) * (64 * 12);
                float4* wl = reinterpret_cast<float4*>(w + (threadIdx.x & 63) * 12);
                wl[0] = make_float4(dqv[0][0], dk[0][0], dv[0][0], dqv[1][0]);
                wl[1] = make_float4(dk[1][0], dv[1][0], dqv[2][0], dk[2][0]);
                wl[2] = make_float4(dv[2][0], dqv[3][0], dk[3][0], dv[3][0]);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                float4* o = reinterpret_cast<float4*>(gqkv + (n * T + (t0 - (int)(threadIdx.x & 63) * 4)) * 3);
                const float4* r4 = reinterpret_cast<const float4*>(w);
#pragma unroll
                for (int k4 = 0; k4 < 3; ++k4) o[k4 * 64 + (threadIdx.x & 63)] = r4[k4 * 64 + (threadIdx.x & 63)];
                return;
            }
        }
        if constexpr (FUSED) {
#pragma unroll
            for (int i = 0; i < QPT; ++i)
#pragma unroll
                for (int c = 0; c < C; ++c) res[i][0][c] = dqv[i][c], res[i][1][c] = dk[i][c], res[i][2][c] = dv[i][c];
        } else {
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            float* o = gqkv + (n * T + t0 + i) * W3 + h * C;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                o[c] = dqv[i][c];
                o[HC + c] = dk[i][c];
                o[2 * HC + c] = dv[i][c];
            }
        }
        }
    }
    }  // active
    if constexpr (FUSED) {
        __syncthreads();  // every wave is done with the key / query records: their place takes the dqkv tile [ntok][3 HC]
        float* dq_s = sm;
        if (active) {
            const int img_l = (int)(n - slice0 / H);
#pragma unroll
            for (int i = 0; i < QPT; ++i) {
                float* o = dq_s + ((size_t)img_l * T + t0 + i) * W3 + h * C;
#pragma unroll
                for (int c = 0; c < C; ++c) o[c] = res[i][0][c], o[HC + c] = res[i][1][c], o[2 * HC + c] = res[i][2][c];
            }
        }
        __syncthreads();
        // dqkv of the block's tokens is one contiguous piece of [N][T][3 HC]
        {
            float* g = gqkv + tok0 * W3;
            const int nq = ntok * W3;
            if ((nq & 3) == 0 && (reinterpret_cast<uintptr_t>(g) & 15) == 0) {
                for (int i = threadIdx.x * 4; i < nq; i += blockDim.x * 4) *reinterpret_cast<float4*>(g + i) = *reinterpret_cast<const float4*>(dq_s + i);
            } else {
                for (int i = threadIdx.x; i < nq; i += blockDim.x) g[i] = dq_s[i];
            }
        }
        // gv[tok][ci] = sum_col dqkv[tok][col] wqkv[ci][col], with the BatchNorm-backward sums of the thread's columns
        float* gvg = sg.gv + tok0 * HC;
        const float* xg = sg.x ? sg.x + tok0 * HC : nullptr;
        auto project = [&](auto nc_tag) {
            constexpr int NC_ = decltype(nc_tag)::value;
            double s1[NC_], s2[NC_];
#pragma unroll
            for (int j = 0; j < NC_; ++j) s1[j] = s2[j] = 0.;
            float mu[NC_], is[NC_];
            const int col_t = (threadIdx.x % (HC / NC_)) * NC_;  // the thread's columns (blockDim.x % (HC / NC) == 0)
#pragma unroll
            for (int j = 0; j < NC_; ++j) mu[j] = sg.mean ? sg.mean[col_t + j] : 0.f, is[j] = sg.mean ? sg.invstd[col_t + j] : 0.f;
            token_gemm<NC_>(dq_s, ntok, W3, wql, HC, [&](int t0_, int col0, const float (&a)[4][NC_]) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const size_t o = (size_t)(t0_ + t) * HC + col0;
                    if constexpr (NC_ == 2) *reinterpret_cast<float2*>(gvg + o) = make_float2(a[t][0], a[t][1]);
                    else gvg[o] = a[t][0];
                    if (sg.mean) {
#pragma unroll
                        for (int j = 0; j < NC_; ++j) {
                            const float xv = xg[o + j];
                            s1[j] += (double)a[t][j];
                            s2[j] += (double)a[t][j] * (double)((xv - mu[j]) * is[j]);
                        }
                    }
                }
            });
            if (sg.bn_partial) column_sums_to_partial<NC_>(s1, s2, reinterpret_cast<double*>(s_red), HC, sg.bn_partial);
        };
        if (HC & 1) project(std::integral_constant<int, 1>{});
        else project(std::integral_constant<int, 2>{});
    }
}

template <int C, int QPT, bool AUX>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ out,
                                                       const float* __restrict__ lse_g, const float* __restrict__ gout,
                                                       const float* __restrict__ aux, int N, int T, int H, int SPB,
                                                       float* __restrict__ gqkv, float qk_scale) {
    extern __shared__ __align__(16) float sm[];
    attn_bwd_body<C, QPT, AUX, false>(sm, qkv, out, lse_g, gout, aux, N, T, H, SPB, gqkv, qk_scale, AttnStageBwd{});
}

template <int C, int QPT, bool AUX>
__global__ __launch_bounds__(256) void attn_stage_bwd_kernel(AttnStageBwd sg, const float* __restrict__ qkv, const float* __restrict__ out,
                                                             const float* __restrict__ lse_g, const float* __restrict__ aux, int N, int T,
                                                             int H, int SPB, float* __restrict__ gqkv, float qk_scale) {
    extern __shared__ __align__(16) float sm[];
    attn_bwd_body<C, QPT, AUX, true>(sm, qkv, out, lse_g, nullptr, aux, N, T, H, SPB, gqkv, qk_scale, sg);
}


// T == 1 (the 1x1 bottleneck block of the encoder): one key, so the softmax weight is 1 -- out = v, lse = the single score,
// dq = dk = 0 and dv = gout.  Elementwise kernels at the launch floor instead of the staged ones (13 + 18 us there).
__global__ __launch_bounds__(256) void attn_t1_fwd_kernel(const float* __restrict__ qkv, int N, int H, int C, float scale,
                                                          float* __restrict__ out, float* __restrict__ lse) {
    const int HC = H * C;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < N * HC) {
        const int n = i / HC, hc = i - n * HC;
        out[i] = qkv[(size_t)n * 3 * HC + 2 * HC + hc];
    }
    if (i < N * H) {
        const int n = i / H, h = i - n * H;
        const float* q = qkv + (size_t)n * 3 * HC + h * C;
        float s = 0.f;
        for (int c = 0; c < C; ++c) s = fmaf(q[c], q[HC + c], s);
        lse[i] = s * scale;
    }
}

__global__ __launch_bounds__(256) void attn_t1_bwd_kernel(const float* __restrict__ gout, int N, int HC, float* __restrict__ gqkv) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * 3 * HC) return;
    const int n = i / (3 * HC), j = i - n * 3 * HC;
    gqkv[i] = j >= 2 * HC ? gout[(size_t)n * HC + j - 2 * HC] : 0.f;
}

// slices per block: as many as 256 threads cover (T/QPT threads each) and as the LDS budget holds; OTVAE_ATTN_SPB (A/B switch,
// read once) caps it -- smaller workgroups of whole waves
static int pick_qpt(int T, int C) { return (T >= 256 && T % 4 == 0 && C <= 4) ? 4 : 1; }
static int pick_spb(int T, int qpt, int floats_per_key) {
    static const int cap_env = getenv("OTVAE_ATTN_SPB") ? atoi(getenv("OTVAE_ATTN_SPB")) : 0;
    int tps = T / qpt;
    int spb = 256 / tps;
    int cap = ATTN_LDS_FLOATS / (T * floats_per_key);
    if (spb > cap) spb = cap;
    if (cap_env > 0 && spb > cap_env && (cap_env * tps) % 64 == 0) spb = cap_env;
    return spb;
}
// threads of a block: its slices' lanes, in whole waves
static int attn_threads(int spb, int T, int qpt) { return imax(64, (spb * (T / qpt) + 63) / 64 * 64); }

#define ATTN_C_SWITCH(C_, MACRO)                                                                              \
    switch (C_) {                                                                                             \
        case 1: MACRO(1); break;                                                                              \
        case 2: MACRO(2); break;                                                                              \
        case 3: MACRO(3); break;                                                                              \
        case 4: MACRO(4); break;                                                                              \
        case 6: MACRO(6); break;                                                                              \
        case 8: MACRO(8); break;                                                                              \
        case 12: MACRO(12); break;                                                                            \
        case 16: MACRO(16); break;                                                                            \
        case 32: MACRO(32); break;                                                                            \
        default:                                                                                              \
            otvae_set_error("attention: head width C=%d not instantiated (1,2,3,4,6,8,12,16,32)", C_);        \
            return OTVAE_EUNSUPPORTED;                                                                        \
    }

static int attn_check(const char* who, int N, int T, int H, int C, int floats_per_key, int* qpt, int* spb) {
    OTVAE_REQUIRE(N > 0 && T > 0 && H > 0 && C > 0, "%s: bad sizes", who);
    *qpt = pick_qpt(T, C);
    if (T / *qpt > 256) {
        otvae_set_error("%s: T = %d unsupported (T <= 256, or T <= 1024 with T %% 4 == 0 and head width <= 4)", who, T);
        return OTVAE_EUNSUPPORTED;
    }
    *spb = pick_spb(T, *qpt, floats_per_key);
    if (*spb < 1) {
        otvae_set_error("%s: one (image, head) slice of T=%d, C=%d does not fit the 64 KiB LDS slab", who, T, C);
        return OTVAE_EUNSUPPORTED;
    }
    return OTVAE_OK;
}

extern "C" int otvae_attn_fwd_scaled(const float* qkv, int N, int T, int H, int C, float scale, float* out, float* lse, float* aux,
                                     void* stream);
extern "C" int otvae_attn_fwd(const float* qkv, int N, int T, int H, int C, float* out, float* lse, float* aux, void* stream) {
    OTVAE_REQUIRE(C > 0, "otvae_attn_fwd: bad sizes");
    return otvae_attn_fwd_scaled(qkv, N, T, H, C, 1.f / (float)C, out, lse, aux, stream);
}

extern "C" int otvae_attn_fwd_scaled(const float* qkv, int N, int T, int H, int C, float scale, float* out, float* lse, float* aux,
                                     void* stream) {
    OTVAE_REQUIRE(qkv && out && lse, "otvae_attn_fwd: NULL tensor");
    OTVAE_REQUIRE(scale > 0.f, "otvae_attn_fwd: scale must be positive");
    if (T == 1 && N > 0 && H > 0 && C > 0) {  // aux stays unwritten: with T == 1 the backward pass does not read it
        attn_t1_fwd_kernel<<<cdiv((int64_t)N * H * C, 256), 256, 0, (hipStream_t)stream>>>(qkv, N, H, C, scale, out, lse);
        OTVAE_CHECK_LAUNCH("otvae_attn_fwd(T=1)");
        return OTVAE_OK;
    }
    int qpt, spb;
    if (C > 2) aux = nullptr;
    const int rkv = aux ? ((2 * C + C * C + 3) & ~3) : 2 * C;  // FwdRec<C, AUX>::KV
    int rc = attn_check("otvae_attn_fwd", N, T, H, C, rkv, &qpt, &spb);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int grid = cdiv((int64_t)N * H, spb), nthr = attn_threads(spb, T, qpt);
    const size_t lds = (size_t)spb * T * rkv * sizeof(float);
#define FWD_K(CC)                                                                                   \
    do {                                                                                            \
        if (qpt == 4) {                                                                             \
            if constexpr (CC <= 2) {                                                                \
                if (aux) attn_fwd_kernel<CC, 4, true><<<grid, nthr, lds, st>>>(qkv, N, T, H, spb, out, lse, aux, scale); \
                else attn_fwd_kernel<CC, 4, false><<<grid, nthr, lds, st>>>(qkv, N, T, H, spb, out, lse, nullptr, scale); \
            } else if constexpr (CC <= 4) attn_fwd_kernel<CC, 4, false><<<grid, nthr, lds, st>>>(qkv, N, T, H, spb, out, lse, nullptr, scale); \
            else { otvae_set_error("otvae_attn_fwd: T >= 256 with head width %d > 4 unsupported", CC); return OTVAE_EUNSUPPORTED; } \
        } else {                                                                                    \
            if constexpr (CC <= 2) {                                                                \
                if (aux) attn_fwd_kernel<CC, 1, true><<<grid, nthr, lds, st>>>(qkv, N, T, H, spb, out, lse, aux, scale); \
                else attn_fwd_kernel<CC, 1, false><<<grid, nthr, lds, st>>>(qkv, N, T, H, spb, out, lse, nullptr, scale); \
            } else attn_fwd_kernel<CC, 1, false><<<grid, nthr, lds, st>>>(qkv, N, T, H, spb, out, lse, nullptr, scale); \
        }                                                                                           \
    } while (0)
    ATTN_C_SWITCH(C, FWD_K)
#undef FWD_K
    OTVAE_CHECK_LAUNCH("otvae_attn_fwd");
    return OTVAE_OK;
}

// ---- the fused AttentionBlock forward (attn_stage_fwd_kernel)
#define ATTN_STAGE_LDS_FLOATS 24576  // 96 KiB of the CU's 160: above 64 KiB a kernel has to be told (stage_lds_ok), two workgroups still share a CU
template <auto Kernel>
static int stage_lds_ok(size_t lds) {
    static size_t granted = 64 * 1024;  // per kernel instantiation
    if (lds <= granted) return OTVAE_OK;
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             ATTN_STAGE_LDS_FLOATS * sizeof(float));
    if (e != hipSuccess) {
        otvae_set_error("otvae_attn_stage_fwd: %zu bytes of LDS refused: %s", lds, hipGetErrorString(e));
        return OTVAE_ELAUNCH;
    }
    granted = ATTN_STAGE_LDS_FLOATS * sizeof(float);
    return OTVAE_OK;
}
static inline bool attn_aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }
// Launch shape, or OTVAE_EUNSUPPORTED when the stage has to run as three launches: a block must own whole images (spb % H == 0), the token
// GEMMs want T % 4 == 0, the per-column statistics a power-of-two width <= 64.
static int attn_stage_shape(int N, int T, int H, int C, bool aux, bool backward, int* qpt, int* spb, int* grid, int* nthr, size_t* lds,
                            bool quiet, int* direct = nullptr) {
    const int HC = H * C;
#define STAGE_NO(...)                                  \
    do {                                               \
        if (!quiet) otvae_set_error(__VA_ARGS__);      \
        return OTVAE_EUNSUPPORTED;                     \
    } while (0)
    if (N <= 0 || T <= 1 || H <= 0 || C <= 0) STAGE_NO("otvae_attn_stage: bad sizes / T == 1");
    // width 64 (48 KiB of qkv weights per workgroup, 16 tokens to spend them on at the 2x2 maps where it occurs) measured SLOWER than the
    // three launches: 26 us against 6.7 + 5.5 + 4.4 (profiles/r03_attn_stage_ab.txt)
    if (T % 4 != 0 || HC > 32 || (HC & (HC - 1)) != 0) STAGE_NO("otvae_attn_stage: needs T %% 4 == 0 and a power-of-two width <= 32 (T=%d, width=%d)", T, HC);
    switch (C) {
        case 1: case 2: case 3: case 4: case 6: case 8: case 12: case 16: case 32: break;
        default: STAGE_NO("otvae_attn_stage: head width %d not instantiated", C);
    }
    *qpt = pick_qpt(T, C);
    const int tps = T / *qpt;
    if (tps > 256 || (*qpt == 4 && C > 4)) STAGE_NO("otvae_attn_stage: T = %d with head width %d unsupported", T, C);
    // forward: {k, v (, u)} records + q + input / output tile; backward: {k, v} + {q, gout, lse, delta} records + gy tile + x tile
    const int rkv = aux ? ((2 * C + C * C + 3) & ~3) : 2 * C;
    // backward: the gy and x tiles are left out (direct = 1: their rows are read from global memory) when that lets a third workgroup
    // share the CU's 160 KiB
    const int rec_slice = T * (2 * C + ((2 * C + 2 + 3) & ~3));
    const int bw_floats = ((7 * HC * HC + 3) & ~3) + 16 * HC;  // transposed weights + wqkv as it is + reduction scratch
    static const int direct_env = getenv("OTVAE_ATTN_STAGE_DIRECT") ? atoi(getenv("OTVAE_ATTN_STAGE_DIRECT")) : 1;  // A/B switch: 0 = tiles
    if (direct) *direct = 0;
    if (backward && direct && direct_env) {
        const int s3 = imin(256 / tps, (ATTN_STAGE_LDS_FLOATS - bw_floats) / rec_slice) / H * H;
        const size_t third = 160 * 1024 / 3;
        if (s3 >= H && ((size_t)s3 * (rec_slice + 2 * T * C) + bw_floats) * sizeof(float) > third &&
            ((size_t)s3 * rec_slice + bw_floats) * sizeof(float) <= third)
            *direct = 1;
    }
    const int per_slice = backward ? rec_slice + ((direct && *direct) ? 0 : 2 * T * C) : T * (rkv + 2 * C);
    const int wfloats = backward ? bw_floats : 4 * HC * HC;  // (forward: both 1x1 kernels' weights)
    int s = imin(256 / tps, (ATTN_STAGE_LDS_FLOATS - wfloats) / per_slice) / H * H;
    if (s < H) STAGE_NO("otvae_attn_stage: the %d heads of an image do not fit one workgroup (T=%d)", H, T);
    // fill the chip: halve the images per block while the grid is short of one block per CU
    while ((int64_t)N * H / s < 256 && s % (2 * H) == 0 && ((s / 2) * tps) % 64 == 0 && (s / 2 / H) * T >= 16) s /= 2;
    if ((s / H) * T < 16) STAGE_NO("otvae_attn_stage: fewer than 16 tokens per workgroup");
    *spb = s;
    *grid = (int)cdiv((int64_t)N * H, s);
    *nthr = attn_threads(s, T, *qpt);
    *lds = ((size_t)s * per_slice + wfloats) * sizeof(float);
#undef STAGE_NO
    return OTVAE_OK;
}

extern "C" int otvae_attn_stage_plan(int N, int T, int H, int C, int need_aux, int* stat_rows) {
    int qpt, spb, grid, nthr;
    size_t lds;
    int rc = attn_stage_shape(N, T, H, C, need_aux && C <= 2, false, &qpt, &spb, &grid, &nthr, &lds, true);
    if (rc) return rc;
    if (stat_rows) *stat_rows = grid;
    return OTVAE_OK;
}

extern "C" int otvae_attn_stage_bwd_plan(int N, int T, int H, int C, int* bn_rows) {
    int qpt, spb, grid, nthr;
    size_t lds;
    int direct;
    int rc = attn_stage_shape(N, T, H, C, false, true, &qpt, &spb, &grid, &nthr, &lds, true, &direct);
    if (rc) return rc;
    if (bn_rows) *bn_rows = grid;
    return OTVAE_OK;
}

static int attn_stage_fwd_impl(const float* x, const float* scale, const float* shift, const float* wqkv, const float* wproj,
                               const float* residual, int N, int T, int H, int C, float qk_scale, float* qkv, float* out, float* lse,
                               float* aux, float* y, double* stat_partial, const BnFold& fold, void* stream);

extern "C" int otvae_attn_stage_fwd(const float* x, const float* scale, const float* shift, const float* wqkv, const float* wproj,
                                    const float* residual, int N, int T, int H, int C, float qk_scale, float* qkv, float* out, float* lse,
                                    float* aux, float* y, double* stat_partial, void* stream) {
    return attn_stage_fwd_impl(x, scale, shift, wqkv, wproj, residual, N, T, H, C, qk_scale, qkv, out, lse, aux, y, stat_partial, BnFold{},
                               stream);
}

// the same launch with the round-4 extras: the BatchNorm in front of the qkv convolution folded in (fold->slots != NULL; scale / shift
// are then ignored) and / or the output's statistics into slots (stat_slots != NULL, channel stride H * C) instead of partials
extern "C" int otvae_attn_stage_fwd_fold(const float* x, const otvae_bn_fold* fold, const float* scale, const float* shift,
                                         const float* wqkv, const float* wproj, const float* residual, int N, int T, int H, int C,
                                         float qk_scale, float* qkv, float* out, float* lse, float* aux, float* y, double* stat_partial,
                                         void* stat_slots, int stat_nslots, void* stream) {
    OTVAE_REQUIRE(!(stat_partial && stat_slots), "otvae_attn_stage_fwd_fold: statistics go to partials OR to slots");
    OTVAE_REQUIRE_SLOTS("otvae_attn_stage_fwd_fold", stat_slots, stat_nslots);
    BnFold f = {};
    if (fold)
        if (int rc = bn_fold_from_abi("otvae_attn_stage_fwd_fold", *fold, H * C, &f)) return rc;
    return attn_stage_fwd_impl(x, scale, shift, wqkv, wproj, residual, N, T, H, C, qk_scale, qkv, out, lse, aux, y,
                               stat_slots ? bn_tag_slots(stat_slots, stat_nslots) : stat_partial, f, stream);
}

static int attn_stage_fwd_impl(const float* x, const float* scale, const float* shift, const float* wqkv, const float* wproj,
                               const float* residual, int N, int T, int H, int C, float qk_scale, float* qkv, float* out, float* lse,
                               float* aux, float* y, double* stat_partial, const BnFold& fold, void* stream) {
    OTVAE_REQUIRE(x && wqkv && wproj && out && lse && y, "otvae_attn_stage_fwd: NULL tensor");
    OTVAE_REQUIRE((scale == nullptr) == (shift == nullptr), "otvae_attn_stage_fwd: scale/shift must come together");
    OTVAE_REQUIRE(qk_scale > 0.f, "otvae_attn_stage_fwd: scale must be positive");
    if (C > 2) aux = nullptr;
    int qpt, spb, grid, nthr;
    size_t lds;
    int rc = attn_stage_shape(N, T, H, C, aux != nullptr, false, &qpt, &spb, &grid, &nthr, &lds, false);
    if (rc) return rc;
    OTVAE_REQUIRE(attn_aligned16(x) && attn_aligned16(scale) && attn_aligned16(shift), "otvae_attn_stage_fwd: x / scale / shift must be 16-byte aligned");
    OTVAE_REQUIRE(((uintptr_t)qkv & 7) == 0 && ((uintptr_t)y & 7) == 0 && ((uintptr_t)residual & 7) == 0,
                  "otvae_attn_stage_fwd: qkv / y / residual must be 8-byte aligned");
    const AttnStage sg = {x, scale, shift, wqkv, wproj, residual, qkv, y, stat_partial, fold};
    hipStream_t st = (hipStream_t)stream;
#define STAGE_K(CC)                                                                                                        \
    do {                                                                                                                   \
        if (qpt == 4) {                                                                                                    \
            if constexpr (CC <= 2) {                                                                                       \
                if (aux) { rc = stage_lds_ok<&attn_stage_fwd_kernel<CC, 4, true>>(lds); if (rc) return rc; (attn_stage_fwd_kernel<CC, 4, true>)<<<grid, nthr, lds, st>>>(sg, N, T, H, spb, out, lse, aux, qk_scale); } \
                else { rc = stage_lds_ok<&attn_stage_fwd_kernel<CC, 4, false>>(lds); if (rc) return rc; (attn_stage_fwd_kernel<CC, 4, false>)<<<grid, nthr, lds, st>>>(sg, N, T, H, spb, out, lse, nullptr, qk_scale); } \
            } else if constexpr (CC <= 4) { rc = stage_lds_ok<&attn_stage_fwd_kernel<CC, 4, false>>(lds); if (rc) return rc; (attn_stage_fwd_kernel<CC, 4, false>)<<<grid, nthr, lds, st>>>(sg, N, T, H, spb, out, lse, nullptr, qk_scale); } \
        } else {                                                                                                           \
            if constexpr (CC <= 2) {                                                                                       \
                if (aux) { rc = stage_lds_ok<&attn_stage_fwd_kernel<CC, 1, true>>(lds); if (rc) return rc; (attn_stage_fwd_kernel<CC, 1, true>)<<<grid, nthr, lds, st>>>(sg, N, T, H, spb, out, lse, aux, qk_scale); } \
                else { rc = stage_lds_ok<&attn_stage_fwd_kernel<CC, 1, false>>(lds); if (rc) return rc; (attn_stage_fwd_kernel<CC, 1, false>)<<<grid, nthr, lds, st>>>(sg, N, T, H, spb, out, lse, nullptr, qk_scale); } \
            } else { rc = stage_lds_ok<&attn_stage_fwd_kernel<CC, 1, false>>(lds); if (rc) return rc; (attn_stage_fwd_kernel<CC, 1, false>)<<<grid, nthr, lds, st>>>(sg, N, T, H, spb, out, lse, nullptr, qk_scale); } \
        }                                                                                                                  \
    } while (0)
    ATTN_C_SWITCH(C, STAGE_K)
#undef STAGE_K
    OTVAE_CHECK_LAUNCH("otvae_attn_stage_fwd");
    return OTVAE_OK;
}

static int attn_stage_bwd_impl(const float* gy, const float* wproj, const float* wqkv, const float* x, const float* mean,
                               const float* invstd, const float* scale, const float* shift, const float* qkv, const float* out,
                               const float* lse, const float* aux, int N, int T, int H, int C, float qk_scale, float* gqkv, float* gv,
                               double* bn_partial, void* stream);

extern "C" int otvae_attn_stage_bwd(const float* gy, const float* wproj, const float* wqkv, const float* x, const float* mean,
                                    const float* invstd, const float* scale, const float* shift, const float* qkv, const float* out,
                                    const float* lse, const float* aux, int N, int T, int H, int C, float qk_scale, float* gqkv, float* gv,
                                    double* bn_partial, void* stream) {
    return attn_stage_bwd_impl(gy, wproj, wqkv, x, mean, invstd, scale, shift, qkv, out, lse, aux, N, T, H, C, qk_scale, gqkv, gv,
                               bn_partial, stream);
}

// the same launch with the BatchNorm-backward sums (sum gv, sum gv * xhat) into statistic slots (channel stride H * C) instead of partials
extern "C" int otvae_attn_stage_bwd_slots(const float* gy, const float* wproj, const float* wqkv, const float* x, const float* mean,
                                          const float* invstd, const float* scale, const float* shift, const float* qkv,
                                          const float* out, const float* lse, const float* aux, int N, int T, int H, int C,
                                          float qk_scale, float* gqkv, float* gv, void* bn_slots, int bn_nslots, void* stream) {
    OTVAE_REQUIRE(bn_slots, "otvae_attn_stage_bwd_slots: NULL slots");
    OTVAE_REQUIRE_SLOTS("otvae_attn_stage_bwd_slots", bn_slots, bn_nslots);
    return attn_stage_bwd_impl(gy, wproj, wqkv, x, mean, invstd, scale, shift, qkv, out, lse, aux, N, T, H, C, qk_scale, gqkv, gv,
                               bn_tag_slots(bn_slots, bn_nslots), stream);
}

static int attn_stage_bwd_impl(const float* gy, const float* wproj, const float* wqkv, const float* x, const float* mean,
                               const float* invstd, const float* scale, const float* shift, const float* qkv, const float* out,
                               const float* lse, const float* aux, int N, int T, int H, int C, float qk_scale, float* gqkv, float* gv,
                               double* bn_partial, void* stream) {
    OTVAE_REQUIRE(gy && wproj && wqkv && out && lse && gqkv && gv, "otvae_attn_stage_bwd: NULL tensor");
    OTVAE_REQUIRE(qkv || x, "otvae_attn_stage_bwd: without the saved qkv the block input x is needed");
    OTVAE_REQUIRE((scale == nullptr) == (shift == nullptr), "otvae_attn_stage_bwd: scale/shift must come together");
    OTVAE_REQUIRE(attn_aligned16(x) && attn_aligned16(scale) && attn_aligned16(shift), "otvae_attn_stage_bwd: x / scale / shift must be 16-byte aligned");
    OTVAE_REQUIRE((mean == nullptr) == (invstd == nullptr), "otvae_attn_stage_bwd: mean/invstd must come together");
    OTVAE_REQUIRE(!mean || (x && bn_partial), "otvae_attn_stage_bwd: x and bn_partial are needed for the BatchNorm sums");
    OTVAE_REQUIRE(qk_scale > 0.f, "otvae_attn_stage_bwd: scale must be positive");
    if (C > 2) aux = nullptr;
    int qpt, spb, grid, nthr;
    size_t lds;
    int direct = 0;
    int rc = attn_stage_shape(N, T, H, C, false, true, &qpt, &spb, &grid, &nthr, &lds, false, &direct);
    if (rc) return rc;
    OTVAE_REQUIRE(((uintptr_t)gv & 7) == 0 && attn_aligned16(gy), "otvae_attn_stage_bwd: gv must be 8-byte, gy 16-byte aligned");
    const AttnStageBwd sg = {gy, wproj, wqkv, x, mean, invstd, gv, bn_partial, scale, shift, direct};
    hipStream_t st = (hipStream_t)stream;
#define STAGE_BK(K_)                                                                                  \
    {                                                                                                 \
        rc = stage_lds_ok<&K_>(lds);                                                                  \
        if (rc) return rc;                                                                            \
        K_<<<grid, nthr, lds, st>>>(sg, qkv, out, lse, aux, N, T, H, spb, gqkv, qk_scale);            \
    }
#define STAGE_B(CC)                                                                                   \
    do {                                                                                              \
        if (qpt == 4) {                                                                               \
            if constexpr (CC <= 2) {                                                                  \
                if (aux) STAGE_BK((attn_stage_bwd_kernel<CC, 4, true>))                               \
                else STAGE_BK((attn_stage_bwd_kernel<CC, 4, false>))                                  \
            } else if constexpr (CC <= 4) STAGE_BK((attn_stage_bwd_kernel<CC, 4, false>))             \
        } else {                                                                                      \
            if constexpr (CC <= 2) {                                                                  \
                if (aux) STAGE_BK((attn_stage_bwd_kernel<CC, 1, true>))                               \
                else STAGE_BK((attn_stage_bwd_kernel<CC, 1, false>))                                  \
            } else STAGE_BK((attn_stage_bwd_kernel<CC, 1, false>))                                    \
        }                                                                                             \
    } while (0)
    ATTN_C_SWITCH(C, STAGE_B)
#undef STAGE_B
#undef STAGE_BK
    OTVAE_CHECK_LAUNCH("otvae_attn_stage_bwd");
    return OTVAE_OK;
}

extern "C" int otvae_attn_bwd_scaled(const float* qkv, const float* out, const float* lse, const float* gout, const float* aux, int N,
                                     int T, int H, int C, float scale, float* gqkv, void* stream);
extern "C" int otvae_attn_bwd(const float* qkv, const float* out, const float* lse, const float* gout, const float* aux, int N,
                              int T, int H, int C, float* gqkv, void* stream) {
    OTVAE_REQUIRE(C > 0, "otvae_attn_bwd: bad sizes");
    return otvae_attn_bwd_scaled(qkv, out, lse, gout, aux, N, T, H, C, 1.f / (float)C, gqkv, stream);
}

extern "C" int otvae_attn_bwd_scaled(const float* qkv, const float* out, const float* lse, const float* gout, const float* aux, int N,
                                     int T, int H, int C, float scale, float* gqkv, void* stream) {
    OTVAE_REQUIRE(qkv && out && lse && gout && gqkv, "otvae_attn_bwd: NULL tensor");
    OTVAE_REQUIRE(scale > 0.f, "otvae_attn_bwd: scale must be positive");
    if (T == 1 && N > 0 && H > 0 && C > 0) {
        attn_t1_bwd_kernel<<<cdiv((int64_t)N * 3 * H * C, 256), 256, 0, (hipStream_t)stream>>>(gout, N, H * C, gqkv);
        OTVAE_CHECK_LAUNCH("otvae_attn_bwd(T=1)");
        return OTVAE_OK;
    }
    int qpt, spb;
    const int rqg = (2 * C + 2 + 3) & ~3;
    int rc = attn_check("otvae_attn_bwd", N, T, H, C, 2 * C + rqg, &qpt, &spb);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int grid = cdiv((int64_t)N * H, spb), nthr = attn_threads(spb, T, qpt);
    const size_t lds = (size_t)spb * T * (2 * C + rqg) * sizeof(float);
#define BWD_K(CC)                                                                                   \
    do {                                                                                            \
        if (qpt == 4) {                                                                             \
            if constexpr (CC <= 2) {                                                                \
                if (aux) attn_bwd_kernel<CC, 4, true><<<grid, nthr, lds, st>>>(qkv, out, lse, gout, aux, N, T, H, spb, gqkv, scale); \
                else attn_bwd_kernel<CC, 4, false><<<grid, nthr, lds, st>>>(qkv, out, lse, gout, nullptr, N, T, H, spb, gqkv, scale); \
            } else if constexpr (CC <= 4) attn_bwd_kernel<CC, 4, false><<<grid, nthr, lds, st>>>(qkv, out, lse, gout, nullptr, N, T, H, spb, gqkv, scale); \
            else { otvae_set_error("otvae_attn_bwd: T >= 256 with head width %d > 4 unsupported", CC); return OTVAE_EUNSUPPORTED; } \
        } else {                                                                                    \
            if constexpr (CC <= 2) {                                                                \
                if (aux) attn_bwd_kernel<CC, 1, true><<<grid, nthr, lds, st>>>(qkv, out, lse, gout, aux, N, T, H, spb, gqkv, scale); \
                else attn_bwd_kernel<CC, 1, false><<<grid, nthr, lds, st>>>(qkv, out, lse, gout, nullptr, N, T, H, spb, gqkv, scale); \
            } else attn_bwd_kernel<CC, 1, false><<<grid, nthr, lds, st>>>(qkv, out, lse, gout, nullptr, N, T, H, spb, gqkv, scale); \
        }                                                                                           \
    } while (0)
    ATTN_C_SWITCH(C, BWD_K)
#undef BWD_K
    OTVAE_CHECK_LAUNCH("otvae_attn_bwd");
    return OTVAE_OK;
}
