// BatchNorm2d (training mode) statistics, finalisation and backward for NHWC activations.
// Replaces F.batch_norm(training=True) + its autograd backward as used by ConvLayer._normalization
// (reference networks/cnn.py:122,184).  Sums are accumulated in fp64 and combined in a fixed order
// (per-thread -> LDS tree -> per-block partial -> serial over <= 256 partials), so results are run-to-run identical.
#include "common.h"

#define BN_MAX_PARTS 256

extern "C" int otvae_bn_stats_nparts(int64_t M, int C) {
    if (M <= 0 || C <= 0) return 0;
    const int rpb = C >= 256 ? 1 : 256 / C;  // rows handled per block iteration
    int64_t want = (M + (int64_t)rpb * 16 - 1) / ((int64_t)rpb * 16);
    if (want < 1) want = 1;
    return (int)(want > BN_MAX_PARTS ? BN_MAX_PARTS : want);
}

// thread (rr, c): rows rr, rr+RPB*P ... of channel c (C <= 256), or loops channels (C > 256)
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, int64_t M, int C, double* __restrict__ partial,
                                                       long long* __restrict__ slots = nullptr, int ld = 0, unsigned smask = 0) {
    __shared__ double sh[2][256];
    const int P = gridDim.x;
    if (C <= 256) {
        const int rpb = 256 / C;
        const int c = threadIdx.x % C, rr = threadIdx.x / C;
        double s = 0.0, q = 0.0;
        if (rr < rpb) {
            for (int64_t m = (int64_t)blockIdx.x * rpb + rr; m < M; m += (int64_t)P * rpb) {
                const float v = x[m * C + c];
                s += (double)v;
                q += (double)v * (double)v;
            }
        }
        sh[0][threadIdx.x] = s;
        sh[1][threadIdx.x] = q;
        __syncthreads();
        if ((int)threadIdx.x < C) {
            double ts = 0.0, tq = 0.0;
            for (int r = 0; r < rpb; ++r) {
                ts += sh[0][r * C + threadIdx.x];
                tq += sh[1][r * C + threadIdx.x];
            }
            if (slots) {
                bn_slot_add(slots, ld, smask + 1u, blockIdx.x & smask, 0, threadIdx.x, ts);
                bn_slot_add(slots, ld, smask + 1u, blockIdx.x & smask, 1, threadIdx.x, tq);
            } else {
                partial[((size_t)0 * C + threadIdx.x) * P + blockIdx.x] = ts;
                partial[((size_t)1 * C + threadIdx.x) * P + blockIdx.x] = tq;
            }
        }
    } else {
        for (int c = threadIdx.x; c < C; c += 256) {
            double s = 0.0, q = 0.0;
            for (int64_t m = blockIdx.x; m < M; m += P) {
                const float v = x[m * C + c];
                s += (double)v;
                q += (double)v * (double)v;
            }
            if (slots) {
                bn_slot_add(slots, ld, smask + 1u, blockIdx.x & smask, 0, c, s);
                bn_slot_add(slots, ld, smask + 1u, blockIdx.x & smask, 1, c, q);
            } else {
                partial[((size_t)0 * C + c) * P + blockIdx.x] = s;
                partial[((size_t)1 * C + c) * P + blockIdx.x] = q;
            }
        }
    }
}

extern "C" int otvae_bn_stats(const float* x, int64_t M, int C, double* partial, void* stream) {
    OTVAE_REQUIRE(x && partial && M > 0 && C > 0, "otvae_bn_stats: bad argument");
    const int P = otvae_bn_stats_nparts(M, C);
    bn_stats_kernel<<<P, 256, 0, (hipStream_t)stream>>>(x, M, C, partial);
    OTVAE_CHECK_LAUNCH("otvae_bn_stats");
    return OTVAE_OK;
}

// ---- statistic slots (common.h): the sums land in BN_SLOTS accumulators instead of P partials; no finalize launch is needed when the
// consumer folds them itself (BnFold), otvae_bn_finalize_slots is the stand-alone form for consumers that do not
extern "C" int64_t otvae_bn_slots_words(int ld, int nslots) { return (ld > 0 && bn_slots_ok(nslots)) ? (int64_t)bn_slot_words(ld, nslots) : -1; }

extern "C" int otvae_bn_stats_slots(const float* x, int64_t M, int C, void* slots, int ld, int nslots, void* stream) {
    OTVAE_REQUIRE(x && slots && M > 0 && C > 0 && ld >= C, "otvae_bn_stats_slots: bad argument");
    OTVAE_REQUIRE_SLOTS("otvae_bn_stats_slots", slots, nslots);
    const int P = otvae_bn_stats_nparts(M, C);
    bn_stats_kernel<<<P, 256, 0, (hipStream_t)stream>>>(x, M, C, nullptr, (long long*)slots, ld, (unsigned)nslots - 1u);
    OTVAE_CHECK_LAUNCH("otvae_bn_stats_slots");
    return OTVAE_OK;
}

__global__ __launch_bounds__(256) void bn_finalize_slots_kernel(int n, BnFold f0, BnFold f1, int C) {
    __shared__ float sink[2][1024];
    for (int b = 0; b < n; ++b) bn_fold_prologue(b == 0 ? f0 : f1, C, sink[0], sink[1], true);
}

extern "C" int otvae_bn_finalize_slots(int n_bn, const otvae_bn_fold* folds, int C, void* stream) {
    OTVAE_REQUIRE(n_bn >= 1 && n_bn <= 2 && folds && C > 0 && C <= 1024, "otvae_bn_finalize_slots: 1 or 2 branches, C <= 1024");
    BnFold f[2] = {};
    for (int b = 0; b < n_bn; ++b)
    {
        OTVAE_REQUIRE(folds[b].slots, "otvae_bn_finalize_slots: branch %d has no slots", b);
        if (int rc = bn_fold_from_abi("otvae_bn_finalize_slots", folds[b], C, &f[b])) return rc;
    }
    bn_finalize_slots_kernel<<<1, 256, 0, (hipStream_t)stream>>>(n_bn, f[0], f[1], C);
    OTVAE_CHECK_LAUNCH("otvae_bn_finalize_slots");
    return OTVAE_OK;
}

struct BnFin {
    const float* gamma[2];
    const float* beta[2];
    float* rmean[2];
    float* rvar[2];
    int64_t* nbt[2];
    float* scale[2];
    float* shift[2];
};

// WPC waves per channel (1 when there are few partials, 4 = one block per channel otherwise): lanes stride over the
// partials, fixed-order shuffle tree, then (WPC == 4) the 4 waves through LDS in wave order
template <int WPC>
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ partial, int P, int ld, int64_t M, int C,
                                                          float eps, float momentum, float* __restrict__ mean, float* __restrict__ invstd,
                                                          int n_bn, BnFin f) {
    __shared__ double red[4][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = WPC == 4 ? (int)blockIdx.x : (int)blockIdx.x * 4 + wave;
    const int sub = WPC == 4 ? wave : 0;
    const bool writer = c < C && lane == 0 && sub == 0;
    // what the writing lane needs besides the sums is requested first: these loads fly together with the partials' instead of
    // forming a second, dependent round trip behind the reduction (the kernel is nothing but memory latency)
    float gam[2] = {0.f, 0.f}, bet[2] = {0.f, 0.f}, rm[2] = {0.f, 0.f}, rv[2] = {0.f, 0.f};
    if (writer)
        for (int b = 0; b < n_bn; ++b) {
            gam[b] = f.gamma[b][c];
            bet[b] = f.beta[b][c];
            if (f.rmean[b]) rm[b] = f.rmean[b][c];
            if (f.rvar[b]) rv[b] = f.rvar[b][c];
        }
    double s = 0.0, q = 0.0;
    if (c < C) {
        const double* ps = partial + ((size_t)0 * ld + c) * P;  // [2][ld][P]: lanes read consecutive p
        const double* pq = partial + ((size_t)1 * ld + c) * P;
        for (int p = sub * 64 + lane; p < P; p += 64 * WPC) {
            s += ps[p];
            q += pq[p];
        }
        s = wave_sum(s);
        q = wave_sum(q);
    }
    if (WPC == 4) {
        if (lane == 0) {
            red[wave][0] = s;
            red[wave][1] = q;
        }
        __syncthreads();
        s = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
        q = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
    }
    if (writer) {
        const double mu = s / (double)M;
        double var = q / (double)M - mu * mu;
        if (var < 0.0) var = 0.0;
        const float fmu = (float)mu;
        const float fis = (float)(1.0 / sqrt(var + (double)eps));
        mean[c] = fmu;
        invstd[c] = fis;
        const double unbiased = M > 1 ? var * ((double)M / (double)(M - 1)) : var;
        for (int b = 0; b < n_bn; ++b) {
            const float sc = gam[b] * fis;
            f.scale[b][c] = sc;
            f.shift[b][c] = fmaf(-fmu, sc, bet[b]);
            // step guard: a non-finite batch statistic (a NaN that reached this layer's input) must not enter the running
            // buffers, where it would stay for good -- the step itself is skipped by the guarded Adam kernel
            const bool fin = isfinite(fmu) && isfinite((float)unbiased);
            if (f.rmean[b] && fin) f.rmean[b][c] = (1.f - momentum) * rm[b] + momentum * fmu;
            if (f.rvar[b] && fin) f.rvar[b][c] = (1.f - momentum) * rv[b] + momentum * (float)unbiased;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int b = 0; b < n_bn; ++b)
            if (f.nbt[b]) *f.nbt[b] += 1;
}

extern "C" int otvae_bn_finalize(const double* partial, int P, int ld, int64_t M, int C, float eps, float momentum, float* mean,
                                 float* invstd, int n_bn, const float* const* gamma, const float* const* beta,
                                 float* const* running_mean, float* const* running_var, int64_t* const* num_batches_tracked,
                                 float* const* scale, float* const* shift, void* stream) {
    OTVAE_REQUIRE(partial && mean && invstd && P > 0 && M > 0 && C > 0 && ld >= C, "otvae_bn_finalize: bad argument");
    OTVAE_REQUIRE(n_bn >= 0 && n_bn <= 2, "otvae_bn_finalize: n_bn must be 0..2");
    BnFin f = {};
    for (int b = 0; b < n_bn; ++b) {
        OTVAE_REQUIRE(gamma[b] && beta[b] && scale[b] && shift[b], "otvae_bn_finalize: NULL gamma/beta/scale/shift");
        f.gamma[b] = gamma[b];
        f.beta[b] = beta[b];
        f.rmean[b] = running_mean ? running_mean[b] : nullptr;
        f.rvar[b] = running_var ? running_var[b] : nullptr;
        f.nbt[b] = num_batches_tracked ? num_batches_tracked[b] : nullptr;
        f.scale[b] = scale[b];
        f.shift[b] = shift[b];
    }
    if (P >= 256)
        bn_finalize_kernel<4><<<C, 256, 0, (hipStream_t)stream>>>(partial, P, ld, M, C, eps, momentum, mean, invstd, n_bn, f);
    else
        bn_finalize_kernel<1><<<cdiv(C, 4), 256, 0, (hipStream_t)stream>>>(partial, P, ld, M, C, eps, momentum, mean, invstd, n_bn, f);
    OTVAE_CHECK_LAUNCH("otvae_bn_finalize");
    return OTVAE_OK;
}

// ---- backward ------------------------------------------------------------------------------------------------
// For y = xhat*gamma + beta, xhat = (x-mean)*invstd, with g = dL/dy:
//   dgamma = sum g*xhat, dbeta = sum g, dx = gamma*invstd*(g - dbeta/M - xhat*dgamma/M)
// With up to two branches b normalising the same x:
//   dx = sum_b k_b*g_b - A*x - B,  k_b = gamma_b*invstd, A = invstd/M * sum_b k_b*dgamma_b,
//   B = sum_b k_b*(dbeta_b/M) - A*mean
struct BnBwdFin {
    const double* partial[2];
    int P[2];
    const float* gamma[2];
    float* dgamma[2];
    float* dbeta[2];
};

template <int WPC>
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(int nb, BnBwdFin f, int CsPad, int64_t M, int C,
                                                              const float* __restrict__ mean, const float* __restrict__ invstd,
                                                              float* __restrict__ coef) {
    __shared__ double red[2][4][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = WPC == 4 ? (int)blockIdx.x : (int)blockIdx.x * 4 + wave;
    const int sub = WPC == 4 ? wave : 0;
    const bool in = c < C;
    const bool writer = in && lane == 0 && sub == 0;
    // the writing lane's per-channel operands are requested before the reduction (one memory round trip instead of two)
    double is = 0.0, mu = 0.0, gam[2] = {0.0, 0.0};
    if (writer) {
        is = (double)invstd[c];
        mu = (double)mean[c];
        for (int b = 0; b < nb; ++b) gam[b] = (double)f.gamma[b][c];
    }
    double s1[2] = {0.0, 0.0}, s2[2] = {0.0, 0.0};
    for (int b = 0; b < nb; ++b) {
        if (in) {
            const double* p1 = f.partial[b] + ((size_t)0 * CsPad + c) * f.P[b];
            const double* p2 = f.partial[b] + ((size_t)1 * CsPad + c) * f.P[b];
            double a = 0.0, q = 0.0;
            for (int p = sub * 64 + lane; p < f.P[b]; p += 64 * WPC) {
                a += p1[p];
                q += p2[p];
            }
            s1[b] = wave_sum(a);
            s2[b] = wave_sum(q);
        }
    }
    if (WPC == 4) {
        if (lane == 0)
            for (int b = 0; b < nb; ++b) {
                red[b][wave][0] = s1[b];
                red[b][wave][1] = s2[b];
            }
        __syncthreads();
        for (int b = 0; b < nb; ++b) {
            s1[b] = (red[b][0][0] + red[b][1][0]) + (red[b][2][0] + red[b][3][0]);
            s2[b] = (red[b][0][1] + red[b][1][1]) + (red[b][2][1] + red[b][3][1]);
        }
    }
    if (!writer) return;
    double A = 0.0, B = 0.0;
    for (int b = 0; b < nb; ++b) {
        const double k = gam[b] * is;
        if (f.dbeta[b]) f.dbeta[b][c] = (float)s1[b];
        if (f.dgamma[b]) f.dgamma[b][c] = (float)s2[b];
        coef[(size_t)(2 + b) * C + c] = (float)k;
        A += k * s2[b];
        B += k * s1[b];
    }
    A = A * is / (double)M;
    B = B / (double)M - A * mu;
    coef[c] = (float)A;
    coef[(size_t)C + c] = (float)B;
}

extern "C" int otvae_bn_bwd_finalize(int nb, const double* const* bn_partial, const int* P, int CsPad, int64_t M, int C,
                                     const float* mean, const float* invstd, const float* const* gamma, float* const* dgamma,
                                     float* const* dbeta, float* coef, void* stream) {
    OTVAE_REQUIRE(nb >= 1 && nb <= 2, "otvae_bn_bwd_finalize: nb must be 1 or 2");
    OTVAE_REQUIRE(bn_partial && P && mean && invstd && gamma && coef && M > 0 && C > 0 && CsPad >= C,
                  "otvae_bn_bwd_finalize: bad argument");
    BnBwdFin f = {};
    for (int b = 0; b < nb; ++b) {
        OTVAE_REQUIRE(bn_partial[b] && gamma[b] && P[b] > 0, "otvae_bn_bwd_finalize: NULL branch %d", b);
        f.partial[b] = bn_partial[b];
        f.P[b] = P[b];
        f.gamma[b] = gamma[b];
        f.dgamma[b] = dgamma ? dgamma[b] : nullptr;
        f.dbeta[b] = dbeta ? dbeta[b] : nullptr;
    }
    if (imax(P[0], nb > 1 ? P[1] : 0) >= 256)
        bn_bwd_finalize_kernel<4><<<C, 256, 0, (hipStream_t)stream>>>(nb, f, CsPad, M, C, mean, invstd, coef);
    else
        bn_bwd_finalize_kernel<1><<<cdiv(C, 4), 256, 0, (hipStream_t)stream>>>(nb, f, CsPad, M, C, mean, invstd, coef);
    OTVAE_CHECK_LAUNCH("otvae_bn_bwd_finalize");
    return OTVAE_OK;
}

template <int NB, bool VEC>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ g0, const float* __restrict__ g1,
                                                           const float* __restrict__ x, const float* __restrict__ coef,
                                                           int64_t total, int C, float* __restrict__ dx) {
    if constexpr (VEC) {
        const int64_t n4 = total >> 2;
        for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
            const int c = (int)((i << 2) % C);
            const float4 xv = reinterpret_cast<const float4*>(x)[i];
            const float4 a = *reinterpret_cast<const float4*>(coef + c);
            const float4 b = *reinterpret_cast<const float4*>(coef + C + c);
            const float4 k0 = *reinterpret_cast<const float4*>(coef + 2 * (size_t)C + c);
            const float4 gv0 = reinterpret_cast<const float4*>(g0)[i];
            float4 r;
            r.x = fmaf(k0.x, gv0.x, fmaf(-a.x, xv.x, -b.x));
            r.y = fmaf(k0.y, gv0.y, fmaf(-a.y, xv.y, -b.y));
            r.z = fmaf(k0.z, gv0.z, fmaf(-a.z, xv.z, -b.z));
            r.w = fmaf(k0.w, gv0.w, fmaf(-a.w, xv.w, -b.w));
            if constexpr (NB == 2) {
                const float4 k1 = *reinterpret_cast<const float4*>(coef + 3 * (size_t)C + c);
                const float4 gv1 = reinterpret_cast<const float4*>(g1)[i];
                r.x = fmaf(k1.x, gv1.x, r.x);
                r.y = fmaf(k1.y, gv1.y, r.y);
                r.z = fmaf(k1.z, gv1.z, r.z);
                r.w = fmaf(k1.w, gv1.w, r.w);
            }
            reinterpret_cast<float4*>(dx)[i] = r;
        }
    } else {
        for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
            const int c = (int)(i % C);
            float r = fmaf(coef[2 * (size_t)C + c], g0[i], fmaf(-coef[c], x[i], -coef[(size_t)C + c]));
            if constexpr (NB == 2) r = fmaf(coef[3 * (size_t)C + c], g1[i], r);
            dx[i] = r;
        }
    }
}

extern "C" int otvae_bn_bwd_apply(int nb, const float* const* gv, const float* x, const float* coef, int64_t M, int C,
                                  float* dx, void* stream) {
    OTVAE_REQUIRE(nb >= 1 && nb <= 2 && gv && gv[0] && x && coef && dx && M > 0 && C > 0, "otvae_bn_bwd_apply: bad argument");
    OTVAE_REQUIRE(nb == 1 || gv[1], "otvae_bn_bwd_apply: second branch NULL");
    const int64_t total = M * C;
    const bool vec = (C % 4 == 0);
    const int grid = imin(cdiv(vec ? total / 4 : total, 256), 4096);
    hipStream_t st = (hipStream_t)stream;
    const float* g1 = nb == 2 ? gv[1] : nullptr;
    if (nb == 1) {
        if (vec) bn_bwd_apply_kernel<1, true><<<grid, 256, 0, st>>>(gv[0], g1, x, coef, total, C, dx);
        else bn_bwd_apply_kernel<1, false><<<grid, 256, 0, st>>>(gv[0], g1, x, coef, total, C, dx);
    } else {
        if (vec) bn_bwd_apply_kernel<2, true><<<grid, 256, 0, st>>>(gv[0], g1, x, coef, total, C, dx);
        else bn_bwd_apply_kernel<2, false><<<grid, 256, 0, st>>>(gv[0], g1, x, coef, total, C, dx);
    }
    OTVAE_CHECK_LAUNCH("otvae_bn_bwd_apply");
    return OTVAE_OK;
}

// ---- the backward pair with the sums in statistic slots: the finalize arithmetic (bn_bwd_finalize_kernel) runs in the apply launch's
// prologue -- every block for itself from the S slots, into LDS -- and the first block writes dgamma / dbeta.  dx == NULL: one block,
// parameter gradients only (the first layer of a network: nobody needs dL/dx).
struct BnBwdFold {
    const long long* slots[2];
    int nslots[2];
    int ld;
    int training;
    long long count;
    const float* mean;
    const float* invstd;
    const float* gamma[2];
    float* dgamma[2];
    float* dbeta[2];
};

template <int NB, bool VEC>
__global__ __launch_bounds__(256) void bn_bwd_apply_slots_kernel(BnBwdFold f, const float* __restrict__ g0, const float* __restrict__ g1,
                                                                 const float* __restrict__ x, int64_t total, int C, int Cpad,
                                                                 float* __restrict__ dx) {
    // dynamic LDS, sized by the layer (this kernel runs beside the weight-gradient stream's LDS-hungry kernels: a fixed table for 1024
    // channels per block kept those off the CUs and cost the step more than the fused launch saved):
    //   acc [NB][4 C] 64-bit limb totals (narrow layers with many slots only) | tab [2 + NB][Cpad] floats: A, B, k_0 (, k_1)
    extern __shared__ __align__(16) unsigned char bw_smem[];
    bool spread[NB];
    int wn[NB], wtot = 0;
#pragma unroll
    for (int b = 0; b < NB; ++b) {   // block-uniform
        spread[b] = C <= BN_FOLD_WIDE && f.nslots[b] > 4;
        wn[b] = spread[b] ? 2 * C * f.nslots[b] : 0;
        wtot += wn[b];
    }
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(bw_smem);
    float* tab = reinterpret_cast<float*>(bw_smem + (wtot ? (size_t)NB * 4 * C * sizeof(unsigned long long) : 0));
    const bool first = blockIdx.x == 0;
    if (wtot) {
        // the slot reads of BOTH branches spread over the block: 16-byte loads, a handful per thread, all in flight together; summed with
        // LDS integer atomics (associative: same bits in any order)
        for (int i = threadIdx.x; i < NB * 4 * C; i += 256) acc[i] = 0ULL;
        __syncthreads();
        for (int w0 = threadIdx.x; w0 < wtot; w0 += 4 * 256) {
            bn_ll2 v[4];
            int at[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int w = w0 + u * 256;
                at[u] = -1;
                if (w < wtot) {
                    const int b = (NB == 2 && w >= wn[0]) ? 1 : 0;
                    if (b) w -= wn[0];
                    const int c = w % C, r = w / C;   // r = s * 2 + stat
                    v[u] = *reinterpret_cast<const bn_ll2*>(f.slots[b] + (((size_t)r * f.ld + c) << 1));
                    at[u] = b * 4 * C + ((r & 1) * C + c) * 2;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (at[u] >= 0) {
                    atomicAdd(&acc[at[u]], (unsigned long long)v[u].x);
                    atomicAdd(&acc[at[u] + 1], (unsigned long long)v[u].y);
                }
        }
        __syncthreads();
    }
    for (int c = threadIdx.x; c < C; c += 256) {
        const double is = (double)f.invstd[c], mu = (double)f.mean[c];
        double A = 0.0, B = 0.0;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            double s1, s2;
            if (spread[b]) bn_block_totals_read(acc + b * 4 * C, f.slots[b], f.ld, f.nslots[b], C, c, s1, s2);
            else bn_slot_totals(f.slots[b], f.ld, f.nslots[b], c, s1, s2);
            const double k = (double)f.gamma[b][c] * is;
            if (first) {
                if (f.dbeta[b]) f.dbeta[b][c] = (float)s1;
                if (f.dgamma[b]) f.dgamma[b][c] = (float)s2;
            }
            tab[(2 + b) * Cpad + c] = (float)k;
            A += k * s2;
            B += k * s1;
        }
        A = A * is / (double)f.count;
        B = B / (double)f.count - A * mu;
        if (!f.training) A = B = 0.0;   // eval mode: BatchNorm is a fixed affine, no batch-statistics terms
        tab[c] = (float)A;
        tab[Cpad + c] = (float)B;
    }
    __syncthreads();
    if (!dx) return;
    if constexpr (VEC) {
        const int64_t n4 = total >> 2;
        for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
            const int c = (int)((i << 2) % C);
            const float4 xv = reinterpret_cast<const float4*>(x)[i];
            const float4 a = *reinterpret_cast<const float4*>(&tab[c]);
            const float4 b = *reinterpret_cast<const float4*>(&tab[Cpad + c]);
            const float4 k0 = *reinterpret_cast<const float4*>(&tab[2 * Cpad + c]);
            const float4 gv0 = reinterpret_cast<const float4*>(g0)[i];
            float4 r;
            r.x = fmaf(k0.x, gv0.x, fmaf(-a.x, xv.x, -b.x));
            r.y = fmaf(k0.y, gv0.y, fmaf(-a.y, xv.y, -b.y));
            r.z = fmaf(k0.z, gv0.z, fmaf(-a.z, xv.z, -b.z));
            r.w = fmaf(k0.w, gv0.w, fmaf(-a.w, xv.w, -b.w));
            if constexpr (NB == 2) {
                const float4 k1 = *reinterpret_cast<const float4*>(&tab[3 * Cpad + c]);
                const float4 gv1 = reinterpret_cast<const float4*>(g1)[i];
                r.x = fmaf(k1.x, gv1.x, r.x);
                r.y = fmaf(k1.y, gv1.y, r.y);
                r.z = fmaf(k1.z, gv1.z, r.z);
                r.w = fmaf(k1.w, gv1.w, r.w);
            }
            reinterpret_cast<float4*>(dx)[i] = r;
        }
    } else {
        for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
            const int c = (int)(i % C);
            float r = fmaf(tab[2 * Cpad + c], g0[i], fmaf(-tab[c], x[i], -tab[Cpad + c]));
            if constexpr (NB == 2) r = fmaf(tab[3 * Cpad + c], g1[i], r);
            dx[i] = r;
        }
    }
}

extern "C" int otvae_bn_bwd_apply_slots(int nb, const float* const* gv, const float* x, const void* const* slots, const int* nslots,
                                        int ld, int64_t M, int C, const float* mean, const float* invstd, const float* const* gamma,
                                        float* const* dgamma, float* const* dbeta, int training, float* dx, void* stream) {
    OTVAE_REQUIRE(nb >= 1 && nb <= 2 && slots && nslots && mean && invstd && gamma && M > 0 && C > 0 && C <= BN_TAB && ld >= C,
                  "otvae_bn_bwd_apply_slots: bad argument (1 or 2 branches, C <= %d)", BN_TAB);
    OTVAE_REQUIRE(!dx || (gv && gv[0] && x && (nb == 1 || gv[1])), "otvae_bn_bwd_apply_slots: gv / x are needed for dx");
    BnBwdFold f = {};
    for (int b = 0; b < nb; ++b) {
        OTVAE_REQUIRE(slots[b] && gamma[b], "otvae_bn_bwd_apply_slots: NULL branch %d", b);
        OTVAE_REQUIRE_SLOTS("otvae_bn_bwd_apply_slots", slots[b], nslots[b]);
        f.slots[b] = (const long long*)slots[b];
        f.nslots[b] = nslots[b];
        f.gamma[b] = gamma[b];
        f.dgamma[b] = dgamma ? dgamma[b] : nullptr;
        f.dbeta[b] = dbeta ? dbeta[b] : nullptr;
    }
    f.ld = ld, f.training = training, f.count = M, f.mean = mean, f.invstd = invstd;
    const int64_t total = M * C;
    const bool vec = (C % 4 == 0);
    // one resident generation of blocks: every block pays the prologue (S x 4 words per channel and branch) before it streams
    const int grid = dx ? imin(cdiv(vec ? total / 4 : total, 256), 2048) : 1;
    hipStream_t st = (hipStream_t)stream;
    const float* g0 = dx ? gv[0] : nullptr;
    const float* g1 = (dx && nb == 2) ? gv[1] : nullptr;
    const int Cpad = (C + 3) & ~3;
    bool any_spread = false;
    for (int b = 0; b < nb; ++b) any_spread = any_spread || (C <= BN_FOLD_WIDE && nslots[b] > 4);
    const size_t lds = (any_spread ? (size_t)nb * 4 * C * sizeof(unsigned long long) : 0) + (size_t)(2 + nb) * Cpad * sizeof(float);
    if (nb == 1) {
        if (vec) bn_bwd_apply_slots_kernel<1, true><<<grid, 256, lds, st>>>(f, g0, g1, x, total, C, Cpad, dx);
        else bn_bwd_apply_slots_kernel<1, false><<<grid, 256, lds, st>>>(f, g0, g1, x, total, C, Cpad, dx);
    } else {
        if (vec) bn_bwd_apply_slots_kernel<2, true><<<grid, 256, lds, st>>>(f, g0, g1, x, total, C, Cpad, dx);
        else bn_bwd_apply_slots_kernel<2, false><<<grid, 256, lds, st>>>(f, g0, g1, x, total, C, Cpad, dx);
    }
    OTVAE_CHECK_LAUNCH("otvae_bn_bwd_apply_slots");
    return OTVAE_OK;
}
