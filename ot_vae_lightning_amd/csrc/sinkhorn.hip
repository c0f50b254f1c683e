// Log-domain Sinkhorn (reference ot/w2_utils.py:276-319) and the helpers the minibatch-OT prior composes it with.
//
//   u = v = 0; Cr = -C/reg; loop: v_j = log b_j - LSE_i(Cr_ij + u_i); u_i = log a_i - LSE_j(Cr_ij + v_j);
//   stop when min over the batch of (|du|_1 + |dv|_1) < threshold;  pi = exp(u_i + v_j + Cr_ij)
//
// Each half-iteration is a row-wise log-sum-exp, and every one depends on the whole result of the previous one, so a
// half-iteration is one launch: one wave per row, lanes strided along the contiguous dimension, (max, sum) combined
// with wave shuffles.  The column pass runs on a transposed copy of Cr made once, so both passes read contiguous
// rows; for the benchmark size (1024x1024 fp32) both copies stay L2/MALL resident across the 100 passes.
// The reference's per-iteration host sync (.item()) becomes a device flag tested at the top of every kernel.
#include "common.h"

template <typename T>
struct MathT;
template <>
struct MathT<float> {
    static __device__ __forceinline__ float exp(float x) { return __expf(x); }
    static __device__ __forceinline__ float log(float x) { return __logf(x); }
    static __device__ __forceinline__ float ninf() { return -INFINITY; }
};
template <>
struct MathT<double> {
    static __device__ __forceinline__ double exp(double x) { return ::exp(x); }
    static __device__ __forceinline__ double log(double x) { return ::log(x); }
    static __device__ __forceinline__ double ninf() { return -(double)INFINITY; }
};

struct SkCtl {
    int done;        // set by sk_check when the batch-min difference drops below the threshold
    int iters;       // iterations actually performed
};

// Cr = -C/reg (row major, into `cr`) and its transpose (into `crt`), 32x32 tiles through LDS
template <typename T>
__global__ __launch_bounds__(256) void sk_init_mat(const T* __restrict__ Cm, int N, int M, T neg_inv_reg, T* __restrict__ cr,
                                                   T* __restrict__ crt) {
    __shared__ T tile[32][33];
    const size_t boff = (size_t)blockIdx.z * N * M;
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int i = i0 + r, j = j0 + tx;
        if (i < N && j < M) {
            const T v = Cm[boff + (size_t)i * M + j] * neg_inv_reg;
            cr[boff + (size_t)i * M + j] = v;
            tile[r][tx] = v;
        }
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int j = j0 + r, i = i0 + tx;
        if (i < N && j < M) crt[boff + (size_t)j * N + i] = tile[tx][r];
    }
}

template <typename T>
__global__ void sk_init_vec(const T* __restrict__ a, const T* __restrict__ b, int nb, int N, int M, T* __restrict__ loga,
                            T* __restrict__ logb, T* __restrict__ u, T* __restrict__ v, SkCtl* ctl, int preset_iters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nb * N) {
        loga[i] = MathT<T>::log(a[i] + (T)1e-8);
        u[i] = (T)0;
    }
    if (i < nb * M) {
        logb[i] = MathT<T>::log(b[i] + (T)1e-8);
        v[i] = (T)0;
    }
    if (i == 0) {
        ctl->done = 0;
        ctl->iters = preset_iters;
    }
}

// out[r] = logm[r] - LSE_l(mat[r][l] + add[l]);  absd[r] = |out_new - out_old|.  One wave per row.
template <typename T>
__global__ __launch_bounds__(256) void sk_pass(const T* __restrict__ mat, const T* __restrict__ add, const T* __restrict__ logm,
                                               int R, int L, T* __restrict__ out, T* __restrict__ absd, const SkCtl* ctl) {
    if (ctl->done) return;
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int b = blockIdx.y;
    const T* row = mat + ((size_t)b * R + r) * L;
    const T* ad = add + (size_t)b * L;
    T mx = MathT<T>::ninf();
    for (int l = lane; l < L; l += 64) {
        const T x = row[l] + ad[l];
        mx = x > mx ? x : mx;
    }
    mx = wave_max(mx);
    T s = (T)0;
    for (int l = lane; l < L; l += 64) s += MathT<T>::exp(row[l] + ad[l] - mx);
    s = wave_sum(s);
    if (lane == 0) {
        const T nv = logm[(size_t)b * R + r] - (mx + MathT<T>::log(s));
        const T old = out[(size_t)b * R + r];
        out[(size_t)b * R + r] = nv;
        if (absd) {
            const T d = nv - old;
            absd[(size_t)b * R + r] = d < (T)0 ? -d : d;
        }
    }
}

// one block: diff_b = sum |du| + sum |dv| per problem, done = (min_b diff_b < threshold); counts the iteration
template <typename T>
__global__ __launch_bounds__(256) void sk_check(const T* __restrict__ adu, const T* __restrict__ adv, int nb, int N, int M,
                                                double threshold, SkCtl* ctl) {
    __shared__ double red[4];
    __shared__ double best;
    if (ctl->done) return;
    if (threadIdx.x == 0) best = INFINITY;
    __syncthreads();
    for (int b = 0; b < nb; ++b) {
        double s = 0.0;
        for (int i = threadIdx.x; i < N; i += 256) s += (double)adu[(size_t)b * N + i];
        for (int i = threadIdx.x; i < M; i += 256) s += (double)adv[(size_t)b * M + i];
        s = wave_sum(s);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            // the reference sums in the tensor dtype; compare in that dtype
            const T d = (T)((red[0] + red[1]) + (red[2] + red[3]));
            if ((double)d < best) best = (double)d;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        ctl->iters += 1;
        if (best < threshold) ctl->done = 1;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void sk_pi(T* __restrict__ pi_cr, const T* __restrict__ u, const T* __restrict__ v, int N,
                                             int M) {
    const size_t boff = (size_t)blockIdx.y * N * M;
    const size_t total = (size_t)N * M;
    for (size_t e = blockIdx.x * (size_t)256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int i = e / M, j = e - (size_t)i * M;
        pi_cr[boff + e] = MathT<T>::exp(u[(size_t)blockIdx.y * N + i] + v[(size_t)blockIdx.y * M + j] + pi_cr[boff + e]);
    }
}

__global__ void sk_copy_iters(const SkCtl* ctl, int32_t* out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *out = ctl->iters;
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" int64_t otvae_sinkhorn_ws(int dtype, int nb, int N, int M) {
    if (nb <= 0 || N <= 0 || M <= 0 || dtype < 0 || dtype > 1) return -1;
    const size_t es = dtype ? 8 : 4;
    return (int64_t)(align256((size_t)nb * N * M * es) + 4 * align256((size_t)nb * (N > M ? N : M) * es) + 256);
}

template <typename T>
static int sinkhorn_impl(const T* a, const T* b, const T* Cm, int nb, int N, int M, double reg, int max_iter, double threshold,
                         void* ws, T* pi, T* u, T* v, int32_t* iters_done, hipStream_t st) {
    char* w = (char*)ws;
    T* crt = (T*)w;
    w += align256((size_t)nb * N * M * sizeof(T));
    const size_t vs = align256((size_t)nb * (N > M ? N : M) * sizeof(T));
    T* loga = (T*)w;
    w += vs;
    T* logb = (T*)w;
    w += vs;
    T* adu = (T*)w;
    w += vs;
    T* adv = (T*)w;
    w += vs;
    SkCtl* ctl = (SkCtl*)w;
    const bool track = threshold > 0.0;

    sk_init_mat<T><<<dim3(cdiv(M, 32), cdiv(N, 32), nb), 256, 0, st>>>(Cm, N, M, (T)(-1.0 / reg), pi, crt);
    OTVAE_CHECK_LAUNCH("otvae_sinkhorn_log(init_mat)");
    const int nv = nb * (N > M ? N : M);
    sk_init_vec<T><<<cdiv(nv, 256), 256, 0, st>>>(a, b, nb, N, M, loga, logb, u, v, ctl, track ? 0 : max_iter);
    OTVAE_CHECK_LAUNCH("otvae_sinkhorn_log(init_vec)");
    for (int it = 0; it < max_iter; ++it) {
        // v_j = log b_j - LSE_i(Cr_ij + u_i): rows of CrT
        sk_pass<T><<<dim3(cdiv(M, 4), nb), 256, 0, st>>>(crt, u, logb, M, N, v, track ? adv : nullptr, ctl);
        // u_i = log a_i - LSE_j(Cr_ij + v_j): rows of Cr (held in pi)
        sk_pass<T><<<dim3(cdiv(N, 4), nb), 256, 0, st>>>(pi, v, loga, N, M, u, track ? adu : nullptr, ctl);
        if (track) sk_check<T><<<1, 256, 0, st>>>(adu, adv, nb, N, M, threshold, ctl);
    }
    OTVAE_CHECK_LAUNCH("otvae_sinkhorn_log(iterations)");
    sk_pi<T><<<dim3(imin(cdiv((size_t)N * M, 256), 1024), nb), 256, 0, st>>>(pi, u, v, N, M);
    OTVAE_CHECK_LAUNCH("otvae_sinkhorn_log(pi)");
    if (iters_done) {
        sk_copy_iters<<<1, 64, 0, st>>>(ctl, iters_done);
        OTVAE_CHECK_LAUNCH("otvae_sinkhorn_log(iters)");
    }
    return OTVAE_OK;
}

extern "C" int otvae_sinkhorn_log(int dtype, const void* a, const void* b, const void* C, int nb, int N, int M, double reg,
                                  int max_iter, double threshold, void* ws, void* pi, void* u, void* v, int32_t* iters_done,
                                  void* stream) {
    OTVAE_REQUIRE(a && b && C && ws && pi && u && v, "otvae_sinkhorn_log: NULL argument");
    OTVAE_REQUIRE(nb > 0 && N > 0 && M > 0 && max_iter >= 0, "otvae_sinkhorn_log: bad sizes");
    OTVAE_REQUIRE(dtype == 0 || dtype == 1, "otvae_sinkhorn_log: dtype must be 0 (fp32) or 1 (fp64)");
    OTVAE_REQUIRE(reg > 0.0, "otvae_sinkhorn_log: reg must be positive");
    OTVAE_REQUIRE(pi != C, "otvae_sinkhorn_log: pi must not alias C");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0)
        return sinkhorn_impl<float>((const float*)a, (const float*)b, (const float*)C, nb, N, M, reg, max_iter, threshold, ws,
                                    (float*)pi, (float*)u, (float*)v, iters_done, st);
    return sinkhorn_impl<double>((const double*)a, (const double*)b, (const double*)C, nb, N, M, reg, max_iter, threshold, ws,
                                 (double*)pi, (double*)u, (double*)v, iters_done, st);
}

// ---- sum_ij C*pi ---------------------------------------------------------------------------------------------
#define COST_PARTS 64
template <typename T>
__global__ __launch_bounds__(256) void ot_cost_partial(const T* __restrict__ Cm, const T* __restrict__ pi, size_t total,
                                                       double* __restrict__ ws) {
    __shared__ double red[4];
    const size_t boff = (size_t)blockIdx.y * total;
    double s = 0.0;
    for (size_t e = blockIdx.x * (size_t)256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256)
        s += (double)Cm[boff + e] * (double)pi[boff + e];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) ws[(size_t)blockIdx.y * COST_PARTS + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
template <typename T>
__global__ void ot_cost_final(const double* __restrict__ ws, int parts, int nb, T* __restrict__ cost) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    double s = 0.0;
    for (int p = 0; p < parts; ++p) s += ws[(size_t)b * COST_PARTS + p];
    cost[b] = (T)s;
}

extern "C" int otvae_ot_cost(int dtype, const void* C, const void* pi, int nb, int N, int M, double* ws, void* cost,
                             void* stream) {
    OTVAE_REQUIRE(C && pi && ws && cost && nb > 0 && N > 0 && M > 0, "otvae_ot_cost: bad argument");
    OTVAE_REQUIRE(dtype == 0 || dtype == 1, "otvae_ot_cost: dtype must be 0 or 1");
    const size_t total = (size_t)N * M;
    const int parts = imin(COST_PARTS, cdiv(total, 1024));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0) {
        ot_cost_partial<float><<<dim3(parts, nb), 256, 0, st>>>((const float*)C, (const float*)pi, total, ws);
        ot_cost_final<float><<<cdiv(nb, 64), 64, 0, st>>>(ws, parts, nb, (float*)cost);
    } else {
        ot_cost_partial<double><<<dim3(parts, nb), 256, 0, st>>>((const double*)C, (const double*)pi, total, ws);
        ot_cost_final<double><<<cdiv(nb, 64), 64, 0, st>>>(ws, parts, nb, (double*)cost);
    }
    OTVAE_CHECK_LAUNCH("otvae_ot_cost");
    return OTVAE_OK;
}

// ---- pairwise squared euclidean cost: |x_i|^2 + |y_j|^2 - 2 x_i.y_j (same expansion as ot/w2_utils.py:121-125) --
template <typename T>
__global__ __launch_bounds__(256) void sqdist_kernel(const T* __restrict__ x, const T* __restrict__ y, int N, int M, int D,
                                                     T* __restrict__ Cm) {
    __shared__ T xs[16][33], ys[16][33];
    const int b = blockIdx.z;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int i = blockIdx.y * 16 + ty, j = blockIdx.x * 16 + tx;
    const T* xb = x + (size_t)b * N * D;
    const T* yb = y + (size_t)b * M * D;
    T dot = 0, xx = 0, yy = 0;
    for (int d0 = 0; d0 < D; d0 += 32) {
        for (int e = threadIdx.x; e < 16 * 32; e += 256) {
            const int r = e >> 5, d = e & 31;
            const int gi = blockIdx.y * 16 + r, gj = blockIdx.x * 16 + r;
            xs[r][d] = (gi < N && d0 + d < D) ? xb[(size_t)gi * D + d0 + d] : (T)0;
            ys[r][d] = (gj < M && d0 + d < D) ? yb[(size_t)gj * D + d0 + d] : (T)0;
        }
        __syncthreads();
#pragma unroll 8
        for (int d = 0; d < 32; ++d) {
            const T a = xs[ty][d], c = ys[tx][d];
            dot += a * c;
            xx += a * a;
            yy += c * c;
        }
        __syncthreads();
    }
    if (i < N && j < M) Cm[((size_t)b * N + i) * M + j] = xx + yy - (T)2 * dot;
}

extern "C" int otvae_sqdist(int dtype, const void* x, const void* y, int nb, int N, int M, int D, void* C, void* stream) {
    OTVAE_REQUIRE(x && y && C && nb > 0 && N > 0 && M > 0 && D > 0, "otvae_sqdist: bad argument");
    OTVAE_REQUIRE(dtype == 0 || dtype == 1, "otvae_sqdist: dtype must be 0 or 1");
    dim3 grid(cdiv(M, 16), cdiv(N, 16), nb);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0)
        sqdist_kernel<float><<<grid, 256, 0, st>>>((const float*)x, (const float*)y, N, M, D, (float*)C);
    else
        sqdist_kernel<double><<<grid, 256, 0, st>>>((const double*)x, (const double*)y, N, M, D, (double*)C);
    OTVAE_CHECK_LAUNCH("otvae_sqdist");
    return OTVAE_OK;
}
