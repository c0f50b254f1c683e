// Differentiable log-domain Sinkhorn: the reference's ``sinkhorn_log`` (ot/w2_utils.py:276-319) is plain torch arithmetic, so
// autograd differentiates it THROUGH every iteration (with respect to a, b and C).  This file is that derivative as kernels:
//
//   forward  (otvae_sinkhorn_log_tape): the same iteration, one launch per half-iteration, every potential kept
//            (u_0 = v_0 = 0, v_{i+1} = log b - LSE_n(Cr + u_i), u_{i+1} = log a - LSE_j(Cr + v_{i+1}), K' iterations done on the device's
//            own stopping test), Cr and Cr^T kept;  pi = exp(u_K' + v_K' + Cr)
//   backward (otvae_sinkhorn_log_bwd): the reverse sweep.  With G = gpi * pi, au[i] / av[i] the adjoints of u_i / v_i:
//            au[K'] = rowsum G, av[K'] = colsum G;  for i = K'-1 .. 0:
//              av[i+1]_j -= sum_n au[i+1]_n Pu_i[n][j],   Pu_i = exp(Cr + v_{i+1} + u_{i+1} - log a)   (rows sum to one)
//              au[i]_n    = -sum_j av[i+1]_j Pv_i[n][j],  Pv_i = exp(Cr + u_i + v_{i+1} - log b)       (columns sum to one)
//            gCr = G - sum_i (au[i+1] Pu_i + av[i+1] Pv_i);  gC = -gCr / reg;  g log a = sum_i au[i+1], g log b = sum_i av[i+1];
//            ga = g log a / (a + 1e-8), gb likewise.
//   Both reverse updates are row reductions (the column one runs on Cr^T), the same shape as the forward's passes: one wave per row.
//   The adjoints of all iterations are kept and gCr is formed by ONE pass over the matrix at the end (2 K' exponentials per entry)
//   instead of 2 K' read-modify-write passes over an N x M accumulator.
//
// This is the opt-in route (``sinkhorn_log`` called with inputs that require a gradient, ``SinkhornPrior(differentiate_plan=True)``);
// the training default keeps the single-launch solver of sinkhorn.hip and the envelope gradient.
#include "common.h"

namespace {

template <typename T>
struct M_;
template <>
struct M_<float> {
    static __device__ __forceinline__ float exp(float x) { return __expf(x); }
    static __device__ __forceinline__ float log(float x) { return __logf(x); }
    static __device__ __forceinline__ float ninf() { return -INFINITY; }
};
template <>
struct M_<double> {
    static __device__ __forceinline__ double exp(double x) { return ::exp(x); }
    static __device__ __forceinline__ double log(double x) { return ::log(x); }
    static __device__ __forceinline__ double ninf() { return -(double)INFINITY; }
};

struct TapeCtl {
    int done;   // the stopping test fired
    int iters;  // iterations performed (K')
};

inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

template <typename T>
struct Tape {
    T *cr, *crt, *loga, *logb, *adu, *adv, *uh, *vh;
    TapeCtl* ctl;
};

template <typename T>
Tape<T> tape_layout(void* tape, int nb, int N, int M, int K) {
    char* w = (char*)tape;
    Tape<T> t;
    const size_t mat = al256((size_t)nb * N * M * sizeof(T));
    const size_t vn = al256((size_t)nb * N * sizeof(T)), vm = al256((size_t)nb * M * sizeof(T));
    t.cr = (T*)w;   w += mat;
    t.crt = (T*)w;  w += mat;
    t.loga = (T*)w; w += vn;
    t.logb = (T*)w; w += vm;
    t.adu = (T*)w;  w += vn;
    t.adv = (T*)w;  w += vm;
    t.uh = (T*)w;   w += al256((size_t)(K + 1) * nb * N * sizeof(T));
    t.vh = (T*)w;   w += al256((size_t)(K + 1) * nb * M * sizeof(T));
    t.ctl = (TapeCtl*)w;
    return t;
}

size_t tape_bytes(int dtype, int nb, int N, int M, int K) {
    const size_t es = dtype ? 8 : 4;
    return 2 * al256((size_t)nb * N * M * es) + 2 * (al256((size_t)nb * N * es) + al256((size_t)nb * M * es)) +
           al256((size_t)(K + 1) * nb * N * es) + al256((size_t)(K + 1) * nb * M * es) + 256;
}

// Cr = -C / reg and its transpose (32 x 32 tiles through LDS), log marginals, u_0 = v_0 = 0, the control block
template <typename T>
__global__ __launch_bounds__(256) void tape_init(const T* __restrict__ Cm, int N, int M, T inv_reg, const T* __restrict__ a,
                                                 const T* __restrict__ b, Tape<T> t, int preset_iters) {
    __shared__ T tile[32][33];
    const int pb = blockIdx.z;
    const size_t boff = (size_t)pb * N * M;
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int i = i0 + r, j = j0 + tx;
        if (i < N && j < M) {
            const T val = Cm[boff + (size_t)i * M + j] * -inv_reg;   // the same product as sk_init_mat (sinkhorn.hip): identical Cr bits
            t.cr[boff + (size_t)i * M + j] = val;
            tile[r][tx] = val;
        }
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int j = j0 + r, i = i0 + tx;
        if (i < N && j < M) t.crt[boff + (size_t)j * N + i] = tile[tx][r];
    }
    if (blockIdx.x == 0 && threadIdx.x < 32 && i0 + (int)threadIdx.x < N) {
        const size_t i = (size_t)pb * N + i0 + threadIdx.x;
        t.loga[i] = M_<T>::log((a ? a[i] : (T)1 / (T)N) + (T)1e-8);
        t.uh[i] = (T)0;
    }
    if (blockIdx.y == 0 && threadIdx.x >= 64 && threadIdx.x < 96 && j0 + (int)threadIdx.x - 64 < M) {
        const size_t j = (size_t)pb * M + j0 + threadIdx.x - 64;
        t.logb[j] = M_<T>::log((b ? b[j] : (T)1 / (T)M) + (T)1e-8);
        t.vh[j] = (T)0;
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 128) {
        t.ctl->done = 0;
        t.ctl->iters = preset_iters;
    }
}

// out[r] = logm[r] - LSE_l(mat[r][l] + add[l]); absd[r] = |out[r] - prev[r]|.  One wave per row (the arithmetic of sk_pass).
template <typename T>
__global__ __launch_bounds__(256) void tape_pass(const T* __restrict__ mat, const T* __restrict__ add, const T* __restrict__ logm, int R,
                                                 int L, const T* __restrict__ prev, T* __restrict__ out, T* __restrict__ absd,
                                                 const TapeCtl* ctl) {
    if (ctl->done) return;
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int b = blockIdx.y;
    const T* row = mat + ((size_t)b * R + r) * L;
    const T* ad = add + (size_t)b * L;
    T mx = M_<T>::ninf();
    for (int l = lane; l < L; l += 64) {
        const T x = row[l] + ad[l];
        mx = x > mx ? x : mx;
    }
    mx = wave_max(mx);
    T s = (T)0;
    for (int l = lane; l < L; l += 64) s += M_<T>::exp(row[l] + ad[l] - mx);
    s = wave_sum(s);
    if (lane == 0) {
        const T nv = logm[(size_t)b * R + r] - (mx + M_<T>::log(s));
        out[(size_t)b * R + r] = nv;
        if (absd) {
            const T d = nv - prev[(size_t)b * R + r];
            absd[(size_t)b * R + r] = d < (T)0 ? -d : d;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void tape_check(const T* __restrict__ adu, const T* __restrict__ adv, int nb, int N, int M,
                                                  double threshold, int track, TapeCtl* ctl) {
    __shared__ double red[4];
    __shared__ double best;
    if (ctl->done) return;
    if (!track) return;
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
            const T d = (T)((red[0] + red[1]) + (red[2] + red[3]));  // the reference sums in the tensor dtype
            if ((double)d < best) best = (double)d;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        ctl->iters += 1;
        if (best < threshold) ctl->done = 1;
    }
}

// pi = exp(u_K' + v_K' + Cr); the first block column also copies the final potentials out
template <typename T>
__global__ __launch_bounds__(256) void tape_pi(Tape<T> t, int nb, int N, int M, T* __restrict__ pi, T* __restrict__ u_out,
                                               T* __restrict__ v_out, int32_t* __restrict__ iters_out) {
    const int K = t.ctl->iters;
    const int b = blockIdx.y;
    const T* u = t.uh + ((size_t)K * nb + b) * N;
    const T* v = t.vh + ((size_t)K * nb + b) * M;
    const size_t boff = (size_t)b * N * M, total = (size_t)N * M;
    for (size_t e = blockIdx.x * (size_t)256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int i = e / M, j = e - (size_t)i * M;
        pi[boff + e] = M_<T>::exp(u[i] + v[j] + t.cr[boff + e]);
    }
    if (blockIdx.x == 0) {
        if (u_out) for (int i = threadIdx.x; i < N; i += 256) u_out[(size_t)b * N + i] = u[i];
        if (v_out) for (int j = threadIdx.x; j < M; j += 256) v_out[(size_t)b * M + j] = v[j];
        if (iters_out && b == 0 && threadIdx.x == 0) *iters_out = K;
    }
}

// ---- reverse sweep -------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void zero_kernel(T* __restrict__ p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = (T)0;
}

// au[K'][n] = sum_j gpi[n][j] pi[n][j]   (one wave per row)
template <typename T>
__global__ __launch_bounds__(256) void bwd_rowsum(const T* __restrict__ gpi, const T* __restrict__ pi, int nb, int N, int M,
                                                  const TapeCtl* ctl, T* __restrict__ auh) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= N) return;
    const int b = blockIdx.y;
    const size_t off = ((size_t)b * N + r) * M;
    T s = (T)0;
    for (int l = lane; l < M; l += 64) s += gpi[off + l] * pi[off + l];
    s = wave_sum(s);
    if (lane == 0) auh[((size_t)ctl->iters * nb + b) * N + r] = s;
}

// column sums of gpi * pi over a chunk of rows: part[ch][b][j]; combined in chunk order by bwd_colsum_final
#define COL_CHUNKS 16
template <typename T>
__global__ __launch_bounds__(256) void bwd_colsum(const T* __restrict__ gpi, const T* __restrict__ pi, int nb, int N, int M,
                                                  T* __restrict__ part) {
    const int j = blockIdx.x * 256 + threadIdx.x, ch = blockIdx.y, b = blockIdx.z;
    if (j >= M) return;
    const int per = (N + COL_CHUNKS - 1) / COL_CHUNKS;
    const int n0 = ch * per, n1 = n0 + per < N ? n0 + per : N;
    T s = (T)0;
    for (int n = n0; n < n1; ++n) {
        const size_t e = ((size_t)b * N + n) * M + j;
        s += gpi[e] * pi[e];
    }
    part[((size_t)ch * nb + b) * M + j] = s;
}
template <typename T>
__global__ __launch_bounds__(256) void bwd_colsum_final(const T* __restrict__ part, int nb, int M, const TapeCtl* ctl, T* __restrict__ avh) {
    const int j = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (j >= M) return;
    T s = (T)0;
    for (int ch = 0; ch < COL_CHUNKS; ++ch) s += part[((size_t)ch * nb + b) * M + j];
    avh[((size_t)ctl->iters * nb + b) * M + j] = s;
}

// out[r] = acc[r] - sum_l adj[l] exp(mat[r][l] + pot[l] - logm[l] + rowpot[r])   for iteration i < K' (else nothing).  One wave per row.
template <typename T>
__global__ __launch_bounds__(256) void bwd_pass(const T* __restrict__ mat, int R, int L, const T* __restrict__ adj, const T* __restrict__ pot,
                                                const T* __restrict__ logm, const T* __restrict__ rowpot, const T* acc, T* out,
                                                int it, const TapeCtl* ctl) {
    if (it >= ctl->iters) return;
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int b = blockIdx.y;
    const T* row = mat + ((size_t)b * R + r) * L;
    const size_t lo = (size_t)b * L;
    const T rp = rowpot[(size_t)b * R + r];
    T s = (T)0;
    for (int l = lane; l < L; l += 64) s += adj[lo + l] * M_<T>::exp(row[l] + pot[lo + l] - logm[lo + l] + rp);
    s = wave_sum(s);
    if (lane == 0) out[(size_t)b * R + r] = (acc ? acc[(size_t)b * R + r] : (T)0) - s;
}

// gC[n][j] = -(1/reg) (gpi pi - sum_{i < K'} (au[i+1][n] exp(cr + v_{i+1}[j] + u_{i+1}[n] - la[n]) + av[i+1][j] exp(cr + u_i[n] + v_{i+1}[j] - lb[j])))
// block = 4 rows x 64 columns, one entry per thread; the per-iteration row / column operands go through LDS
template <typename T>
__global__ __launch_bounds__(256) void bwd_final(const T* __restrict__ gpi, const T* __restrict__ pi, Tape<T> t, const T* __restrict__ auh,
                                                 const T* __restrict__ avh, int nb, int N, int M, T inv_reg, T* __restrict__ gC) {
    __shared__ T rowv[3][4];   // u_{i+1} - la, au[i+1], u_i      of the block's 4 rows
    __shared__ T colv[3][64];  // v_{i+1}, av[i+1], lb            of the block's 64 columns
    const int b = blockIdx.z;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int n = blockIdx.y * 4 + ty, j = blockIdx.x * 64 + tx;
    const bool in = n < N && j < M;
    const size_t e = ((size_t)b * N + (in ? n : 0)) * M + (in ? j : 0);
    const int K = t.ctl->iters;
    const T cr = in ? t.cr[e] : (T)0;
    T acc = in ? gpi[e] * pi[e] : (T)0;
    if (threadIdx.x < 64) colv[2][tx] = j < M ? t.logb[(size_t)b * M + j] : (T)0;
    for (int i = 0; i < K; ++i) {
        __syncthreads();
        if (threadIdx.x < 64 && j < M) {
            colv[0][tx] = t.vh[((size_t)(i + 1) * nb + b) * M + j];
            colv[1][tx] = avh[((size_t)(i + 1) * nb + b) * M + j];
        }
        if (threadIdx.x >= 64 && threadIdx.x < 68) {
            const int rn = blockIdx.y * 4 + (threadIdx.x - 64);
            if (rn < N) {
                rowv[0][threadIdx.x - 64] = t.uh[((size_t)(i + 1) * nb + b) * N + rn] - t.loga[(size_t)b * N + rn];
                rowv[1][threadIdx.x - 64] = auh[((size_t)(i + 1) * nb + b) * N + rn];
                rowv[2][threadIdx.x - 64] = t.uh[((size_t)i * nb + b) * N + rn];
            }
        }
        __syncthreads();
        if (in) {
            acc -= rowv[1][ty] * M_<T>::exp(cr + colv[0][tx] + rowv[0][ty]);
            acc -= colv[1][tx] * M_<T>::exp(cr + rowv[2][ty] + colv[0][tx] - colv[2][tx]);
        }
    }
    if (in) gC[e] = -inv_reg * acc;
}

// g[r] = (sum_{i=1..K'} adj[i][r]) / (marg[r] + 1e-8)
template <typename T>
__global__ __launch_bounds__(256) void bwd_marginal(const T* __restrict__ adjh, const T* __restrict__ marg, int nb, int R,
                                                    const TapeCtl* ctl, T* __restrict__ g) {
    const int r = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (r >= R) return;
    const int K = ctl->iters;
    T s = (T)0;
    for (int i = 1; i <= K; ++i) s += adjh[((size_t)i * nb + b) * R + r];
    g[(size_t)b * R + r] = s / (marg[(size_t)b * R + r] + (T)1e-8);
}

template <typename T>
int tape_forward(const T* a, const T* b, const T* Cm, int nb, int N, int M, double reg, int K, double threshold, void* tape, T* pi, T* u,
                 T* v, int32_t* iters_done, hipStream_t st) {
    Tape<T> t = tape_layout<T>(tape, nb, N, M, K);
    const bool track = threshold > 0.0;
    tape_init<T><<<dim3(cdiv(M, 32), cdiv(N, 32), nb), 256, 0, st>>>(Cm, N, M, (T)(1.0 / reg), a, b, t, track ? 0 : K);
    OTVAE_CHECK_LAUNCH("otvae_sinkhorn_log_tape(init)");
    for (int i = 0; i < K; ++i) {
        const T *ui = t.uh + (size_t)i * nb * N, *vi = t.vh + (size_t)i * nb * M;
        T *un = t.uh + (size_t)(i + 1) * nb * N, *vn = t.vh + (size_t)(i + 1) * nb * M;
        tape_pass<T><<<dim3(cdiv(M, 4), nb), 256, 0, st>>>(t.crt, ui, t.logb, M, N, vi, vn, track ? t.adv : nullptr, t.ctl);
        tape_pass<T><<<dim3(cdiv(N, 4), nb), 256, 0, st>>>(t.cr, vn, t.loga, N, M, ui, un, track ? t.adu : nullptr, t.ctl);
        if (track) tape_check<T><<<1, 256, 0, st>>>(t.adu, t.adv, nb, N, M, threshold, 1, t.ctl);
    }
    OTVAE_CHECK_LAUNCH("otvae_sinkhorn_log_tape(iterations)");
    tape_pi<T><<<dim3(imin(cdiv((int64_t)N * M, 256), 1024), nb), 256, 0, st>>>(t, nb, N, M, pi, u, v, iters_done);
    OTVAE_CHECK_LAUNCH("otvae_sinkhorn_log_tape(pi)");
    return OTVAE_OK;
}

template <typename T>
int tape_backward(const T* gpi, const T* pi, const T* a, const T* b, int nb, int N, int M, double reg, int K, void* tape, void* ws, T* gC,
                  T* ga, T* gb, hipStream_t st) {
    Tape<T> t = tape_layout<T>(tape, nb, N, M, K);
    char* w = (char*)ws;
    const size_t nau = (size_t)(K + 1) * nb * N, nav = (size_t)(K + 1) * nb * M;
    T* auh = (T*)w;
    w += al256(nau * sizeof(T));
    T* avh = (T*)w;
    w += al256(nav * sizeof(T));
    T* part = (T*)w;
    const size_t nz = (al256(nau * sizeof(T)) + al256(nav * sizeof(T))) / sizeof(T);   // both adjoint histories, contiguous
    zero_kernel<T><<<imin(cdiv((int64_t)nz, 256), 1024), 256, 0, st>>>(auh, nz);
    bwd_rowsum<T><<<dim3(cdiv(N, 4), nb), 256, 0, st>>>(gpi, pi, nb, N, M, t.ctl, auh);
    bwd_colsum<T><<<dim3(cdiv(M, 256), COL_CHUNKS, nb), 256, 0, st>>>(gpi, pi, nb, N, M, part);
    bwd_colsum_final<T><<<dim3(cdiv(M, 256), nb), 256, 0, st>>>(part, nb, M, t.ctl, avh);
    OTVAE_CHECK_LAUNCH("otvae_sinkhorn_log_bwd(seed)");
    for (int i = K - 1; i >= 0; --i) {
        T* av1 = avh + (size_t)(i + 1) * nb * M;
        const T* au1 = auh + (size_t)(i + 1) * nb * N;
        const T *u1 = t.uh + (size_t)(i + 1) * nb * N, *v1 = t.vh + (size_t)(i + 1) * nb * M, *u0 = t.uh + (size_t)i * nb * N;
        // av[i+1] -= Pu_i^T au[i+1]: rows of Cr^T (one per column j), operands along n
        bwd_pass<T><<<dim3(cdiv(M, 4), nb), 256, 0, st>>>(t.crt, M, N, au1, u1, t.loga, v1, av1, av1, i, t.ctl);
        // au[i] = -Pv_i av[i+1]: rows of Cr, operands along j
        bwd_pass<T><<<dim3(cdiv(N, 4), nb), 256, 0, st>>>(t.cr, N, M, av1, v1, t.logb, u0, (const T*)nullptr, auh + (size_t)i * nb * N, i, t.ctl);
    }
    OTVAE_CHECK_LAUNCH("otvae_sinkhorn_log_bwd(sweep)");
    if (gC) {
        bwd_final<T><<<dim3(cdiv(M, 64), cdiv(N, 4), nb), 256, 0, st>>>(gpi, pi, t, auh, avh, nb, N, M, (T)(1.0 / reg), gC);
        OTVAE_CHECK_LAUNCH("otvae_sinkhorn_log_bwd(cost gradient)");
    }
    if (ga && a) bwd_marginal<T><<<dim3(cdiv(N, 256), nb), 256, 0, st>>>(auh, a, nb, N, t.ctl, ga);
    if (gb && b) bwd_marginal<T><<<dim3(cdiv(M, 256), nb), 256, 0, st>>>(avh, b, nb, M, t.ctl, gb);
    OTVAE_CHECK_LAUNCH("otvae_sinkhorn_log_bwd(marginals)");
    return OTVAE_OK;
}

}  // namespace

extern "C" int64_t otvae_sinkhorn_tape_bytes(int dtype, int nb, int N, int M, int max_iter) {
    if (nb <= 0 || N <= 0 || M <= 0 || max_iter < 0 || dtype < 0 || dtype > 1) return -1;
    return (int64_t)tape_bytes(dtype, nb, N, M, max_iter);
}

extern "C" int64_t otvae_sinkhorn_bwd_ws(int dtype, int nb, int N, int M, int max_iter) {
    if (nb <= 0 || N <= 0 || M <= 0 || max_iter < 0 || dtype < 0 || dtype > 1) return -1;
    const size_t es = dtype ? 8 : 4;
    return (int64_t)(al256((size_t)(max_iter + 1) * nb * N * es) + al256((size_t)(max_iter + 1) * nb * M * es) +
                     al256((size_t)COL_CHUNKS * nb * M * es) + 256);
}

extern "C" int otvae_sinkhorn_log_tape(int dtype, const void* a, const void* b, const void* C, int nb, int N, int M, double reg,
                                       int max_iter, double threshold, void* tape, void* pi, void* u, void* v, int32_t* iters_done,
                                       void* stream) {
    OTVAE_REQUIRE(C && tape && pi, "otvae_sinkhorn_log_tape: NULL argument");
    OTVAE_REQUIRE(nb > 0 && N > 0 && M > 0 && max_iter >= 0, "otvae_sinkhorn_log_tape: bad sizes");
    OTVAE_REQUIRE(dtype == 0 || dtype == 1, "otvae_sinkhorn_log_tape: dtype must be 0 (fp32) or 1 (fp64)");
    OTVAE_REQUIRE(reg > 0.0, "otvae_sinkhorn_log_tape: reg must be positive");
    OTVAE_REQUIRE(pi != C, "otvae_sinkhorn_log_tape: pi must not alias C");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0)
        return tape_forward<float>((const float*)a, (const float*)b, (const float*)C, nb, N, M, reg, max_iter, threshold, tape, (float*)pi,
                                   (float*)u, (float*)v, iters_done, st);
    return tape_forward<double>((const double*)a, (const double*)b, (const double*)C, nb, N, M, reg, max_iter, threshold, tape, (double*)pi,
                                (double*)u, (double*)v, iters_done, st);
}

extern "C" int otvae_sinkhorn_log_bwd(int dtype, const void* gpi, const void* pi, const void* a, const void* b, int nb, int N, int M,
                                      double reg, int max_iter, void* tape, void* ws, void* gC, void* ga, void* gb, void* stream) {
    OTVAE_REQUIRE(gpi && pi && tape && ws, "otvae_sinkhorn_log_bwd: NULL argument");
    OTVAE_REQUIRE(nb > 0 && N > 0 && M > 0 && max_iter >= 0, "otvae_sinkhorn_log_bwd: bad sizes");
    OTVAE_REQUIRE(dtype == 0 || dtype == 1, "otvae_sinkhorn_log_bwd: dtype must be 0 (fp32) or 1 (fp64)");
    OTVAE_REQUIRE(reg > 0.0, "otvae_sinkhorn_log_bwd: reg must be positive");
    OTVAE_REQUIRE((!ga || a) && (!gb || b), "otvae_sinkhorn_log_bwd: a marginal's gradient needs the marginal");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0)
        return tape_backward<float>((const float*)gpi, (const float*)pi, (const float*)a, (const float*)b, nb, N, M, reg, max_iter, tape, ws,
                                    (float*)gC, (float*)ga, (float*)gb, st);
    return tape_backward<double>((const double*)gpi, (const double*)pi, (const double*)a, (const double*)b, nb, N, M, reg, max_iter, tape,
                                 ws, (double*)gC, (double*)ga, (double*)gb, st);
}
