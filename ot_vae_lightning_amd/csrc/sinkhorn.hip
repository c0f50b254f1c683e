// Log-domain Sinkhorn (reference ot/w2_utils.py:276-319) and the helpers the minibatch-OT prior composes it with.
//
//   u = v = 0; Cr = -C/reg; loop: v_j = log b_j - LSE_i(Cr_ij + u_i); u_i = log a_i - LSE_j(Cr_ij + v_j);
//   stop when min over the batch of (|du|_1 + |dv|_1) < threshold;  pi = exp(u_i + v_j + Cr_ij)
//
// Default path: each half-iteration is a row-wise log-sum-exp, and every one depends on the whole result of the previous one, so a
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
    int done;            // set by sk_check when the batch-min difference drops below the threshold
    int iters;           // iterations actually performed
    unsigned bar_count;  // persistent kernel: monotone arrival counter of the grid barrier
    int timeout;         // persistent kernel: a barrier wait ran out (results invalid; iters reports -1)
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
        ctl->bar_count = 0;
        ctl->timeout = 0;
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


// ------------------------------------------------------------------------------------------------ persistent solver
// All iterations in ONE launch.  A half-iteration needs every entry of the other potential, i.e. a grid-wide exchange;
// with one launch per half-iteration the 100 passes of the 50-iteration solve cost ~100 launch ramps (5-9 us each) for
// ~1 us of arithmetic.  Here <= one workgroup per CU stays resident, a wave owns one row of Cr and one row of Cr^T and
// (when they are <= 1024 long and there are enough waves) keeps both IN REGISTERS for the whole solve, so that per
// half-iteration only the 4 KiB potential crosses the chip: written with agent-scope (sc1, write-through) stores,
// published by a monotone arrival counter, read with agent-scope loads (cdna_hip_programming.md section 6 Guideline 16,
// sc1 form).  Every wait is bounded; the arithmetic per row is the one of sk_pass, so both paths give identical bits.
template <typename T>
struct CohT;
template <>
struct CohT<float> {
    static __device__ __forceinline__ float ld(const float* p) {
        return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    static __device__ __forceinline__ void st(float* p, float v) {
        __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
};
template <>
struct CohT<double> {
    static __device__ __forceinline__ double ld(const double* p) {
        return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                                 __HIP_MEMORY_SCOPE_AGENT));
    }
    static __device__ __forceinline__ void st(double* p, double v) {
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
};

#define SK_SPIN_LIMIT (1u << 22)

// returns false (in every thread of the block) if the wait ran out
__device__ __forceinline__ bool sk_grid_barrier(SkCtl* ctl, unsigned nblocks, unsigned& epoch) {
    __shared__ int s_ok;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through stores have been performed
    __syncthreads();
    if (threadIdx.x == 0) {
        int ok = 1;
        const unsigned target = nblocks * (epoch + 1);
        const unsigned prev = __hip_atomic_fetch_add(&ctl->bar_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev + 1 < target) {
            unsigned spins = 0;
            while (__hip_atomic_load(&ctl->bar_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > SK_SPIN_LIMIT) {
                    ok = 0;
                    __hip_atomic_store(&ctl->timeout, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        s_ok = ok;
    }
    ++epoch;
    __syncthreads();
    return s_ok != 0;
}

#define SK_EPL 16  // row elements per lane that may live in registers (rows up to 1024)

template <typename T, bool CACHE>
__global__ __launch_bounds__(256) void sk_persistent(T* __restrict__ pi_cr, const T* __restrict__ crt, const T* __restrict__ loga,
                                                     const T* __restrict__ logb, T* __restrict__ u, T* __restrict__ v,
                                                     T* __restrict__ adu, T* __restrict__ adv, int nb, int N, int M, int max_iter,
                                                     double threshold, SkCtl* ctl) {
    __shared__ double red[4];
    __shared__ double s_best;
    const int lane = threadIdx.x & 63;
    const int wave_g = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    const bool track = threshold > 0.0;
    unsigned epoch = 0;
    const int RU = nb * N, RV = nb * M;  // rows of the u pass (rows of Cr) / of the v pass (rows of Cr^T)

    // register-resident rows (CACHE: one row of each matrix per wave at most)
    T cr_row[CACHE ? SK_EPL : 1], ct_row[CACHE ? SK_EPL : 1];
    if constexpr (CACHE) {
#pragma unroll
        for (int k = 0; k < SK_EPL; ++k) {
            const int l = lane + 64 * k;
            cr_row[k] = (wave_g < RU && l < M) ? pi_cr[(size_t)wave_g * M + l] : MathT<T>::ninf();
            ct_row[k] = (wave_g < RV && l < N) ? crt[(size_t)wave_g * N + l] : MathT<T>::ninf();
        }
    }

    // one row: out[r] = logm[r] - LSE_l(row[l] + add[l]) with the potential `add` read coherently
    auto row_pass = [&](const T* __restrict__ mat, const T (&cached)[CACHE ? SK_EPL : 1], const T* __restrict__ add,
                        const T* __restrict__ logm, int R, int L, int rows_per_problem, T* __restrict__ out, T* __restrict__ absd) {
        for (int r = wave_g; r < R; r += nwaves) {
            const int b = r / rows_per_problem;
            const T* ad = add + (size_t)b * L;
            T mx = MathT<T>::ninf();
            T s = (T)0;
            if constexpr (CACHE) {
                T x[SK_EPL];
#pragma unroll
                for (int k = 0; k < SK_EPL; ++k) {
                    const int l = lane + 64 * k;
                    x[k] = l < L ? cached[k] + CohT<T>::ld(ad + l) : MathT<T>::ninf();
                    mx = x[k] > mx ? x[k] : mx;
                }
                mx = wave_max(mx);
#pragma unroll
                for (int k = 0; k < SK_EPL; ++k)
                    if (lane + 64 * k < L) s += MathT<T>::exp(x[k] - mx);
            } else {
                const T* row = mat + (size_t)r * L;
                for (int l = lane; l < L; l += 64) {
                    const T x = row[l] + CohT<T>::ld(ad + l);
                    mx = x > mx ? x : mx;
                }
                mx = wave_max(mx);
                for (int l = lane; l < L; l += 64) s += MathT<T>::exp(row[l] + CohT<T>::ld(ad + l) - mx);
            }
            s = wave_sum(s);
            if (lane == 0) {
                const T nv = logm[r] - (mx + MathT<T>::log(s));
                if (absd) {
                    const T d = nv - CohT<T>::ld(out + r);
                    CohT<T>::st(absd + r, d < (T)0 ? -d : d);
                }
                CohT<T>::st(out + r, nv);
            }
        }
    };

    int it = 0;
    bool alive = true;
    for (; it < max_iter && alive; ++it) {
        row_pass(crt, ct_row, u, logb, RV, N, M, v, track ? adv : nullptr);
        alive = sk_grid_barrier(ctl, gridDim.x, epoch);
        if (!alive) break;
        row_pass(pi_cr, cr_row, v, loga, RU, M, N, u, track ? adu : nullptr);
        alive = sk_grid_barrier(ctl, gridDim.x, epoch);
        if (!alive) break;
        if (track) {
            // every block evaluates the same test on the same numbers in the same order: a uniform decision without
            // another exchange (sk_check's arithmetic)
            if (threadIdx.x == 0) s_best = INFINITY;
            __syncthreads();
            for (int b = 0; b < nb; ++b) {
                double s = 0.0;
                for (int i = threadIdx.x; i < N; i += 256) s += (double)CohT<T>::ld(adu + (size_t)b * N + i);
                for (int i = threadIdx.x; i < M; i += 256) s += (double)CohT<T>::ld(adv + (size_t)b * M + i);
                s = wave_sum(s);
                if (lane == 0) red[threadIdx.x >> 6] = s;
                __syncthreads();
                if (threadIdx.x == 0) {
                    const T d = (T)((red[0] + red[1]) + (red[2] + red[3]));
                    if ((double)d < s_best) s_best = (double)d;
                }
                __syncthreads();
            }
            const bool stop = s_best < threshold;
            if (blockIdx.x == 0 && threadIdx.x == 0) ctl->iters = it + 1;
            __syncthreads();
            if (stop) {
                ++it;
                break;
            }
        }
    }
    if (!alive) {
        if (blockIdx.x == 0 && threadIdx.x == 0) ctl->iters = -1;
        return;
    }
    // pi = exp(u_i + v_j + Cr_ij), rows of Cr
    for (int r = wave_g; r < RU; r += nwaves) {
        const int b = r / N;
        const T ui = CohT<T>::ld(u + r);
        const T* vb = v + (size_t)b * M;
        T* row = pi_cr + (size_t)r * M;
        if constexpr (CACHE) {
#pragma unroll
            for (int k = 0; k < SK_EPL; ++k) {
                const int l = lane + 64 * k;
                if (l < M) row[l] = MathT<T>::exp(ui + CohT<T>::ld(vb + l) + cr_row[k]);
            }
        } else {
            for (int l = lane; l < M; l += 64) row[l] = MathT<T>::exp(ui + CohT<T>::ld(vb + l) + row[l]);
        }
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
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) n_cu = v;
        if (n_cu <= 0) n_cu = 1;
    }
    // Measured on MI355X (1024 x 1024 fp32, 50 iterations): persistent 1.09 ms vs 0.91 ms for one launch per
    // half-iteration -- a grid barrier + coherent (L2-bypassing) reads of the potential cost ~10 us per phase across
    // 8 XCDs, more than the ~5 us launch ramp they replace -- so the persistent solver is opt-in (OTVAE_SK_PERSISTENT=1).
    if (getenv("OTVAE_SK_PERSISTENT") && !getenv("OTVAE_SK_MULTILAUNCH")) {
        // one resident workgroup per CU at most (256 threads, no dynamic LDS: always co-resident), bounded waits
        const int rows = nb * (N > M ? N : M);
        int G = imax(1, imin(n_cu, cdiv(rows, 4)));
        if (getenv("OTVAE_SK_BLOCKS")) G = imax(1, imin(G, atoi(getenv("OTVAE_SK_BLOCKS"))));
        const bool cache = (nb * N <= 4 * G) && (nb * M <= 4 * G) && N <= 64 * SK_EPL && M <= 64 * SK_EPL;
        if (cache)
            sk_persistent<T, true><<<G, 256, 0, st>>>(pi, crt, loga, logb, u, v, adu, adv, nb, N, M, max_iter, threshold, ctl);
        else
            sk_persistent<T, false><<<G, 256, 0, st>>>(pi, crt, loga, logb, u, v, adu, adv, nb, N, M, max_iter, threshold, ctl);
        OTVAE_CHECK_LAUNCH("otvae_sinkhorn_log(persistent)");
    } else {
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
    }
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
