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
    int timeout;         // persistent kernel: a wait ran out (the plan, the potentials and the cost are poisoned with NaN; iters = -1)
    unsigned spin_limit; // persistent kernel: polls a wait may spend before it gives up (OTVAE_SK_SPIN_LIMIT, default 2^22)
};

// Cr = -C/(reg * cmax) (row major, into `cr`) and its transpose (into `crt`), 32x32 tiles through LDS.  cmax = the
// maximum of problem b's cost matrix, reduced here from the P partial maxima its producer left (otvae_sqdist's epilogue),
// or 1 when pmax == NULL: the `cost_matrix / max_per_mat` of batch_ot_gmm (ot/w2_utils.py:265-266) without a pass of its
// own.  The blocks of the first tile row / column also initialise the potentials and the log-marginals (a, b == NULL:
// uniform 1/N, 1/M), block (0,0,0) the control block.
template <typename T>
__global__ __launch_bounds__(256) void sk_init_mat(const T* __restrict__ Cm, int N, int M, T inv_reg, const T* __restrict__ pmax, int P,
                                                   T* __restrict__ cr, T* __restrict__ crt, const T* __restrict__ a,
                                                   const T* __restrict__ b, T* __restrict__ loga, T* __restrict__ logb,
                                                   T* __restrict__ u, T* __restrict__ v, SkCtl* ctl, int preset_iters,
                                                   unsigned spin_limit, unsigned long long* __restrict__ tu,
                                                   unsigned long long* __restrict__ tv, T* __restrict__ scale_out) {
    __shared__ T tile[32][33];
    __shared__ T s_red[4];
    const int pb = blockIdx.z;
    const size_t boff = (size_t)pb * N * M;
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    T neg_scale = -inv_reg;
    if (pmax) {
        T mx = MathT<T>::ninf();
        for (int i = threadIdx.x; i < P; i += 256) {
            const T q = pmax[(size_t)pb * P + i];
            mx = q > mx ? q : mx;
        }
        mx = wave_max(mx);
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = mx;
        __syncthreads();
        const T m01 = s_red[0] > s_red[1] ? s_red[0] : s_red[1], m23 = s_red[2] > s_red[3] ? s_red[2] : s_red[3];
        const T cmax = m01 > m23 ? m01 : m23;
        neg_scale = -inv_reg / cmax;
        if (scale_out && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) scale_out[pb] = cmax;
    }
    for (int r = ty; r < 32; r += 8) {
        const int i = i0 + r, j = j0 + tx;
        if (i < N && j < M) {
            const T val = Cm[boff + (size_t)i * M + j] * neg_scale;
            cr[boff + (size_t)i * M + j] = val;
            tile[r][tx] = val;
        }
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int j = j0 + r, i = i0 + tx;
        if (i < N && j < M) crt[boff + (size_t)j * N + i] = tile[tx][r];
    }
    if (blockIdx.x == 0 && threadIdx.x < 32 && i0 + (int)threadIdx.x < N) {
        const size_t i = (size_t)pb * N + i0 + threadIdx.x;
        loga[i] = MathT<T>::log((a ? a[i] : (T)1 / (T)N) + (T)1e-8);
        u[i] = (T)0;
        if (tu) tu[i] = 0ull;  // {epoch 0, +0.0f}: the tagged potentials of sk_persistent_tagged
    }
    if (blockIdx.y == 0 && threadIdx.x >= 64 && threadIdx.x < 96 && j0 + (int)threadIdx.x - 64 < M) {
        const size_t j = (size_t)pb * M + j0 + threadIdx.x - 64;
        logb[j] = MathT<T>::log((b ? b[j] : (T)1 / (T)M) + (T)1e-8);
        v[j] = (T)0;
        if (tv) tv[j] = 0ull;
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 128) {
        ctl->done = 0;
        ctl->iters = preset_iters;
        ctl->bar_count = 0;
        ctl->timeout = 0;
        ctl->spin_limit = spin_limit;
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

#define SK_SPIN_LIMIT (1u << 22)  // default of SkCtl::spin_limit

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
            const unsigned limit = ctl->spin_limit;
            while (__hip_atomic_load(&ctl->bar_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > limit) {
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

// NTH threads per workgroup (NTH / 64 waves), every wave owns RPW rows of Cr and RPW rows of Cr^T (rows wave_g + q * nwaves)
// and, with CACHE, keeps them in registers for the whole solve.  POTLDS: the potential of the other side is fetched ONCE per
// workgroup and half-iteration (one coherent load per thread) into LDS instead of 16 coherent loads per lane and row.
// Few large workgroups (32 x 1024 threads for a 1024 x 1024 problem) keep the barrier cheap: its cost is the serialised
// arrivals on one counter plus one round trip, and it was ~10 us with 256 workgroups.
template <typename T, bool CACHE, int RPW, int NTH, bool POTLDS>
__global__ __launch_bounds__(NTH) void sk_persistent(T* __restrict__ pi_cr, const T* __restrict__ crt, const T* __restrict__ loga,
                                                     const T* __restrict__ logb, T* __restrict__ u, T* __restrict__ v,
                                                     T* __restrict__ adu, T* __restrict__ adv, int nb, int N, int M, int max_iter,
                                                     double threshold, SkCtl* ctl) {
    extern __shared__ __align__(16) char sk_smem[];
    T* pot = reinterpret_cast<T*>(sk_smem);  // POTLDS: nb * max(N, M) entries
    __shared__ double red[4];
    __shared__ double s_best;
    constexpr int WPB = NTH / 64;
    const int lane = threadIdx.x & 63;
    const int wave_g = blockIdx.x * WPB + (threadIdx.x >> 6), nwaves = gridDim.x * WPB;
    const bool track = threshold > 0.0;
    unsigned epoch = 0;
    const int RU = nb * N, RV = nb * M;  // rows of the u pass (rows of Cr) / of the v pass (rows of Cr^T)

    // register-resident rows (CACHE: RPW rows of each matrix per wave at most)
    constexpr int NC = CACHE ? RPW : 1, EC = CACHE ? SK_EPL : 1;
    T cr_row[NC][EC], ct_row[NC][EC];
    if constexpr (CACHE) {
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            const int r = wave_g + q * nwaves;
#pragma unroll
            for (int k = 0; k < SK_EPL; ++k) {
                const int l = lane + 64 * k;
                cr_row[q][k] = (r < RU && l < M) ? pi_cr[(size_t)r * M + l] : MathT<T>::ninf();
                ct_row[q][k] = (r < RV && l < N) ? crt[(size_t)r * N + l] : MathT<T>::ninf();
            }
        }
    }

    // one pass: out[r] = logm[r] - LSE_l(row[l] + add[l]) with the potential `add` read coherently
    auto row_pass = [&](const T* __restrict__ mat, const T (&cached)[NC][EC], const T* __restrict__ add,
                        const T* __restrict__ logm, int R, int L, int rows_per_problem, T* __restrict__ out, T* __restrict__ absd) {
        if constexpr (POTLDS) {
            __syncthreads();  // the previous pass's readers are done with pot
            for (int i = threadIdx.x; i < nb * L; i += NTH) pot[i] = CohT<T>::ld(add + i);
            __syncthreads();
        }
        int q = 0;
        for (int r = wave_g; r < R; r += nwaves, ++q) {
            const int b = r / rows_per_problem;
            const T* ad = add + (size_t)b * L;
            const T* pl = pot + (size_t)b * L;
            auto potential = [&](int l) -> T { return POTLDS ? pl[l] : CohT<T>::ld(ad + l); };
            T mx = MathT<T>::ninf();
            T s = (T)0;
            if constexpr (CACHE) {
                T x[SK_EPL];
#pragma unroll
                for (int qq = 0; qq < RPW; ++qq) {  // q is wave-uniform: a compile-time index into the register rows
                    if (qq != q) continue;
#pragma unroll
                    for (int k = 0; k < SK_EPL; ++k) {
                        const int l = lane + 64 * k;
                        x[k] = l < L ? cached[qq][k] + potential(l) : MathT<T>::ninf();
                        mx = x[k] > mx ? x[k] : mx;
                    }
                }
                mx = wave_max(mx);
#pragma unroll
                for (int k = 0; k < SK_EPL; ++k)
                    if (lane + 64 * k < L) s += MathT<T>::exp(x[k] - mx);
            } else {
                const T* row = mat + (size_t)r * L;
                for (int l = lane; l < L; l += 64) {
                    const T x = row[l] + potential(l);
                    mx = x > mx ? x : mx;
                }
                mx = wave_max(mx);
                for (int l = lane; l < L; l += 64) s += MathT<T>::exp(row[l] + potential(l) - mx);
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
            // another exchange (sk_check's arithmetic: 256 threads, 4 waves, whatever the workgroup size)
            if (threadIdx.x == 0) s_best = INFINITY;
            __syncthreads();
            for (int b = 0; b < nb; ++b) {
                if (threadIdx.x < 256) {
                    double s = 0.0;
                    for (int i = threadIdx.x; i < N; i += 256) s += (double)CohT<T>::ld(adu + (size_t)b * N + i);
                    for (int i = threadIdx.x; i < M; i += 256) s += (double)CohT<T>::ld(adv + (size_t)b * M + i);
                    s = wave_sum(s);
                    if (lane == 0) red[threadIdx.x >> 6] = s;
                }
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
    int q = 0;
    for (int r = wave_g; r < RU; r += nwaves, ++q) {
        const int b = r / N;
        const T ui = CohT<T>::ld(u + r);
        const T* vb = v + (size_t)b * M;
        T* row = pi_cr + (size_t)r * M;
        if constexpr (CACHE) {
#pragma unroll
            for (int qq = 0; qq < RPW; ++qq) {
                if (qq != q) continue;
#pragma unroll
                for (int k = 0; k < SK_EPL; ++k) {
                    const int l = lane + 64 * k;
                    if (l < M) row[l] = MathT<T>::exp(ui + CohT<T>::ld(vb + l) + cr_row[qq][k]);
                }
            }
        } else {
            for (int l = lane; l < M; l += 64) row[l] = MathT<T>::exp(ui + CohT<T>::ld(vb + l) + row[l]);
        }
    }
}

// Flag-in-data variant (fp32, fixed iteration count): every potential entry travels as ONE 64-bit word {epoch, value}
// written with a single agent-scope store, and the consumers poll the words they need until the epoch of the phase
// they wait for shows up.  There is no barrier: a half-iteration costs one store -> poll round trip instead of store ->
// arrive -> poll the counter -> reload (4 round trips, ~5.7 us).  A workgroup may only write epoch p+1 of a potential
// after it has read every entry of the OTHER potential at epoch p, which every workgroup writes only after reading the
// first one at epoch p-1: the data dependence itself keeps a word from being overwritten before all its readers have
// it.  Every poll is bounded (SK_SPIN_LIMIT); the arithmetic per row is sk_pass's.
template <int RPW, int NTH>
__global__ __launch_bounds__(NTH) void sk_persistent_tagged(float* __restrict__ pi_cr, const float* __restrict__ crt,
                                                            const float* __restrict__ loga, const float* __restrict__ logb,
                                                            float* __restrict__ u, float* __restrict__ v,
                                                            unsigned long long* __restrict__ tu, unsigned long long* __restrict__ tv,
                                                            int nb, int N, int M, int max_iter, SkCtl* ctl) {
    extern __shared__ __align__(16) char sk_smem[];
    float* pot = reinterpret_cast<float*>(sk_smem);
    __shared__ int s_fail;
    constexpr int WPB = NTH / 64;
    const int lane = threadIdx.x & 63;
    const int wave_g = blockIdx.x * WPB + (threadIdx.x >> 6), nwaves = gridDim.x * WPB;
    const int RU = nb * N, RV = nb * M;
    float cr_row[RPW][SK_EPL], ct_row[RPW][SK_EPL];
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
        const int r = wave_g + q * nwaves;
#pragma unroll
        for (int k = 0; k < SK_EPL; ++k) {
            const int l = lane + 64 * k;
            cr_row[q][k] = (r < RU && l < M) ? pi_cr[(size_t)r * M + l] : -INFINITY;
            ct_row[q][k] = (r < RV && l < N) ? crt[(size_t)r * N + l] : -INFINITY;
        }
    }
    if (threadIdx.x == 0) s_fail = 0;
    const unsigned spin_limit = ctl->spin_limit;
    __syncthreads();

    // fetch all `count` entries of a tagged potential at epoch `want` into LDS; false (block-uniform) if a poll ran out
    auto fetch = [&](const unsigned long long* __restrict__ src, int count, unsigned want) -> bool {
        __syncthreads();  // the previous pass's readers are done with pot
        // a thread polls up to 4 words at once (their loads in flight together), round after round
        for (int i0 = threadIdx.x; i0 < count; i0 += 4 * NTH) {
            unsigned long long w[4];
            unsigned pending = 0, spins = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (i0 + j * NTH < count) pending |= 1u << j;
            while (pending) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (pending & (1u << j)) w[j] = __hip_atomic_load(src + i0 + j * NTH, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if ((pending & (1u << j)) && (unsigned)(w[j] >> 32) == want) {
                        pot[i0 + j * NTH] = __uint_as_float((unsigned)w[j]);
                        pending &= ~(1u << j);
                    }
                if (pending) {
                    if (++spins > spin_limit || __hip_atomic_load(&ctl->timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                        s_fail = 1;
                        __hip_atomic_store(&ctl->timeout, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
        }
        __syncthreads();
        return s_fail == 0;
    };
    float own_u[RPW];  // the u this wave computed last for the rows it owns (all lanes): what the plan needs at the end
#pragma unroll
    for (int q = 0; q < RPW; ++q) own_u[q] = 0.f;
    auto rows_pass = [&](const float (&cached)[RPW][SK_EPL], const float* __restrict__ logm, int R, int L, int rows_per_problem,
                         float* __restrict__ out, unsigned long long* __restrict__ tout, unsigned epoch, bool keep) {
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            const int r = wave_g + q * nwaves;
            if (r >= R) continue;
            const float* pl = pot + (size_t)(r / rows_per_problem) * L;
            float x[SK_EPL], mx = -INFINITY, s = 0.f;
#pragma unroll
            for (int k = 0; k < SK_EPL; ++k) {
                const int l = lane + 64 * k;
                x[k] = l < L ? cached[q][k] + pl[l] : -INFINITY;
                mx = x[k] > mx ? x[k] : mx;
            }
            mx = wave_max(mx);
#pragma unroll
            for (int k = 0; k < SK_EPL; ++k)
                if (lane + 64 * k < L) s += MathT<float>::exp(x[k] - mx);
            s = wave_sum(s);
            const float nv = logm[r] - (mx + MathT<float>::log(s));  // every lane holds the wave-wide mx and s
            if (keep) own_u[q] = nv;
            if (lane == 0) {
                __hip_atomic_store(tout + r, ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(nv),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                out[r] = nv;
            }
        }
    };

    bool alive = true;
    unsigned epoch = 0;  // epoch of the last completed half-iteration
    for (int it = 0; it < max_iter && alive; ++it) {
        alive = fetch(tu, RU, epoch == 0 ? 0u : epoch);          // u as of the previous u pass (epoch 2 it), 0 initially
        if (!alive) break;
        rows_pass(ct_row, logb, RV, N, M, v, tv, epoch + 1, false);  // v pass = epoch 2 it + 1
        alive = fetch(tv, RV, epoch + 1);
        if (!alive) break;
        rows_pass(cr_row, loga, RU, M, N, u, tu, epoch + 2, true);   // u pass = epoch 2 it + 2
        epoch += 2;
    }
    if (!alive) {
        if (blockIdx.x == 0 && threadIdx.x == 0) ctl->iters = -1;
        return;
    }
    // pi = exp(u_i + v_j + Cr_ij): the final v is in LDS already (fetched for the last u pass); the final u of the other
    // workgroups' rows is not needed -- every wave writes the rows it owns, whose u it computed itself
    if (max_iter == 0) alive = fetch(tv, RV, 0u);  // v = 0 everywhere
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
        const int r = wave_g + q * nwaves;
        if (r >= RU) continue;
        const float ui = own_u[q];
        const float* pl = pot + (size_t)(r / N) * M;
        float* row = pi_cr + (size_t)r * M;
#pragma unroll
        for (int k = 0; k < SK_EPL; ++k) {
            const int l = lane + 64 * k;
            if (l < M) row[l] = MathT<float>::exp(ui + pl[l] + cr_row[q][k]);
        }
    }
}

__global__ void sk_copy_iters(const SkCtl* ctl, int32_t* out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *out = ctl->iters;
}

// A persistent solve whose wait ran out has produced nothing: make that impossible to miss.  The plan and the potentials
// become NaN (so does every loss computed from them) and iters reports -1; the host wrapper raises when it can look.
template <typename T>
__global__ __launch_bounds__(256) void sk_finish(SkCtl* ctl, T* __restrict__ pi, size_t total, T* __restrict__ u, size_t nu,
                                                 T* __restrict__ v, size_t nv, int32_t* __restrict__ iters_out) {
    const bool starved = __hip_atomic_load(&ctl->timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    if (iters_out && blockIdx.x == 0 && threadIdx.x == 0) *iters_out = starved ? -1 : ctl->iters;
    if (!starved) return;
    const T nan = (T)NAN;
    for (size_t e = blockIdx.x * (size_t)256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) pi[e] = nan;
    for (size_t e = blockIdx.x * (size_t)256 + threadIdx.x; e < nu; e += (size_t)gridDim.x * 256) u[e] = nan;
    for (size_t e = blockIdx.x * (size_t)256 + threadIdx.x; e < nv; e += (size_t)gridDim.x * 256) v[e] = nan;
    if (blockIdx.x == 0 && threadIdx.x == 0) ctl->iters = -1;
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static unsigned sk_spin_limit() {
    if (const char* e = getenv("OTVAE_SK_SPIN_LIMIT")) {
        const long long q = atoll(e);
        if (q >= 0 && q <= 0x7fffffffLL) return (unsigned)q;
    }
    return SK_SPIN_LIMIT;
}

// workgroups of `kernel` (NTH threads, `lds` bytes of dynamic LDS) the device can hold at once: a kernel whose
// workgroups wait for each other must not be launched with more
template <typename K>
static int sk_resident_blocks(K kernel, int nth, size_t lds, int n_cu) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, nth, lds) != hipSuccess || per_cu <= 0) return 0;
    return per_cu * n_cu;
}

extern "C" int64_t otvae_sinkhorn_ws(int dtype, int nb, int N, int M) {
    if (nb <= 0 || N <= 0 || M <= 0 || dtype < 0 || dtype > 1) return -1;
    const size_t es = dtype ? 8 : 4;
    // Cr^T, log a, log b, |du|, |dv|, the two tagged potentials of the flag-in-data solver, the control block
    return (int64_t)(align256((size_t)nb * N * M * es) + 4 * align256((size_t)nb * (N > M ? N : M) * es) +
                     2 * align256((size_t)nb * (N > M ? N : M) * 8) + 256);
}

template <typename T>
static int sinkhorn_impl(const T* a, const T* b, const T* Cm, int nb, int N, int M, double reg, int max_iter, double threshold,
                         const T* pmax, int P, T* cmax_out, void* ws, T* pi, T* u, T* v, int32_t* iters_done, hipStream_t st) {
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
    const size_t ts = align256((size_t)nb * (N > M ? N : M) * 8);
    unsigned long long* tu = (unsigned long long*)w;
    w += ts;
    unsigned long long* tv = (unsigned long long*)w;
    w += ts;
    SkCtl* ctl = (SkCtl*)w;
    const bool track = threshold > 0.0;

    sk_init_mat<T><<<dim3(cdiv(M, 32), cdiv(N, 32), nb), 256, 0, st>>>(Cm, N, M, (T)(1.0 / reg), pmax, P, pi, crt, a, b, loga, logb, u, v,
                                                                       ctl, track ? 0 : max_iter, sk_spin_limit(), tu, tv, cmax_out);
    OTVAE_CHECK_LAUNCH("otvae_sinkhorn_log(init)");
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) n_cu = v;
        if (n_cu <= 0) n_cu = 1;
    }
    // Persistent solver: few LARGE workgroups (512 threads = 8 waves), up to 4 (fp32) / 1 (fp64) rows of each matrix per
    // wave in registers, the potential through LDS.  Eligible when a row fits 16 registers per lane (N, M <= 1024), the
    // rows fit <= 128 workgroups and the potentials fit the default 64 KiB of dynamic LDS.  OTVAE_SK_MULTILAUNCH=1 forces
    // one launch per half-iteration; OTVAE_SK_PERSISTENT=256 selects the first-generation shape (one 256-thread workgroup
    // per CU, one row per wave, no LDS staging), which measured 1.09 ms against 0.91 ms for the launches on the
    // 1024 x 1024 fp32 problem; OTVAE_SK_RPW=n fixes the rows per wave (experiments).
    const int rows = nb * (N > M ? N : M);
    constexpr int RPWMAX = sizeof(T) == 4 ? 4 : 1;  // registers: 2 x RPW x 16 (x 2 for fp64) of the 256 a 512-thread workgroup has
    const char* pers = getenv("OTVAE_SK_PERSISTENT");
    const bool legacy = pers && atoi(pers) == 256;
    int rpw = 1;
    // One row per wave while that needs <= 128 workgroups.  Measured (1024 x 1024 fp32, 50 iterations, ms per solve):
    // launches 0.87; 512 threads x 1 row/wave (128 workgroups) 0.57, x 2 (64) 0.70, x 4 (32) 1.09; 1024 threads x 1 (64)
    // 0.59; 256 threads x 1 (256) 0.78 -- the rows a wave walks serially cost more than the extra barrier arrivals.
    while (rpw < RPWMAX && cdiv(rows, 8 * rpw) > 128) rpw *= 2;
    if (const char* e = getenv("OTVAE_SK_RPW")) {
        const int want = atoi(e);
        if (want == 1 || want == 2 || (want == 4 && RPWMAX == 4)) rpw = want;
    }
    const int G = cdiv(rows, 8 * rpw);
    const size_t pot_bytes = (size_t)rows * sizeof(T);
    bool eligible = N <= 64 * SK_EPL && M <= 64 * SK_EPL && G <= 128 && pot_bytes <= 60 * 1024;
    if (eligible) {
        // the workgroups of a persistent solve wait for each other: all G must be resident at once
        int cap;
        if (sizeof(T) == 4 && !track && !getenv("OTVAE_SK_BARRIER"))
            cap = rpw == 1 ? sk_resident_blocks(sk_persistent_tagged<1, 512>, 512, pot_bytes, n_cu)
                  : rpw == 2 ? sk_resident_blocks(sk_persistent_tagged<2, 512>, 512, pot_bytes, n_cu)
                             : sk_resident_blocks(sk_persistent_tagged<4, 512>, 512, pot_bytes, n_cu);
        else
            cap = rpw == 1 ? sk_resident_blocks(sk_persistent<T, true, 1, 512, true>, 512, pot_bytes, n_cu)
                  : rpw == 2 ? sk_resident_blocks(sk_persistent<T, true, 2, 512, true>, 512, pot_bytes, n_cu)
                             : sk_resident_blocks(sk_persistent<T, true, RPWMAX, 512, true>, 512, pot_bytes, n_cu);
        if (G > cap) eligible = false;
    }
    if (!getenv("OTVAE_SK_MULTILAUNCH") && (legacy || (eligible && !(pers && atoi(pers) == 0)))) {
        if (legacy) {
            int G1 = imax(1, imin(n_cu, cdiv(rows, 4)));
            const bool cache = (nb * N <= 4 * G1) && (nb * M <= 4 * G1) && N <= 64 * SK_EPL && M <= 64 * SK_EPL;
            if (cache)
                sk_persistent<T, true, 1, 256, false><<<G1, 256, 0, st>>>(pi, crt, loga, logb, u, v, adu, adv, nb, N, M, max_iter, threshold, ctl);
            else
                sk_persistent<T, false, 1, 256, false><<<G1, 256, 0, st>>>(pi, crt, loga, logb, u, v, adu, adv, nb, N, M, max_iter, threshold, ctl);
        } else if (sizeof(T) == 4 && !track && !getenv("OTVAE_SK_BARRIER")) {
            // fixed iteration count in fp32 (the training configuration): flag-in-data exchange, no barrier
            float* fpi = reinterpret_cast<float*>(pi);
            const float *fcrt = reinterpret_cast<const float*>(crt), *fla = reinterpret_cast<const float*>(loga),
                        *flb = reinterpret_cast<const float*>(logb);
            float *fu = reinterpret_cast<float*>(u), *fv = reinterpret_cast<float*>(v);
            if (rpw == 1) sk_persistent_tagged<1, 512><<<G, 512, pot_bytes, st>>>(fpi, fcrt, fla, flb, fu, fv, tu, tv, nb, N, M, max_iter, ctl);
            else if (rpw == 2) sk_persistent_tagged<2, 512><<<G, 512, pot_bytes, st>>>(fpi, fcrt, fla, flb, fu, fv, tu, tv, nb, N, M, max_iter, ctl);
            else sk_persistent_tagged<4, 512><<<G, 512, pot_bytes, st>>>(fpi, fcrt, fla, flb, fu, fv, tu, tv, nb, N, M, max_iter, ctl);
        } else if (rpw == 1) {
            sk_persistent<T, true, 1, 512, true><<<G, 512, pot_bytes, st>>>(pi, crt, loga, logb, u, v, adu, adv, nb, N, M, max_iter, threshold, ctl);
        } else if (rpw == 2) {
            sk_persistent<T, true, 2, 512, true><<<G, 512, pot_bytes, st>>>(pi, crt, loga, logb, u, v, adu, adv, nb, N, M, max_iter, threshold, ctl);
        } else {
            sk_persistent<T, true, RPWMAX, 512, true><<<G, 512, pot_bytes, st>>>(pi, crt, loga, logb, u, v, adu, adv, nb, N, M, max_iter, threshold, ctl);
        }
        OTVAE_CHECK_LAUNCH("otvae_sinkhorn_log(persistent)");
        sk_finish<T><<<64, 256, 0, st>>>(ctl, pi, (size_t)nb * N * M, u, (size_t)nb * N, v, (size_t)nb * M, iters_done);
        OTVAE_CHECK_LAUNCH("otvae_sinkhorn_log(finish)");
        iters_done = nullptr;  // written by sk_finish
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
    OTVAE_REQUIRE(C && ws && pi && u && v, "otvae_sinkhorn_log: NULL argument");
    OTVAE_REQUIRE(nb > 0 && N > 0 && M > 0 && max_iter >= 0, "otvae_sinkhorn_log: bad sizes");
    OTVAE_REQUIRE(dtype == 0 || dtype == 1, "otvae_sinkhorn_log: dtype must be 0 (fp32) or 1 (fp64)");
    OTVAE_REQUIRE(reg > 0.0, "otvae_sinkhorn_log: reg must be positive");
    OTVAE_REQUIRE(pi != C, "otvae_sinkhorn_log: pi must not alias C");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0)
        return sinkhorn_impl<float>((const float*)a, (const float*)b, (const float*)C, nb, N, M, reg, max_iter, threshold, nullptr, 0,
                                    nullptr, ws, (float*)pi, (float*)u, (float*)v, iters_done, st);
    return sinkhorn_impl<double>((const double*)a, (const double*)b, (const double*)C, nb, N, M, reg, max_iter, threshold, nullptr, 0,
                                 nullptr, ws, (double*)pi, (double*)u, (double*)v, iters_done, st);
}

extern "C" int otvae_sinkhorn_log_normalized(int dtype, const void* a, const void* b, const void* C, const void* pmax, int P, int nb,
                                             int N, int M, double reg, int max_iter, double threshold, void* ws, void* pi, void* u,
                                             void* v, void* cmax, int32_t* iters_done, void* stream) {
    OTVAE_REQUIRE(C && pmax && P > 0 && ws && pi && u && v, "otvae_sinkhorn_log_normalized: NULL argument");
    OTVAE_REQUIRE(nb > 0 && N > 0 && M > 0 && max_iter >= 0, "otvae_sinkhorn_log_normalized: bad sizes");
    OTVAE_REQUIRE(dtype == 0 || dtype == 1, "otvae_sinkhorn_log_normalized: dtype must be 0 (fp32) or 1 (fp64)");
    OTVAE_REQUIRE(reg > 0.0, "otvae_sinkhorn_log_normalized: reg must be positive");
    OTVAE_REQUIRE(pi != C, "otvae_sinkhorn_log_normalized: pi must not alias C");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0)
        return sinkhorn_impl<float>((const float*)a, (const float*)b, (const float*)C, nb, N, M, reg, max_iter, threshold,
                                    (const float*)pmax, P, (float*)cmax, ws, (float*)pi, (float*)u, (float*)v, iters_done, st);
    return sinkhorn_impl<double>((const double*)a, (const double*)b, (const double*)C, nb, N, M, reg, max_iter, threshold,
                                 (const double*)pmax, P, (double*)cmax, ws, (double*)pi, (double*)u, (double*)v, iters_done, st);
}

// ---- sum_ij C*pi ---------------------------------------------------------------------------------------------
#define COST_PARTS 256
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
// cost[b * rep + r] = scale * sum of problem b's partials, r < rep (the prior hands the VAE one loss entry per sample)
template <typename T>
__global__ void ot_cost_final(const double* __restrict__ ws, int parts, int nb, T* __restrict__ cost, double scale, int rep_n) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nb * rep_n) return;
    const int b = e / rep_n;
    double s = 0.0;
    for (int p = 0; p < parts; ++p) s += ws[(size_t)b * COST_PARTS + p];
    cost[e] = (T)(scale * s);  // a timed-out solve left NaN in the plan (sk_finish): the cost is NaN
}

static int ot_cost_launch(int dtype, const void* C, const void* pi, int nb, int N, int M, double* ws, void* cost, double scale,
                          int rep_n, hipStream_t st) {
    const size_t total = (size_t)N * M;
    const int parts = imin(COST_PARTS, cdiv(total, 1024));
    if (dtype == 0) {
        ot_cost_partial<float><<<dim3(parts, nb), 256, 0, st>>>((const float*)C, (const float*)pi, total, ws);
        ot_cost_final<float><<<cdiv(nb * rep_n, 64), 64, 0, st>>>(ws, parts, nb, (float*)cost, scale, rep_n);
    } else {
        ot_cost_partial<double><<<dim3(parts, nb), 256, 0, st>>>((const double*)C, (const double*)pi, total, ws);
        ot_cost_final<double><<<cdiv(nb * rep_n, 64), 64, 0, st>>>(ws, parts, nb, (double*)cost, scale, rep_n);
    }
    return OTVAE_OK;
}

extern "C" int otvae_ot_cost(int dtype, const void* C, const void* pi, int nb, int N, int M, double* ws, void* cost,
                             void* stream) {
    OTVAE_REQUIRE(C && pi && ws && cost && nb > 0 && N > 0 && M > 0, "otvae_ot_cost: bad argument");
    OTVAE_REQUIRE(dtype == 0 || dtype == 1, "otvae_ot_cost: dtype must be 0 or 1");
    ot_cost_launch(dtype, C, pi, nb, N, M, ws, cost, 1.0, 1, (hipStream_t)stream);
    OTVAE_CHECK_LAUNCH("otvae_ot_cost");
    return OTVAE_OK;
}

// ---- pairwise squared euclidean cost: |x_i|^2 + |y_j|^2 - 2 x_i.y_j (same expansion as ot/w2_utils.py:121-125) --
template <typename T>
__global__ __launch_bounds__(256) void sqdist_kernel(const T* __restrict__ x, const T* __restrict__ y, int N, int M, int D,
                                                     T* __restrict__ Cm, T* __restrict__ pmax) {
    __shared__ T xs[16][33], ys[16][33];
    __shared__ T s_mx[4];
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
    const T val = xx + yy - (T)2 * dot;
    if (i < N && j < M) Cm[((size_t)b * N + i) * M + j] = val;
    if (pmax) {  // this tile's maximum (the per-matrix maximum is the maximum of these: sk_init_mat)
        T mx = wave_max((i < N && j < M) ? val : MathT<T>::ninf());
        if ((threadIdx.x & 63) == 0) s_mx[threadIdx.x >> 6] = mx;
        __syncthreads();
        if (threadIdx.x == 0) {
            const T m01 = s_mx[0] > s_mx[1] ? s_mx[0] : s_mx[1], m23 = s_mx[2] > s_mx[3] ? s_mx[2] : s_mx[3];
            pmax[((size_t)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = m01 > m23 ? m01 : m23;
        }
    }
}

// fp32 on the matrix cores: a workgroup owns a 64 x 64 tile of C (4 waves, 32 x 32 each = 2 x 2 MFMA 16x16x4 tiles), K = D
// walked in chunks of 32 staged through LDS with coalesced float4 rows.  A lane reads its A / B operands of four
// consecutive MFMA steps as ONE ds_read_b128 along k (the k index an MFMA step sees is 4*(lane>>4) + step for A and B
// alike, so the products pair up correctly whatever the order of k).  The row norms come from the same LDS tiles.
#define SQ_KC 32
#define SQ_LD 36  // floats per LDS row: 16-byte aligned rows, 4 banks apart
__global__ __launch_bounds__(256) void sqdist_mfma_kernel(const float* __restrict__ x, const float* __restrict__ y, int N, int M, int D,
                                                          float* __restrict__ Cm, float* __restrict__ pmax) {
    __shared__ __align__(16) float xs[64 * SQ_LD], ys[64 * SQ_LD];
    __shared__ float nx[64], ny[64], s_mx[4];
    const int b = blockIdx.z, i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, wi = w >> 1, wj = w & 1;
    const float* xb = x + (size_t)b * N * D;
    const float* yb = y + (size_t)b * M * D;
    const bool vec = (D & 3) == 0;
    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[a][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float nrm = 0.f;  // thread t: half (t & 1) of row (t >> 1) of the 128 staged rows (0..63 of x, 64..127 of y)
    auto load4 = [&](const float* base, int row, int rows, int d) -> float4 {
        if (row >= rows) return make_float4(0.f, 0.f, 0.f, 0.f);
        const float* p = base + (size_t)row * D + d;
        if (vec && d + 3 < D) return *reinterpret_cast<const float4*>(p);
        return make_float4(d < D ? p[0] : 0.f, d + 1 < D ? p[1] : 0.f, d + 2 < D ? p[2] : 0.f, d + 3 < D ? p[3] : 0.f);
    };
    const int sr = t >> 3, sc = (t & 7) * 4;  // staging: rows sr, sr + 32; 4 floats at column sc
    float4 px[2], py[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        px[u] = load4(xb, i0 + sr + 32 * u, N, sc);
        py[u] = load4(yb, j0 + sr + 32 * u, M, sc);
    }
    for (int d0 = 0; d0 < D; d0 += SQ_KC) {
        __syncthreads();  // the previous chunk's readers are done
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            *reinterpret_cast<float4*>(&xs[(sr + 32 * u) * SQ_LD + sc]) = px[u];
            *reinterpret_cast<float4*>(&ys[(sr + 32 * u) * SQ_LD + sc]) = py[u];
        }
        __syncthreads();
        if (d0 + SQ_KC < D) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                px[u] = load4(xb, i0 + sr + 32 * u, N, d0 + SQ_KC + sc);
                py[u] = load4(yb, j0 + sr + 32 * u, M, d0 + SQ_KC + sc);
            }
        }
        {
            const int rr = t >> 1;
            const float* src = (rr < 64 ? xs + rr * SQ_LD : ys + (rr - 64) * SQ_LD) + (t & 1) * 16;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 f = *reinterpret_cast<const float4*>(src + 4 * q);
                nrm += (f.x * f.x + f.y * f.y) + (f.z * f.z + f.w * f.w);
            }
        }
#pragma unroll
        for (int kk = 0; kk < SQ_KC; kk += 16) {
            float4 av[2], bv[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                av[a] = *reinterpret_cast<const float4*>(&xs[(wi * 32 + a * 16 + (lane & 15)) * SQ_LD + kk + 4 * (lane >> 4)]);
                bv[a] = *reinterpret_cast<const float4*>(&ys[(wj * 32 + a * 16 + (lane & 15)) * SQ_LD + kk + 4 * (lane >> 4)]);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    acc[a][c] = mfma16(av[a].x, bv[c].x, acc[a][c]);
                    acc[a][c] = mfma16(av[a].y, bv[c].y, acc[a][c]);
                    acc[a][c] = mfma16(av[a].z, bv[c].z, acc[a][c]);
                    acc[a][c] = mfma16(av[a].w, bv[c].w, acc[a][c]);
                }
        }
    }
    nrm += __shfl_xor(nrm, 1, 64);
    if ((t & 1) == 0) {
        if ((t >> 1) < 64) nx[t >> 1] = nrm;
        else ny[(t >> 1) - 64] = nrm;
    }
    __syncthreads();
    float mx = -INFINITY;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int li = wi * 32 + a * 16 + (lane >> 4) * 4 + r, lj = wj * 32 + c * 16 + (lane & 15);
                const float val = nx[li] + ny[lj] - 2.f * acc[a][c][r];
                if (i0 + li < N && j0 + lj < M) {
                    Cm[((size_t)b * N + i0 + li) * M + j0 + lj] = val;
                    mx = val > mx ? val : mx;
                }
            }
    if (pmax) {
        mx = wave_max(mx);
        if (lane == 0) s_mx[w] = mx;
        __syncthreads();
        if (t == 0) pmax[((size_t)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3]));
    }
}

static int sqdist_tile(int dtype) { return dtype == 0 ? 64 : 16; }

extern "C" int otvae_sqdist_max_parts(int dtype, int N, int M) {
    if (N <= 0 || M <= 0 || dtype < 0 || dtype > 1) return -1;
    const int t = sqdist_tile(dtype);
    return cdiv(N, t) * cdiv(M, t);
}

static int sqdist_launch(int dtype, const void* x, const void* y, int nb, int N, int M, int D, void* C, void* pmax, hipStream_t st) {
    const int t = sqdist_tile(dtype);
    dim3 grid(cdiv(M, t), cdiv(N, t), nb);
    if (dtype == 0)
        sqdist_mfma_kernel<<<grid, 256, 0, st>>>((const float*)x, (const float*)y, N, M, D, (float*)C, (float*)pmax);
    else
        sqdist_kernel<double><<<grid, 256, 0, st>>>((const double*)x, (const double*)y, N, M, D, (double*)C, (double*)pmax);
    return OTVAE_OK;
}

extern "C" int otvae_sqdist(int dtype, const void* x, const void* y, int nb, int N, int M, int D, void* C, void* stream) {
    OTVAE_REQUIRE(x && y && C && nb > 0 && N > 0 && M > 0 && D > 0, "otvae_sqdist: bad argument");
    OTVAE_REQUIRE(dtype == 0 || dtype == 1, "otvae_sqdist: dtype must be 0 or 1");
    OTVAE_REQUIRE(dtype == 1 || ((uintptr_t)x % 16 == 0 && (uintptr_t)y % 16 == 0), "otvae_sqdist: fp32 inputs must be 16-byte aligned");
    sqdist_launch(dtype, x, y, nb, N, M, D, C, nullptr, (hipStream_t)stream);
    OTVAE_CHECK_LAUNCH("otvae_sqdist");
    return OTVAE_OK;
}

extern "C" int otvae_sqdist_max(int dtype, const void* x, const void* y, int nb, int N, int M, int D, void* C, void* pmax,
                                void* stream) {
    OTVAE_REQUIRE(x && y && C && pmax && nb > 0 && N > 0 && M > 0 && D > 0, "otvae_sqdist_max: bad argument");
    OTVAE_REQUIRE(dtype == 0 || dtype == 1, "otvae_sqdist_max: dtype must be 0 or 1");
    OTVAE_REQUIRE(dtype == 1 || ((uintptr_t)x % 16 == 0 && (uintptr_t)y % 16 == 0), "otvae_sqdist_max: fp32 inputs must be 16-byte aligned");
    sqdist_launch(dtype, x, y, nb, N, M, D, C, pmax, (hipStream_t)stream);
    OTVAE_CHECK_LAUNCH("otvae_sqdist_max");
    return OTVAE_OK;
}


// ---- gradient of <C, pi> with C_ij = |z_i - y_j|^2 and the plan held fixed (the envelope argument of the OT prior):
// gz[i][d] = 2 g sum_j pi_ij (z_id - y_jd) = 2 g (z_id sum_j pi_ij - sum_j pi_ij y_jd): a [N x M] x [M x D] product plus the
// plan's row sums.  32 x 32 output tiles, the plan and y staged through LDS in 32-wide slices of j, 2 x 2 outputs per lane.
template <typename T>
__global__ __launch_bounds__(256) void ot_cost_grad_kernel(const T* __restrict__ z, const T* __restrict__ y, const T* __restrict__ pi,
                                                           const T* __restrict__ g, int ng, T scale, const T* __restrict__ gadd,
                                                           int N, int M, int D, T* __restrict__ gz) {
    __shared__ T ps[32][33], ys[32][33];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int i0 = blockIdx.y * 32, d0 = blockIdx.x * 32;
    T acc[2][2] = {{(T)0, (T)0}, {(T)0, (T)0}}, rs[2] = {(T)0, (T)0};
    // the next slice's elements travel in registers while the current one is multiplied (4 + 4 per lane)
    T pn[4], yn[4];
    auto fetch = [&](int j0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = threadIdx.x + 256 * u, r = e >> 5, c = e & 31;
            pn[u] = (i0 + r < N && j0 + c < M) ? pi[(size_t)(i0 + r) * M + j0 + c] : (T)0;   // [i][j]
            yn[u] = (j0 + r < M && d0 + c < D) ? y[(size_t)(j0 + r) * D + d0 + c] : (T)0;    // [j][d]
        }
    };
    fetch(0);
    for (int j0 = 0; j0 < M; j0 += 32) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = threadIdx.x + 256 * u, r = e >> 5, c = e & 31;
            ps[r][c] = pn[u];
            ys[r][c] = yn[u];
        }
        __syncthreads();
        if (j0 + 32 < M) fetch(j0 + 32);
#pragma unroll 8
        for (int j = 0; j < 32; ++j) {
            const T p0 = ps[ty][j], p1 = ps[ty + 16][j], y0 = ys[j][tx], y1 = ys[j][tx + 16];
            rs[0] += p0;
            rs[1] += p1;
            acc[0][0] += p0 * y0;
            acc[0][1] += p0 * y1;
            acc[1][0] += p1 * y0;
            acc[1][1] += p1 * y1;
        }
        __syncthreads();
    }
    T gs = (T)0;  // the upstream gradient of every replica of the cost, summed in index order by every thread alike
    for (int q = 0; q < ng; ++q) gs += g[q];
    const T two_g = (T)2 * scale * gs;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int i = i0 + ty + 16 * a, d = d0 + tx + 16 * c;
            if (i < N && d < D) {
                const T val = two_g * (rs[a] * z[(size_t)i * D + d] - acc[a][c]);
                gz[(size_t)i * D + d] = gadd ? gadd[(size_t)i * D + d] + val : val;
            }
        }
}

// fp32 on the matrix cores.  A workgroup owns one 16 x 16 tile of gz; its 4 waves split K = M into 4 contiguous ranges and
// each accumulates its own MFMA 16x16x4 tile straight from global memory (the plan and y are L2 resident: 4 MiB + 0.5 MiB):
// per 16 k one 16-byte load of the plan (4 consecutive k of a row: the A operands of four MFMA steps) and four dword loads
// of y, four such groups in flight per wave, no barrier inside the loop.  The plan's row sums fall out of the same A
// values.  The 4 partial tiles are combined through LDS in wave order: a fixed summation order, bit-identical from run to run.
__global__ __launch_bounds__(256) void ot_cost_grad_mfma_kernel(const float* __restrict__ z, const float* __restrict__ y,
                                                                const float* __restrict__ pi, const float* __restrict__ g, int ng,
                                                                float scale, const float* __restrict__ gadd, int N, int M, int D,
                                                                float* __restrict__ gz) {
    __shared__ float part[3][5][64];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int i0 = blockIdx.y * 16, d0 = blockIdx.x * 16;
    const int row = i0 + (lane & 15), col = d0 + (lane & 15), kq = 4 * (lane >> 4);
    const bool vecm = (M & 3) == 0;
    const int span = ((M + 63) / 64) * 16;  // k per wave, a multiple of 16
    const int k_lo = w * span, k_hi = min(M, k_lo + span);
    const float* prow = pi + (size_t)min(row, N - 1) * M;
    const bool row_ok = row < N, col_ok = col < D;
    const float* ycol = y + min(col, D - 1);
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    float rs = 0.f;
    auto loadA = [&](int k) -> float4 {
        const int kk = k + kq;
        if (!row_ok || kk >= k_hi) return make_float4(0.f, 0.f, 0.f, 0.f);
        if (vecm && kk + 3 < k_hi) return *reinterpret_cast<const float4*>(prow + kk);
        return make_float4(prow[kk], kk + 1 < k_hi ? prow[kk + 1] : 0.f, kk + 2 < k_hi ? prow[kk + 2] : 0.f,
                           kk + 3 < k_hi ? prow[kk + 3] : 0.f);
    };
    auto loadB = [&](int k, int s_) -> float {
        const int kk = k + kq + s_;
        return (col_ok && kk < k_hi) ? ycol[(size_t)kk * D] : 0.f;
    };
    for (int k = k_lo; k < k_hi; k += 64) {
        float4 av[4];
        float bv[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            av[u] = loadA(k + 16 * u);
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) bv[u][s_] = loadB(k + 16 * u, s_);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            rs += (av[u].x + av[u].y) + (av[u].z + av[u].w);
            acc = mfma16(av[u].x, bv[u][0], acc);
            acc = mfma16(av[u].y, bv[u][1], acc);
            acc = mfma16(av[u].z, bv[u][2], acc);
            acc = mfma16(av[u].w, bv[u][3], acc);
        }
    }
    rs += __shfl_xor(rs, 16, 64);
    rs += __shfl_xor(rs, 32, 64);  // every lane: this wave's share of the row sum of row (lane & 15)
    if (w > 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) part[w - 1][r][lane] = acc[r];
        part[w - 1][4][lane] = rs;
    }
    __syncthreads();
    if (w != 0) return;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] += part[q][r][lane];
        rs += part[q][4][lane];
    }
    float gs = 0.f;  // sum of the upstream gradients of the cost's replicas: lanes take every 64th, fixed shuffle tree
    for (int q = lane; q < ng; q += 64) gs += g[q];
    gs = wave_sum(gs);
    const float two_g = 2.f * scale * gs;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int li = (lane >> 4) * 4 + r;
        const float rsi = __shfl(rs, li, 64);
        const int i = i0 + li;
        if (i < N && col_ok) {
            const size_t e = (size_t)i * D + col;
            const float v = two_g * (rsi * z[e] - acc[r]);
            gz[e] = gadd ? gadd[e] + v : v;
        }
    }
}

extern "C" int otvae_ot_cost_grad(int dtype, const void* z, const void* y, const void* pi, const void* g, int ng, double scale,
                                  const void* gadd, int N, int M, int D, void* gz, void* stream) {
    OTVAE_REQUIRE(z && y && pi && g && gz && ng > 0 && N > 0 && M > 0 && D > 0, "otvae_ot_cost_grad: bad argument");
    OTVAE_REQUIRE(dtype == 0 || dtype == 1, "otvae_ot_cost_grad: dtype must be 0 or 1");
    const dim3 grid(cdiv(D, 32), cdiv(N, 32));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0 && (uintptr_t)pi % 16 == 0 && !getenv("OTVAE_OT_GRAD_VALU"))
        ot_cost_grad_mfma_kernel<<<dim3(cdiv(D, 16), cdiv(N, 16)), 256, 0, st>>>((const float*)z, (const float*)y, (const float*)pi,
                                                                                 (const float*)g, ng, (float)scale, (const float*)gadd, N,
                                                                                 M, D, (float*)gz);
    else if (dtype == 0)
        ot_cost_grad_kernel<float><<<grid, 256, 0, st>>>((const float*)z, (const float*)y, (const float*)pi, (const float*)g, ng,
                                                         (float)scale, (const float*)gadd, N, M, D, (float*)gz);
    else
        ot_cost_grad_kernel<double><<<grid, 256, 0, st>>>((const double*)z, (const double*)y, (const double*)pi, (const double*)g, ng,
                                                          scale, (const double*)gadd, N, M, D, (double*)gz);
    OTVAE_CHECK_LAUNCH("otvae_ot_cost_grad");
    return OTVAE_OK;
}

// ---- the minibatch-OT prior's forward in one call: C = |z_i - y_j|^2 (+ tile maxima), Cr = -C / (reg max C), the solve with
// uniform marginals, cost = sum C * pi.  5 launches (cost tiles, init, solve, finish, read-out x 2), nothing in between.
extern "C" int64_t otvae_sinkhorn_prior_ws(int dtype, int N, int M) {
    const int64_t base = otvae_sinkhorn_ws(dtype, 1, N, M);
    if (base < 0) return -1;
    const size_t es = dtype ? 8 : 4;
    return base + (int64_t)align256((size_t)otvae_sqdist_max_parts(dtype, N, M) * es) + (int64_t)align256(COST_PARTS * sizeof(double));
}

extern "C" int otvae_sinkhorn_prior_fwd(int dtype, const void* z, const void* y, int N, int M, int D, double reg, int max_iter,
                                        double threshold, double loss_scale, int cost_rep, void* ws, void* C, void* pi, void* u,
                                        void* v, void* cost, void* cmax, int32_t* iters_done, void* stream) {
    OTVAE_REQUIRE(z && y && ws && C && pi && u && v && cost, "otvae_sinkhorn_prior_fwd: NULL argument");
    OTVAE_REQUIRE(N > 0 && M > 0 && D > 0 && max_iter >= 0 && reg > 0.0 && cost_rep > 0, "otvae_sinkhorn_prior_fwd: bad sizes");
    OTVAE_REQUIRE(dtype == 0 || dtype == 1, "otvae_sinkhorn_prior_fwd: dtype must be 0 (fp32) or 1 (fp64)");
    OTVAE_REQUIRE(dtype == 1 || ((uintptr_t)z % 16 == 0 && (uintptr_t)y % 16 == 0), "otvae_sinkhorn_prior_fwd: fp32 inputs must be 16-byte aligned");
    const size_t es = dtype ? 8 : 4;
    const int P = otvae_sqdist_max_parts(dtype, N, M);
    char* w = (char*)ws;
    void* pmax = w;
    w += align256((size_t)P * es);
    double* cws = (double*)w;
    w += align256(COST_PARTS * sizeof(double));
    hipStream_t st = (hipStream_t)stream;
    sqdist_launch(dtype, z, y, 1, N, M, D, C, pmax, st);
    OTVAE_CHECK_LAUNCH("otvae_sinkhorn_prior_fwd(cost matrix)");
    int rc;
    if (dtype == 0)
        rc = sinkhorn_impl<float>(nullptr, nullptr, (const float*)C, 1, N, M, reg, max_iter, threshold, (const float*)pmax, P, (float*)cmax, w,
                                  (float*)pi, (float*)u, (float*)v, iters_done, st);
    else
        rc = sinkhorn_impl<double>(nullptr, nullptr, (const double*)C, 1, N, M, reg, max_iter, threshold, (const double*)pmax, P,
                                   (double*)cmax, w, (double*)pi, (double*)u, (double*)v, iters_done, st);
    if (rc) return rc;
    ot_cost_launch(dtype, C, pi, 1, N, M, cws, cost, loss_scale, cost_rep, st);
    OTVAE_CHECK_LAUNCH("otvae_sinkhorn_prior_fwd(read-out)");
    return OTVAE_OK;
}
