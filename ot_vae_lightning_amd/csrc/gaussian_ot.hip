// fp64 Gaussian optimal-transport arithmetic:
//   * streaming sufficient statistics (n, sum x, sum x x^T) of GaussianModel.update/_stats
//     (reference ot/distribution_models/gaussian_model.py:99-108,144-157; utils/__init__.py:204-206)
//   * mean_cov (ot/matrix_utils.py:145-158)
//   * eigh-based matrix functions sqrtm / invsqrtm / min_eig / make_psd (ot/matrix_utils.py:37-142) through a
//     parallel cyclic Jacobi eigensolver that keeps the matrix in LDS (one workgroup per matrix, D <= 128)
//   * the small dense fp64 products, trace and affine map of w2_gaussian / compute_transport_operators /
//     apply_transport (ot/w2_utils.py:40-80,756-769,517-520)
//   * CodebookModel nearest-atom assignment (ot/distribution_models/codebook_model.py:150-160)
// Everything is reduced in a fixed order (no atomics): results are run-to-run identical.
#include "common.h"

// ================================================================================================ statistics
// Augmented sample x' = [x, 1] (dimension D+1): S' = sum_b x' x'^T holds sum x x^T, sum x (last column) and n.
#define GS_TILE 16
#define GS_KSPLIT_MAX 16

static int gs_ksplit(int B) { return imax(1, imin(GS_KSPLIT_MAX, B / 64)); }

extern "C" int64_t otvae_gauss_stats_ws(int nb, int B, int D, int diag) {
    if (nb <= 0 || B <= 0 || D <= 0) return -1;
    if (diag) return 8;
    return (int64_t)nb * gs_ksplit(B) * (D + 1) * (D + 1) * (int64_t)sizeof(double);
}

template <typename TIN>
__global__ __launch_bounds__(256) void gs_partial_kernel(const TIN* __restrict__ x, int B, int D, int ksplit,
                                                         double* __restrict__ ws) {
    __shared__ double xi[GS_TILE][GS_TILE + 1], xj[GS_TILE][GS_TILE + 1];
    const int D1 = D + 1;
    const int nt = (D1 + GS_TILE - 1) / GS_TILE;
    const int ti = blockIdx.x / nt, tj = blockIdx.x % nt;
    const int ks = blockIdx.y, b = blockIdx.z;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int rows_per = (B + ksplit - 1) / ksplit;
    const int r0 = ks * rows_per, r1 = min(B, r0 + rows_per);
    const TIN* xb = x + (size_t)b * B * D;
    double acc = 0.0;
    for (int r = r0; r < r1; r += GS_TILE) {
        // stage 16 rows x 16 columns of each operand (column D is the constant 1)
        const int row = r + ty;
        const int ci = ti * GS_TILE + tx, cj = tj * GS_TILE + tx;
        xi[ty][tx] = (row < r1 && ci < D1) ? (ci < D ? (double)xb[(size_t)row * D + ci] : 1.0) : 0.0;
        xj[ty][tx] = (row < r1 && cj < D1) ? (cj < D ? (double)xb[(size_t)row * D + cj] : 1.0) : 0.0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < GS_TILE; ++k) acc = fma(xi[k][ty], xj[k][tx], acc);
        __syncthreads();
    }
    const int i = ti * GS_TILE + ty, j = tj * GS_TILE + tx;
    if (i < D1 && j < D1) ws[(((size_t)b * ksplit + ks) * D1 + i) * D1 + j] = acc;
}

__global__ __launch_bounds__(256) void gs_final_kernel(const double* __restrict__ ws, int D, int ksplit, int accumulate,
                                                       double decay, double* __restrict__ n_obs, double* __restrict__ sum_x,
                                                       double* __restrict__ sum_xx) {
    const int D1 = D + 1;
    const int b = blockIdx.y;
    const size_t total = (size_t)D1 * D1;
    for (size_t e = blockIdx.x * (size_t)256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int i = e / D1, j = e - (size_t)i * D1;
        double s = 0.0;
        for (int k = 0; k < ksplit; ++k) s += ws[((size_t)b * ksplit + k) * total + e];
        double* dst = nullptr;
        if (i < D && j < D) dst = sum_xx + ((size_t)b * D + i) * D + j;
        else if (i < D && j == D) dst = sum_x + (size_t)b * D + i;
        else if (i == D && j == D) dst = n_obs + b;
        if (dst) {
            if (!accumulate) *dst = s;
            else if (decay < 0.0) *dst = *dst + s;
            else *dst = *dst * decay + s * (1.0 - decay);
        }
    }
}

template <typename TIN>
__global__ __launch_bounds__(256) void gs_diag_kernel(const TIN* __restrict__ x, int B, int D, int accumulate, double decay,
                                                      double* __restrict__ n_obs, double* __restrict__ sum_x,
                                                      double* __restrict__ sum_xx) {
    const int b = blockIdx.y;
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d < D) {
        const TIN* xb = x + (size_t)b * B * D;
        double s = 0.0, q = 0.0;
        for (int r = 0; r < B; ++r) {
            const double v = (double)xb[(size_t)r * D + d];
            s += v;
            q += v * v;
        }
        double* ds = sum_x + (size_t)b * D + d;
        double* dq = sum_xx + (size_t)b * D + d;
        if (!accumulate) { *ds = s; *dq = q; }
        else if (decay < 0.0) { *ds += s; *dq += q; }
        else { *ds = *ds * decay + s * (1.0 - decay); *dq = *dq * decay + q * (1.0 - decay); }
    }
    if (d == 0) {
        const double n = (double)B;
        if (!accumulate) n_obs[b] = n;
        else if (decay < 0.0) n_obs[b] += n;
        else n_obs[b] = n_obs[b] * decay + n * (1.0 - decay);
    }
}

extern "C" int otvae_gauss_stats(int in_dtype, const void* samples, int nb, int B, int D, int diag, int accumulate,
                                 double decay, double* ws, double* n_obs, double* sum_x, double* sum_xx, void* stream) {
    OTVAE_REQUIRE(samples && n_obs && sum_x && sum_xx && nb > 0 && B > 0 && D > 0, "otvae_gauss_stats: bad argument");
    OTVAE_REQUIRE(in_dtype == 0 || in_dtype == 1, "otvae_gauss_stats: in_dtype must be 0 or 1");
    hipStream_t st = (hipStream_t)stream;
    if (diag) {
        dim3 grid(cdiv(D, 256), nb);
        if (in_dtype == 0) gs_diag_kernel<float><<<grid, 256, 0, st>>>((const float*)samples, B, D, accumulate, decay, n_obs, sum_x, sum_xx);
        else gs_diag_kernel<double><<<grid, 256, 0, st>>>((const double*)samples, B, D, accumulate, decay, n_obs, sum_x, sum_xx);
        OTVAE_CHECK_LAUNCH("otvae_gauss_stats(diag)");
        return OTVAE_OK;
    }
    OTVAE_REQUIRE(ws, "otvae_gauss_stats: workspace missing");
    const int ks = gs_ksplit(B);
    const int nt = cdiv(D + 1, GS_TILE);
    dim3 grid(nt * nt, ks, nb);
    if (in_dtype == 0) gs_partial_kernel<float><<<grid, 256, 0, st>>>((const float*)samples, B, D, ks, ws);
    else gs_partial_kernel<double><<<grid, 256, 0, st>>>((const double*)samples, B, D, ks, ws);
    OTVAE_CHECK_LAUNCH("otvae_gauss_stats(partial)");
    gs_final_kernel<<<dim3(imin(cdiv((size_t)(D + 1) * (D + 1), 256), 256), nb), 256, 0, st>>>(ws, D, ks, accumulate, decay, n_obs,
                                                                                           sum_x, sum_xx);
    OTVAE_CHECK_LAUNCH("otvae_gauss_stats(final)");
    return OTVAE_OK;
}

__global__ __launch_bounds__(256) void mean_cov_kernel(const double* __restrict__ n_obs, const double* __restrict__ sum_x,
                                                       const double* __restrict__ sum_xx, int D, int diag,
                                                       double* __restrict__ mean, double* __restrict__ cov) {
    const int b = blockIdx.y;
    const double n = n_obs[b];
    const size_t total = diag ? (size_t)D : (size_t)D * D;
    for (size_t e = blockIdx.x * (size_t)256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        if (diag) {
            const double m = sum_x[(size_t)b * D + e] / n;
            mean[(size_t)b * D + e] = m;
            cov[(size_t)b * D + e] = sum_xx[(size_t)b * D + e] / n - m * m;
        } else {
            const int i = e / D, j = e - (size_t)i * D;
            const double mi = sum_x[(size_t)b * D + i] / n, mj = sum_x[(size_t)b * D + j] / n;
            cov[(size_t)b * D * D + e] = sum_xx[(size_t)b * D * D + e] / n - mi * mj;
            if (j == 0) mean[(size_t)b * D + i] = mi;
        }
    }
}

extern "C" int otvae_mean_cov(const double* n_obs, const double* sum_x, const double* sum_xx, int nb, int D, int diag,
                              double* mean, double* cov, void* stream) {
    OTVAE_REQUIRE(n_obs && sum_x && sum_xx && mean && cov && nb > 0 && D > 0, "otvae_mean_cov: bad argument");
    const size_t total = diag ? (size_t)D : (size_t)D * D;
    mean_cov_kernel<<<dim3(imin(cdiv(total, 256), 256), nb), 256, 0, (hipStream_t)stream>>>(n_obs, sum_x, sum_xx, D, diag, mean, cov);
    OTVAE_CHECK_LAUNCH("otvae_mean_cov");
    return OTVAE_OK;
}

// ================================================================================================ eigensolver
// Parallel cyclic Jacobi.  A (symmetrised from its lower triangle) lives in LDS with row stride D+1; the accumulated
// rotations V^T live in the global workspace (rows = eigenvectors).  Each of the De-1 steps of a sweep rotates De/2
// disjoint (p,q) pairs at once (round-robin tournament): rows phase | barrier | columns phase | barrier.
#define EIGH_THREADS 512
#define EIGH_MAX_D 128
#define EIGH_MAX_SWEEPS 24

#define EIGB 16                 // block size of the block-Jacobi driver (D > EIGH_MAX_D): 32 x 32 sub-problems
#define EIGH_BLOCK_MAX_D 2048
#define EIGH_BLOCK_SWEEPS 14

static int eigb_dp(int D) { return (D + 2 * EIGB - 1) / (2 * EIGB) * (2 * EIGB); }

extern "C" int64_t otvae_eigh_onesided_ws(int nb, int D);
int eigh_onesided(const double* A, int nb, int D, int fn, double* out, double* eigvals, void* ws, hipStream_t st, const double* Vinit,
                  double* g0, const int* warm);  // eigh_onesided.hip
extern "C" int64_t otvae_eigh_block_onesided_ws(int nb, int D);
int eigh_block_onesided(const double* A, int nb, int D, int fn, double* out, double* eigvals, void* ws, hipStream_t st);

extern "C" int64_t otvae_eigh_ws(int nb, int D) {
    if (nb <= 0 || D <= 0) return -1;
    if (D <= EIGH_MAX_D) {  // whichever of the two small-matrix solvers runs (OTVAE_EIGH_TWOSIDED=1 selects the first generation)
        const int64_t two_sided = (int64_t)nb * D * D * (int64_t)sizeof(double), one_sided = otvae_eigh_onesided_ws(nb, D);
        return two_sided > one_sided ? two_sided : one_sided;
    }
    // block driver (one matrix at a time): Aw[Dp][Dp], Vt[Dp][Dp], S and U [Dp/32][32][32], sub-eigenvalues, dense
    // D x D copies for the f(A) product, convergence flag -- or the one-sided multi-workgroup solver's share (D <= 1024)
    const int64_t Dp = eigb_dp(D), np = Dp / (2 * EIGB);
    const int64_t two_sided = (2 * Dp * Dp + 2 * np * 4 * EIGB * EIGB + np * 2 * EIGB + 2 * (int64_t)D * D + 8) * (int64_t)sizeof(double);
    const int64_t one_sided = D <= 1024 ? otvae_eigh_block_onesided_ws(nb, D) : 0;
    return two_sided > one_sided ? two_sided : one_sided;
}

__global__ __launch_bounds__(EIGH_THREADS) void eigh_kernel(const double* __restrict__ Ain, int D, int fn,
                                                            double* __restrict__ out, double* __restrict__ eigvals,
                                                            double* __restrict__ vt_ws, const int* __restrict__ skip) {
    extern __shared__ __align__(16) double lds[];
    if (skip != nullptr && *skip) return;  // block-Jacobi driver: the big matrix has already converged
    const int LD = D + 1;
    const int De = (D + 1) & ~1;  // even number of players; index D (if De > D) is a dummy
    const int half = De / 2;
    double* A = lds;                          // [D][LD]
    double* cs = A + (size_t)D * LD;          // [half]
    double* sn = cs + half;                   // [half]
    int* pp = (int*)(sn + half);              // [half]
    int* qq = pp + half;                      // [half]
    double* red = (double*)(qq + half);       // [16]  (2*half ints keep 8-byte alignment)
    const int tid = threadIdx.x;
    const int nwave = EIGH_THREADS / 64;
    const size_t boff = (size_t)blockIdx.x * D * D;
    double* VT = vt_ws + boff;
    const double* Ab = Ain + boff;

    for (int e = tid; e < D * D; e += EIGH_THREADS) {
        const int i = e / D, j = e - i * D;
        A[i * LD + j] = (i >= j) ? Ab[(size_t)i * D + j] : Ab[(size_t)j * D + i];
        VT[e] = (i == j) ? 1.0 : 0.0;
    }
    __syncthreads();

    for (int sweep = 0; sweep < EIGH_MAX_SWEEPS; ++sweep) {
        // convergence: off-diagonal mass vs total
        double off = 0.0, tot = 0.0;
        for (int e = tid; e < D * D; e += EIGH_THREADS) {
            const int i = e / D, j = e - i * D;
            const double v = A[i * LD + j];
            tot += v * v;
            if (i != j) off += v * v;
        }
        off = wave_sum(off);
        tot = wave_sum(tot);
        if ((tid & 63) == 0) {
            red[tid >> 6] = off;
            red[8 + (tid >> 6)] = tot;
        }
        __syncthreads();
        double offs = 0.0, tots = 0.0;
        for (int w = 0; w < nwave; ++w) {
            offs += red[w];
            tots += red[8 + w];
        }
        __syncthreads();
        if (offs <= 1e-30 * tots || tots == 0.0) break;  // uniform

        for (int step = 0; step < De - 1; ++step) {
            // rotation parameters
            if (tid < half) {
                int p, q;
                if (tid == 0) {
                    p = De - 1;
                    q = step % (De - 1);
                } else {
                    p = (step + tid) % (De - 1);
                    q = (step - tid + (De - 1)) % (De - 1);
                }
                if (p > q) { const int t = p; p = q; q = t; }
                double c = 1.0, s = 0.0;
                if (q < D) {
                    const double apq = A[p * LD + q];
                    if (apq != 0.0) {
                        const double app = A[p * LD + p], aqq = A[q * LD + q];
                        const double tau = (aqq - app) / (2.0 * apq);
                        const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                        c = 1.0 / sqrt(1.0 + t * t);
                        s = t * c;
                    }
                } else {
                    q = -1;  // dummy pairing: nothing to do
                }
                cs[tid] = c;
                sn[tid] = s;
                pp[tid] = p;
                qq[tid] = q;
            }
            __syncthreads();
            // rows: A <- J^T A ; VT <- J^T VT
            for (int e = tid; e < half * D; e += EIGH_THREADS) {
                const int k = e / D, j = e - k * D;
                const int p = pp[k], q = qq[k];
                if (q < 0) continue;
                const double c = cs[k], s = sn[k];
                const double ap = A[p * LD + j], aq = A[q * LD + j];
                A[p * LD + j] = c * ap - s * aq;
                A[q * LD + j] = s * ap + c * aq;
                const double vp = VT[(size_t)p * D + j], vq = VT[(size_t)q * D + j];
                VT[(size_t)p * D + j] = c * vp - s * vq;
                VT[(size_t)q * D + j] = s * vp + c * vq;
            }
            __syncthreads();
            // columns: A <- A J
            for (int e = tid; e < half * D; e += EIGH_THREADS) {
                const int k = e / D, i = e - k * D;
                const int p = pp[k], q = qq[k];
                if (q < 0) continue;
                const double c = cs[k], s = sn[k];
                const double ap = A[i * LD + p], aq = A[i * LD + q];
                A[i * LD + p] = c * ap - s * aq;
                A[i * LD + q] = s * ap + c * aq;
            }
            __syncthreads();
        }
    }
    // eigenvalues and f(lambda)
    __syncthreads();
    for (int k = tid; k < D; k += EIGH_THREADS) {
        const double lam = A[k * LD + k];
        eigvals[(size_t)blockIdx.x * D + k] = lam;
    }
    if (fn == 0 || out == nullptr) return;
    if (fn == 3) {  // the eigenvectors themselves, one per row (row k belongs to eigvals[k])
        double* ob = out + boff;
        for (int e = tid; e < D * D; e += EIGH_THREADS) ob[e] = VT[e];
        return;
    }
    // stash f(lambda) in A's pad column A[k][D]
    for (int k = tid; k < D; k += EIGH_THREADS) {
        const double lam = A[k * LD + k];
        A[k * LD + D] = (fn == 1) ? sqrt(lam) : 1.0 / sqrt(lam);
    }
    __syncthreads();
    double* ob = out + boff;
    for (int e = tid; e < D * D; e += EIGH_THREADS) {
        const int i = e / D, j = e - i * D;
        double s = 0.0;
        for (int k = 0; k < D; ++k) s = fma(A[k * LD + D] * VT[(size_t)k * D + i], VT[(size_t)k * D + j], s);
        ob[e] = s;
    }
}


// ------------------------------------------------------------------------------------------------ D > 128: block Jacobi
// Two-sided block Jacobi with the parallel (round-robin) ordering: the index set is cut into blocks of EIGB; a round
// pairs the blocks into Dp/(2 EIGB) disjoint pairs (I, J); each pair's 32 x 32 sub-matrix [A_II A_IJ; A_JI A_JJ] is
// diagonalised in LDS by eigh_kernel (U = its eigenvector rows), then rows I u J of A and of V^T are replaced by
// U * rows, and columns I u J of A by cols * U^T.  A and V^T (8 MiB each at D = 1024) stay L2/MALL-resident.  No host
// synchronisation: convergence is a device flag that turns the remaining launches into no-ops.
__device__ __forceinline__ void eb_pair(int round, int k, int nblk, int& I, int& J) {
    const int m = nblk - 1;
    if (k == 0) {
        I = m;
        J = round % m;
    } else {
        I = (round + k) % m;
        J = (round - k + m) % m;
    }
    if (I > J) {
        const int t = I;
        I = J;
        J = t;
    }
}
__device__ __forceinline__ int eb_index(int a, int I, int J) { return a < EIGB ? I * EIGB + a : J * EIGB + (a - EIGB); }

__global__ __launch_bounds__(256) void eb_init_kernel(const double* __restrict__ Ain, int D, int Dp, double* __restrict__ Aw,
                                                      double* __restrict__ Vt, int* __restrict__ done) {
    for (size_t e = blockIdx.x * 256 + threadIdx.x; e < (size_t)Dp * Dp; e += (size_t)gridDim.x * 256) {
        const int i = e / Dp, j = e - (size_t)i * Dp;
        double v = 0.0;
        if (i < D && j < D) v = (i >= j) ? Ain[(size_t)i * D + j] : Ain[(size_t)j * D + i];
        Aw[e] = v;
        Vt[e] = (i == j) ? 1.0 : 0.0;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *done = 0;
}

__global__ __launch_bounds__(1024) void eb_conv_kernel(const double* __restrict__ Aw, int Dp, int* __restrict__ done) {
    __shared__ double red[2][16];
    if (*done) return;
    double off = 0.0, tot = 0.0;
    for (size_t e = threadIdx.x; e < (size_t)Dp * Dp; e += 1024) {
        const int i = e / Dp, j = e - (size_t)i * Dp;
        const double v = Aw[e];
        tot += v * v;
        if (i != j) off += v * v;
    }
    off = wave_sum(off);
    tot = wave_sum(tot);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = off;
        red[1][threadIdx.x >> 6] = tot;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double o = 0.0, t = 0.0;
        for (int w = 0; w < 16; ++w) {
            o += red[0][w];
            t += red[1][w];
        }
        if (o <= 1e-27 * t || t == 0.0) *done = 1;
    }
}

__global__ __launch_bounds__(256) void eb_gather_kernel(const double* __restrict__ Aw, int Dp, int round, int nblk,
                                                        double* __restrict__ S, const int* __restrict__ done) {
    if (*done) return;
    int I, J;
    eb_pair(round, blockIdx.x, nblk, I, J);
    double* Sk = S + (size_t)blockIdx.x * 4 * EIGB * EIGB;
    for (int e = threadIdx.x; e < 4 * EIGB * EIGB; e += 256) {
        const int a = e / (2 * EIGB), c = e - a * (2 * EIGB);
        Sk[e] = Aw[(size_t)eb_index(a, I, J) * Dp + eb_index(c, I, J)];
    }
}

// rows I u J of M (= A for blockIdx.z == 0, V^T for 1) <- U_k * rows; one thread per column
__global__ __launch_bounds__(256) void eb_rows_kernel(double* __restrict__ Aw, double* __restrict__ Vt, const double* __restrict__ U,
                                                      int Dp, int round, int nblk, const int* __restrict__ done) {
    __shared__ double Us[2 * EIGB][2 * EIGB + 1];
    if (*done) return;
    int I, J;
    eb_pair(round, blockIdx.y, nblk, I, J);
    const double* Uk = U + (size_t)blockIdx.y * 4 * EIGB * EIGB;
    for (int e = threadIdx.x; e < 4 * EIGB * EIGB; e += 256) Us[e / (2 * EIGB)][e % (2 * EIGB)] = Uk[e];
    __syncthreads();
    double* M = blockIdx.z == 0 ? Aw : Vt;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= Dp) return;
    double v[2 * EIGB];
#pragma unroll
    for (int c = 0; c < 2 * EIGB; ++c) v[c] = M[(size_t)eb_index(c, I, J) * Dp + j];
    for (int r = 0; r < 2 * EIGB; ++r) {
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < 2 * EIGB; ++c) s = fma(Us[r][c], v[c], s);
        M[(size_t)eb_index(r, I, J) * Dp + j] = s;
    }
}

// columns I u J of A <- cols * U_k^T; one thread per row
__global__ __launch_bounds__(256) void eb_cols_kernel(double* __restrict__ Aw, const double* __restrict__ U, int Dp, int round,
                                                      int nblk, const int* __restrict__ done) {
    __shared__ double Us[2 * EIGB][2 * EIGB + 1];
    if (*done) return;
    int I, J;
    eb_pair(round, blockIdx.y, nblk, I, J);
    const double* Uk = U + (size_t)blockIdx.y * 4 * EIGB * EIGB;
    for (int e = threadIdx.x; e < 4 * EIGB * EIGB; e += 256) Us[e / (2 * EIGB)][e % (2 * EIGB)] = Uk[e];
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Dp) return;
    double v[2 * EIGB];
    double* row = Aw + (size_t)i * Dp;
#pragma unroll
    for (int c = 0; c < 2 * EIGB; ++c) v[c] = row[eb_index(c, I, J)];
    for (int r = 0; r < 2 * EIGB; ++r) {
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < 2 * EIGB; ++c) s = fma(v[c], Us[r][c], s);
        row[eb_index(r, I, J)] = s;
    }
}

// eigenvalues, and dense D x D operands of f(A) = Vd^T (diag(f(lambda)) Vd)
__global__ __launch_bounds__(256) void eb_finish_kernel(const double* __restrict__ Aw, const double* __restrict__ Vt, int D, int Dp,
                                                        int fn, double* __restrict__ eigvals, double* __restrict__ Vd,
                                                        double* __restrict__ Td) {
    for (size_t e = blockIdx.x * 256 + threadIdx.x; e < (size_t)D * D; e += (size_t)gridDim.x * 256) {
        const int k = e / D, j = e - (size_t)k * D;
        const double lam = Aw[(size_t)k * Dp + k];
        if (j == 0) eigvals[k] = lam;
        if (fn != 0) {
            const double v = Vt[(size_t)k * Dp + j];
            Vd[e] = v;  // fn == 3: Vd is the caller's `out`
            if (fn != 3) Td[e] = ((fn == 1) ? sqrt(lam) : 1.0 / sqrt(lam)) * v;
        }
    }
}

static int eigh_block(const double* A, int nb, int D, int fn, double* out, double* eigvals, double* ws, hipStream_t st);

static size_t eigh_lds_bytes(int D) {
    const int De = (D + 1) & ~1, half = De / 2;
    return ((size_t)D * (D + 1) + 2 * half + 16) * sizeof(double) + (2 * half + 2) * sizeof(int);
}
static size_t g_eigh_lds_set = 0;

extern "C" int otvae_eigh_fn(const double* A, int nb, int D, int fn, double* out, double* eigvals, void* ws, void* stream) {
    OTVAE_REQUIRE(A && eigvals && ws && nb > 0 && D > 0, "otvae_eigh_fn: bad argument");
    OTVAE_REQUIRE(fn >= 0 && fn <= 3, "otvae_eigh_fn: fn must be 0, 1, 2 or 3");
    OTVAE_REQUIRE(fn == 0 || out, "otvae_eigh_fn: out missing");
    if (D > EIGH_MAX_D && D <= 1024 && !getenv("OTVAE_EIGH_TWOSIDED"))
        return eigh_block_onesided(A, nb, D, fn, out, eigvals, ws, (hipStream_t)stream);
    if (D > EIGH_MAX_D) return eigh_block(A, nb, D, fn, out, eigvals, (double*)ws, (hipStream_t)stream);
    if (!getenv("OTVAE_EIGH_TWOSIDED")) return eigh_onesided(A, nb, D, fn, out, eigvals, ws, (hipStream_t)stream, nullptr, nullptr, nullptr);
    const size_t lds = eigh_lds_bytes(D);
    if (lds > 65536 && lds > g_eigh_lds_set) {
        if (hipFuncSetAttribute((const void*)eigh_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            otvae_set_error("otvae_eigh_fn: cannot raise dynamic LDS limit");
            return OTVAE_ELAUNCH;
        }
        g_eigh_lds_set = 160 * 1024;
    }
    eigh_kernel<<<nb, EIGH_THREADS, lds, (hipStream_t)stream>>>(A, D, fn, out, eigvals, (double*)ws, nullptr);
    OTVAE_CHECK_LAUNCH("otvae_eigh_fn");
    return OTVAE_OK;
}

// The same decomposition started from an orthonormal basis the caller already has (Vinit[b][k][:] = vector k: the eigenvectors of
// a nearby matrix, e.g. the previous training step's latent covariance under GaussianW2Prior): the one-sided iteration then
// begins with the nearly orthogonal columns A v_k and needs 2-4 sweeps instead of ~9.  A must be symmetric in BOTH triangles
// here (it enters a product); g0: scratch of nb * D * D doubles.  warm (nullable): a device int the kernels read -- 0 means
// "Vinit holds nothing yet" and the call runs cold, so that a captured training step can make that decision per replay.
// D > 128 (or the two-sided solver): Vinit is ignored.
extern "C" int otvae_eigh_fn_warm(const double* A, const double* Vinit, const int* warm, int nb, int D, int fn, double* out,
                                  double* eigvals, void* ws, double* g0, void* stream) {
    OTVAE_REQUIRE(A && eigvals && ws && nb > 0 && D > 0, "otvae_eigh_fn_warm: bad argument");
    OTVAE_REQUIRE(fn >= 0 && fn <= 3, "otvae_eigh_fn_warm: fn must be 0, 1, 2 or 3");
    OTVAE_REQUIRE(fn == 0 || out, "otvae_eigh_fn_warm: out missing");
    OTVAE_REQUIRE(!Vinit || g0, "otvae_eigh_fn_warm: scratch for the start columns missing");
    if (D > EIGH_MAX_D || getenv("OTVAE_EIGH_TWOSIDED") || !Vinit) return otvae_eigh_fn(A, nb, D, fn, out, eigvals, ws, stream);
    return eigh_onesided(A, nb, D, fn, out, eigvals, ws, (hipStream_t)stream, Vinit, g0, warm);
}

// make_psd: A_b += shift_b I, shift_b = |min(lmin_b, 0)| (+1e-8 if strict), optionally only if some matrix fails
__global__ __launch_bounds__(256) void make_psd_kernel(double* __restrict__ A, const double* __restrict__ eigvals, int nb,
                                                       int D, int strict, int cond_any) {
    __shared__ int any_bad;
    if (threadIdx.x == 0) any_bad = cond_any ? 0 : 1;
    __syncthreads();
    if (cond_any) {
        int bad = 0;
        for (int b = threadIdx.x; b < nb; b += 256) {
            double mn = INFINITY;
            for (int k = 0; k < D; ++k) mn = fmin(mn, eigvals[(size_t)b * D + k]);
            if (strict ? !(mn > 0.0) : !(mn >= 0.0)) bad = 1;
        }
        if (bad) any_bad = 1;  // benign race: all writers store 1
        __syncthreads();
    }
    if (!any_bad) return;
    for (int b = 0; b < nb; ++b) {
        double mn = INFINITY;
        for (int k = 0; k < D; ++k) mn = fmin(mn, eigvals[(size_t)b * D + k]);
        double shift = fabs(fmin(mn, 0.0));
        if (strict) shift += 1e-8;
        for (int i = threadIdx.x; i < D; i += 256) A[((size_t)b * D + i) * D + i] += shift;
    }
}

extern "C" int otvae_make_psd(double* A, const double* eigvals, int nb, int D, int strict, int cond_any, void* stream) {
    OTVAE_REQUIRE(A && eigvals && nb > 0 && D > 0, "otvae_make_psd: bad argument");
    make_psd_kernel<<<1, 256, 0, (hipStream_t)stream>>>(A, eigvals, nb, D, strict, cond_any);
    OTVAE_CHECK_LAUNCH("otvae_make_psd");
    return OTVAE_OK;
}

// Lower Cholesky factor A = L L^T (fp64), one workgroup per matrix, left-looking by columns: what MultivariateNormal's sampler
// factors the noise covariance with (torch.distributions, used by the reference's stochastic apply_transport, ot/w2_utils.py:
// 521-525).  Column j: L[j][j] = sqrt(A[j][j] - |L[j][:j]|^2), L[i][j] = (A[i][j] - L[i][:j] . L[j][:j]) / L[j][j]; row j of L is
// staged in LDS for the column's dot products.  info[b] = 1 + index of the first non-positive pivot (0 = success).
__global__ __launch_bounds__(256) void cholesky_kernel(const double* __restrict__ A, int D, double* __restrict__ L,
                                                       int* __restrict__ info) {
    extern __shared__ double chol_row[];  // [D]
    __shared__ double red[4];
    __shared__ double s_piv;
    const double* Ab = A + (size_t)blockIdx.x * D * D;
    double* Lb = L + (size_t)blockIdx.x * D * D;
    int bad = 0;
    for (int j = 0; j < D; ++j) {
        double s = 0.0;
        for (int k = threadIdx.x; k < j; k += 256) {
            const double v = Lb[(size_t)j * D + k];
            chol_row[k] = v;
            s = fma(v, v, s);
        }
        s = wave_sum(s);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            const double piv = Ab[(size_t)j * D + j] - ((red[0] + red[1]) + (red[2] + red[3]));
            s_piv = piv > 0.0 ? sqrt(piv) : NAN;
        }
        __syncthreads();
        const double ljj = s_piv;
        if (!(ljj > 0.0) && !bad) bad = j + 1;
        for (int i = j + threadIdx.x; i < D; i += 256) {
            if (i == j) {
                Lb[(size_t)j * D + j] = ljj;
                continue;
            }
            double acc = Ab[(size_t)i * D + j];  // lower triangle of A
            const double* li = Lb + (size_t)i * D;
            for (int k = 0; k < j; ++k) acc = fma(-li[k], chol_row[k], acc);
            Lb[(size_t)i * D + j] = acc / ljj;
        }
        for (int i = threadIdx.x; i < j; i += 256) Lb[(size_t)i * D + j] = 0.0;  // strict upper triangle
        __syncthreads();
    }
    if (threadIdx.x == 0 && info) info[blockIdx.x] = bad;
}

// Blocked right-looking variant for D > 128 (one workgroup reads the whole trailing matrix through one CU otherwise: 90 ms at
// D = 1024): panels of 64 columns; per panel (i) the 64 x 64 diagonal block is factored in LDS by one workgroup, (ii) the rows
// below solve X L11^T = A21 (a thread per row, L11 in LDS), (iii) the trailing matrix takes A22 -= L21 L21^T in 64 x 64 tiles
// (lower triangle only).  L is built in place in its output buffer, which starts as a copy of A's lower triangle.
#define CHB 64
// ls / is: distance between consecutive matrices' factors (doubles) and info words (ints): D * D and 1 for the C entry point, the
// per-matrix workspace pitch when the eigensolver factors a whole batch into its own buffers
__global__ __launch_bounds__(256) void chol_copy_lower_kernel(const double* __restrict__ A, int D, double* __restrict__ L, size_t ls,
                                                              int* __restrict__ info, size_t is) {
    const size_t boff = (size_t)blockIdx.y * D * D;
    double* Lb = L + (size_t)blockIdx.y * ls;
    for (size_t e = blockIdx.x * (size_t)256 + threadIdx.x; e < (size_t)D * D; e += (size_t)gridDim.x * 256) {
        const int i = (int)(e / D), j = (int)(e - (size_t)i * D);
        Lb[e] = j <= i ? A[boff + e] : 0.0;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && info) info[blockIdx.y * is] = 0;
}

// (i) + (ii): every workgroup factors the diagonal block [k0, k0+nbk) in its own LDS (64^3 / 3 flops: cheaper than handing it
// over) and solves its 256 rows below against it; WRITE_DIAG: the launch of one workgroup per matrix that stores the factored
// diagonal block afterwards (no workgroup of the solve launch may see it half written).
template <bool WRITE_DIAG>
__global__ __launch_bounds__(256) void chol_panel_kernel(double* __restrict__ L, size_t ls, int D, int k0, int* __restrict__ info,
                                                         size_t is) {
    __shared__ double d[CHB][CHB + 1];
    __shared__ int s_bad;
    double* Lb = L + (size_t)blockIdx.y * ls;
    const int nbk = min(CHB, D - k0);
    for (int e = threadIdx.x; e < nbk * nbk; e += 256) {
        const int i = e / nbk, j = e - i * nbk;
        d[i][j] = j <= i ? Lb[(size_t)(k0 + i) * D + k0 + j] : 0.0;
    }
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    for (int j = 0; j < nbk; ++j) {  // unblocked factorisation of the diagonal block, column by column
        if (threadIdx.x == 0) {
            const double piv = d[j][j];
            if (!(piv > 0.0) && !s_bad) s_bad = k0 + j + 1;
            d[j][j] = piv > 0.0 ? sqrt(piv) : NAN;
        }
        __syncthreads();
        const double ljj = d[j][j];
        for (int i = j + 1 + threadIdx.x; i < nbk; i += 256) d[i][j] /= ljj;
        __syncthreads();
        for (int e = threadIdx.x; e < (nbk - j - 1) * (nbk - j - 1); e += 256) {  // trailing update inside the block (lower part)
            const int i = j + 1 + e / (nbk - j - 1), c = j + 1 + e % (nbk - j - 1);
            if (c <= i) d[i][c] = fma(-d[i][j], d[c][j], d[i][c]);
        }
        __syncthreads();
    }
    if (WRITE_DIAG) {
        for (int e = threadIdx.x; e < nbk * nbk; e += 256) {
            const int i = e / nbk, j = e - i * nbk;
            if (j <= i) Lb[(size_t)(k0 + i) * D + k0 + j] = d[i][j];
        }
        if (threadIdx.x == 0 && s_bad && info && info[blockIdx.y * is] == 0) info[blockIdx.y * is] = s_bad;
        return;
    }
    // rows below the panel: row r of A21 <- solve x L11^T = a  (forward substitution along the row; the row's 64 entries live in
    // registers -- re-reading its own earlier results from global memory made this launch 218 us at D = 1024)
    const int r = k0 + nbk + blockIdx.x * 256 + threadIdx.x;
    if (r < D) {
        double* row = Lb + (size_t)r * D + k0;
        double x[CHB];
#pragma unroll
        for (int j = 0; j < CHB; ++j) x[j] = j < nbk ? row[j] : 0.0;
#pragma unroll
        for (int j = 0; j < CHB; ++j) {
            if (j < nbk) {
                double acc = x[j];
#pragma unroll
                for (int c = 0; c < j; ++c) acc = fma(-x[c], d[j][c], acc);
                x[j] = acc / d[j][j];
            }
        }
#pragma unroll
        for (int j = 0; j < CHB; ++j)
            if (j < nbk) row[j] = x[j];
    }
}

// (iii) A22[i][j] -= sum_c L21[i][c] L21[j][c] for the 64 x 64 tile (ti, tj), tj <= ti, of the trailing matrix
__global__ __launch_bounds__(256) void chol_syrk_kernel(double* __restrict__ L, size_t ls, int D, int k0) {
    __shared__ double a[CHB][17], b[CHB][17];
    double* Lb = L + (size_t)blockIdx.z * ls;
    const int t0 = k0 + CHB;
    const int ti = blockIdx.y, tj = blockIdx.x;
    if (tj > ti) return;
    const int i0 = t0 + ti * CHB, j0 = t0 + tj * CHB;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;  // 16 x 16 threads, 4 x 4 outputs each
    double acc[4][4] = {};
    for (int c0 = 0; c0 < CHB; c0 += 16) {
        for (int e = threadIdx.x; e < CHB * 16; e += 256) {
            const int rr = e >> 4, cc = e & 15;
            a[rr][cc] = (i0 + rr < D) ? Lb[(size_t)(i0 + rr) * D + k0 + c0 + cc] : 0.0;
            b[rr][cc] = (j0 + rr < D) ? Lb[(size_t)(j0 + rr) * D + k0 + c0 + cc] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            double av[4], bv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                av[u] = a[ty + 16 * u][c];
                bv[u] = b[tx + 16 * u][c];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[u][v] = fma(av[u], bv[v], acc[u][v]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int i = i0 + ty + 16 * u, j = j0 + tx + 16 * v;
            if (i < D && j <= i) Lb[(size_t)i * D + j] -= acc[u][v];
        }
}

// the blocked factorisation of nb matrices side by side; L / info of matrix b at L + b * ls, info + b * is
int cholesky_blocked(const double* A, int nb, int D, double* L, size_t ls, int* info, size_t is, hipStream_t st) {
    chol_copy_lower_kernel<<<dim3(imin(cdiv((size_t)D * D, 256), 1024), nb), 256, 0, st>>>(A, D, L, ls, info, is);
    for (int k0 = 0; k0 < D; k0 += CHB) {
        const int below = D - k0 - CHB;
        if (below > 0) chol_panel_kernel<false><<<dim3(cdiv(below, 256), nb), 256, 0, st>>>(L, ls, D, k0, info, is);
        chol_panel_kernel<true><<<dim3(1, nb), 256, 0, st>>>(L, ls, D, k0, info, is);
        if (below > 0) {
            const int nt = cdiv(below, CHB);
            chol_syrk_kernel<<<dim3(nt, nt, nb), 256, 0, st>>>(L, ls, D, k0);
        }
    }
    OTVAE_CHECK_LAUNCH("otvae_cholesky(blocked)");
    return OTVAE_OK;
}

extern "C" int otvae_cholesky(const double* A, int nb, int D, double* L, int* info, void* stream) {
    OTVAE_REQUIRE(A && L && nb > 0 && D > 0 && A != L, "otvae_cholesky: bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (D > 128) return cholesky_blocked(A, nb, D, L, (size_t)D * D, info, 1, st);

    OTVAE_REQUIRE((size_t)D * 8 <= 64 * 1024, "otvae_cholesky: D = %d exceeds the LDS row buffer (8192)", D);
    cholesky_kernel<<<nb, 256, (size_t)D * sizeof(double), (hipStream_t)stream>>>(A, D, L, info);
    OTVAE_CHECK_LAUNCH("otvae_cholesky");
    return OTVAE_OK;
}

// ================================================================================================ dense helpers
template <typename T>
__global__ __launch_bounds__(256) void gemm_tile_kernel(int transA, int transB, int m, int n, int k, T alpha,
                                                        const T* __restrict__ A, size_t sA, const T* __restrict__ B,
                                                        size_t sB, T beta, T* __restrict__ Cm) {
    __shared__ T as[16][17], bs[16][17];
    const int b = blockIdx.z;
    const T* Ab = A + (size_t)b * sA;
    const T* Bb = B + (size_t)b * sB;
    T* Cb = Cm + (size_t)b * m * n;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int i = blockIdx.y * 16 + ty, j = blockIdx.x * 16 + tx;
    T acc = (T)0;
    for (int k0 = 0; k0 < k; k0 += 16) {
        const int ka = k0 + tx;  // A tile: rows i (ty), cols k (tx)
        as[ty][tx] = (i < m && ka < k) ? (transA ? Ab[(size_t)ka * m + i] : Ab[(size_t)i * k + ka]) : (T)0;
        const int kb = k0 + ty;  // B tile: rows k (ty), cols j (tx)
        bs[ty][tx] = (kb < k && j < n) ? (transB ? Bb[(size_t)j * k + kb] : Bb[(size_t)kb * n + j]) : (T)0;
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) acc = fma(as[ty][kk], bs[kk][tx], acc);
        __syncthreads();
    }
    if (i < m && j < n) {
        const size_t o = (size_t)i * n + j;
        Cb[o] = alpha * acc + (beta != (T)0 ? beta * Cb[o] : (T)0);
    }
}
#define gemm_f64_kernel gemm_tile_kernel<double>

// fp32 twin for the small weighted sums of the mixture / codebook models ([B, K] x [K, d], reference base.py:241-251,
// codebook_model.py:145-148): a plain tiled product, the operands are a few hundred KB at most
extern "C" int otvae_gemm_f32(int transA, int transB, int nb, int m, int n, int k, float alpha, const float* A, int a_bcast,
                              const float* B, int b_bcast, float beta, float* C, void* stream) {
    OTVAE_REQUIRE(A && B && C && nb > 0 && m > 0 && n > 0 && k > 0, "otvae_gemm_f32: bad argument");
    gemm_tile_kernel<float><<<dim3(cdiv(n, 16), cdiv(m, 16), nb), 256, 0, (hipStream_t)stream>>>(
        transA, transB, m, n, k, alpha, A, a_bcast ? 0 : (size_t)m * k, B, b_bcast ? 0 : (size_t)k * n, beta, C);
    OTVAE_CHECK_LAUNCH("otvae_gemm_f32");
    return OTVAE_OK;
}

// y[r][:] = softmax(scale * x[r][:]) over the last dimension and its backward gx = scale * y o (gy - sum_k y gy): the assignment
// distributions of the mixture models (softmax(energy / temperature), base.py:216-224; F.gumbel_softmax, :234-235).  A wave per row.
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const T* __restrict__ x, long rows, int K, T scale, T* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const T* xr = x + r * K;
    T* yr = y + r * K;
    T mx = -INFINITY;
    for (int k = lane; k < K; k += 64) mx = fmax(mx, scale * xr[k]);
    mx = wave_max(mx);
    T sum = (T)0;
    for (int k = lane; k < K; k += 64) sum += exp(scale * xr[k] - mx);
    sum = wave_sum(sum);
    const T inv = (T)1 / sum;
    for (int k = lane; k < K; k += 64) yr[k] = exp(scale * xr[k] - mx) * inv;
}

template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const T* __restrict__ y, const T* __restrict__ gy, long rows, int K, T scale,
                                                               T* __restrict__ gx) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const T* yr = y + r * K;
    const T* gr = gy + r * K;
    T dot = (T)0;
    for (int k = lane; k < K; k += 64) dot = fma(yr[k], gr[k], dot);
    dot = wave_sum(dot);
    for (int k = lane; k < K; k += 64) gx[r * K + k] = scale * yr[k] * (gr[k] - dot);
}

extern "C" int otvae_softmax_rows(int dtype, const void* x, int64_t rows, int K, double scale, void* y, void* stream) {
    OTVAE_REQUIRE(x && y && rows > 0 && K > 0 && (dtype == 0 || dtype == 1), "otvae_softmax_rows: bad argument");
    const int grid = cdiv(rows, 4);
    if (dtype == 0)
        softmax_rows_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>((const float*)x, rows, K, (float)scale, (float*)y);
    else
        softmax_rows_kernel<double><<<grid, 256, 0, (hipStream_t)stream>>>((const double*)x, rows, K, scale, (double*)y);
    OTVAE_CHECK_LAUNCH("otvae_softmax_rows");
    return OTVAE_OK;
}

extern "C" int otvae_softmax_rows_bwd(int dtype, const void* y, const void* gy, int64_t rows, int K, double scale, void* gx,
                                      void* stream) {
    OTVAE_REQUIRE(y && gy && gx && rows > 0 && K > 0 && (dtype == 0 || dtype == 1), "otvae_softmax_rows_bwd: bad argument");
    const int grid = cdiv(rows, 4);
    if (dtype == 0)
        softmax_rows_bwd_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>((const float*)y, (const float*)gy, rows, K, (float)scale,
                                                                               (float*)gx);
    else
        softmax_rows_bwd_kernel<double><<<grid, 256, 0, (hipStream_t)stream>>>((const double*)y, (const double*)gy, rows, K, scale,
                                                                                (double*)gx);
    OTVAE_CHECK_LAUNCH("otvae_softmax_rows_bwd");
    return OTVAE_OK;
}

// out[r] = log sum_k exp(x[r][k]) (one wave per row) and its backward gx[r][k] = g[r] exp(x[r][k] - out[r]): the mixture log-density
// read-out of GaussianMixtureModel.predict (gaussian_model.py:129-132 on a MixtureSameFamily: logsumexp over the components)
template <typename T>
__global__ __launch_bounds__(256) void lse_rows_kernel(const T* __restrict__ x, long rows, int K, T* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const T* xr = x + r * K;
    T mx = -INFINITY;
    for (int k = lane; k < K; k += 64) mx = fmax(mx, xr[k]);
    mx = wave_max(mx);
    T sum = (T)0;
    if (mx > -INFINITY && mx < INFINITY)
        for (int k = lane; k < K; k += 64) sum += exp(xr[k] - mx);
    sum = wave_sum(sum);
    // torch.logsumexp: an all -inf row gives -inf, a row holding +inf gives +inf
    if (lane == 0) out[r] = (mx > -INFINITY && mx < INFINITY) ? mx + log(sum) : mx;
}

template <typename T>
__global__ __launch_bounds__(256) void lse_rows_bwd_kernel(const T* __restrict__ x, const T* __restrict__ lse, const T* __restrict__ g,
                                                           long rows, int K, T* __restrict__ gx) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const T l = lse[r], gr = g[r];
    for (int k = lane; k < K; k += 64) gx[r * K + k] = gr * exp(x[r * K + k] - l);
}

extern "C" int otvae_lse_rows(int dtype, const void* x, int64_t rows, int K, void* out, void* stream) {
    OTVAE_REQUIRE(x && out && rows > 0 && K > 0 && (dtype == 0 || dtype == 1), "otvae_lse_rows: bad argument");
    const int grid = cdiv(rows, 4);
    if (dtype == 0) lse_rows_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>((const float*)x, rows, K, (float*)out);
    else lse_rows_kernel<double><<<grid, 256, 0, (hipStream_t)stream>>>((const double*)x, rows, K, (double*)out);
    OTVAE_CHECK_LAUNCH("otvae_lse_rows");
    return OTVAE_OK;
}

extern "C" int otvae_lse_rows_bwd(int dtype, const void* x, const void* lse, const void* g, int64_t rows, int K, void* gx, void* stream) {
    OTVAE_REQUIRE(x && lse && g && gx && rows > 0 && K > 0 && (dtype == 0 || dtype == 1), "otvae_lse_rows_bwd: bad argument");
    const int grid = cdiv(rows, 4);
    if (dtype == 0)
        lse_rows_bwd_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>((const float*)x, (const float*)lse, (const float*)g, rows, K, (float*)gx);
    else
        lse_rows_bwd_kernel<double><<<grid, 256, 0, (hipStream_t)stream>>>((const double*)x, (const double*)lse, (const double*)g, rows, K,
                                                                            (double*)gx);
    OTVAE_CHECK_LAUNCH("otvae_lse_rows_bwd");
    return OTVAE_OK;
}

// The same product on the fp64 matrix cores for the large operands (D = 1024 latent transport: the products around the
// eigendecompositions were 2-4 ms each on the kernel above).  64 x 64 output tile per workgroup, 4 waves x (16 rows x 64 columns) =
// 4 v_mfma_f64_16x16x4_f64 accumulators per wave, K in chunks of 16 through LDS (k-major for both operands, so that lane
// (i | j = l % 16, k = l / 16) reads its A / B element with one ds_read_b64).  Accumulator layout (probed on gfx950,
// tools/probe/mfma_f64.hip): register r of lane l holds D[4 r + l / 16][l % 16].
typedef double double4v __attribute__((ext_vector_type(4)));
#define GM_T 64
#define GM_KC 16
#define GM_LD (GM_T + 4)

__global__ __launch_bounds__(256) void gemm_f64_mfma_kernel(int transA, int transB, int m, int n, int k, double alpha,
                                                            const double* __restrict__ A, size_t sA, const double* __restrict__ B,
                                                            size_t sB, double beta, double* __restrict__ Cm) {
    __shared__ double as[GM_KC][GM_LD], bs[GM_KC][GM_LD];
    const int b = blockIdx.z;
    const double* Ab = A + (size_t)b * sA;
    const double* Bb = B + (size_t)b * sB;
    double* Cb = Cm + (size_t)b * m * n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = blockIdx.y * GM_T, j0 = blockIdx.x * GM_T;
    const int l16 = lane & 15, lk = lane >> 4;
    double4v acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = double4v{0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < k; k0 += GM_KC) {
        // stage: each operand along its contiguous direction
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (transA) {  // stored [k][m]
                const int kk = (tid >> 6) + 4 * u, i = tid & 63;
                as[kk][i] = (k0 + kk < k && i0 + i < m) ? Ab[(size_t)(k0 + kk) * m + i0 + i] : 0.0;
            } else {       // stored [m][k]
                const int i = (tid >> 4) + 16 * u, kk = tid & 15;
                as[kk][i] = (k0 + kk < k && i0 + i < m) ? Ab[(size_t)(i0 + i) * k + k0 + kk] : 0.0;
            }
            if (transB) {  // stored [n][k]
                const int j = (tid >> 4) + 16 * u, kk = tid & 15;
                bs[kk][j] = (k0 + kk < k && j0 + j < n) ? Bb[(size_t)(j0 + j) * k + k0 + kk] : 0.0;
            } else {       // stored [k][n]
                const int kk = (tid >> 6) + 4 * u, j = tid & 63;
                bs[kk][j] = (k0 + kk < k && j0 + j < n) ? Bb[(size_t)(k0 + kk) * n + j0 + j] : 0.0;
            }
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < GM_KC / 4; ++ks) {
            const double av = as[ks * 4 + lk][wave * 16 + l16];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bs[ks * 4 + lk][t * 16 + l16], acc[t], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = i0 + wave * 16 + 4 * r + lk, j = j0 + t * 16 + l16;
            if (i < m && j < n) {
                const size_t o = (size_t)i * n + j;
                Cb[o] = alpha * acc[t][r] + (beta != 0.0 ? beta * Cb[o] : 0.0);
            }
        }
}

void gemm_f64_launch(int transA, int transB, int nb, int m, int n, int k, double alpha, const double* A, size_t sA, const double* B,
                     size_t sB, double beta, double* C, hipStream_t st) {
    if (imax(m, n) >= 256 && k >= 64) {  // enough tiles to matter: the matrix cores
        gemm_f64_mfma_kernel<<<dim3(cdiv(n, GM_T), cdiv(m, GM_T), nb), 256, 0, st>>>(transA, transB, m, n, k, alpha, A, sA, B, sB, beta, C);
    } else {
        gemm_f64_kernel<<<dim3(cdiv(n, 16), cdiv(m, 16), nb), 256, 0, st>>>(transA, transB, m, n, k, alpha, A, sA, B, sB, beta, C);
    }
}

extern "C" int otvae_gemm_f64(int transA, int transB, int nb, int m, int n, int k, double alpha, const double* A, int a_bcast,
                              const double* B, int b_bcast, double beta, double* C, void* stream) {
    OTVAE_REQUIRE(A && B && C && nb > 0 && m > 0 && n > 0 && k > 0, "otvae_gemm_f64: bad argument");
    gemm_f64_launch(transA, transB, nb, m, n, k, alpha, A, a_bcast ? 0 : (size_t)m * k, B, b_bcast ? 0 : (size_t)k * n, beta, C,
                    (hipStream_t)stream);
    OTVAE_CHECK_LAUNCH("otvae_gemm_f64");
    return OTVAE_OK;
}

static int eigh_block(const double* A, int nb, int D, int fn, double* out, double* eigvals, double* ws, hipStream_t st) {
    if (D > EIGH_BLOCK_MAX_D) {
        otvae_set_error("otvae_eigh_fn: D = %d > %d is not implemented", D, EIGH_BLOCK_MAX_D);
        return OTVAE_EUNSUPPORTED;
    }
    const int Dp = eigb_dp(D), nblk = Dp / EIGB, np = nblk / 2, S2 = 2 * EIGB;
    double* Aw = ws;
    double* Vt = Aw + (size_t)Dp * Dp;
    double* S = Vt + (size_t)Dp * Dp;
    double* U = S + (size_t)np * S2 * S2;
    double* evs = U + (size_t)np * S2 * S2;
    double* Vd = evs + (size_t)np * S2;
    double* Td = Vd + (size_t)D * D;
    int* done = reinterpret_cast<int*>(Td + (size_t)D * D);
    const size_t lds = eigh_lds_bytes(S2);
    const int chunks = cdiv(Dp, 256);
    for (int b = 0; b < nb; ++b) {
        const double* Ab = A + (size_t)b * D * D;
        eb_init_kernel<<<imin(cdiv((size_t)Dp * Dp, 256), 2048), 256, 0, st>>>(Ab, D, Dp, Aw, Vt, done);
        for (int sweep = 0; sweep < EIGH_BLOCK_SWEEPS; ++sweep) {
            eb_conv_kernel<<<1, 1024, 0, st>>>(Aw, Dp, done);
            for (int round = 0; round < nblk - 1; ++round) {
                eb_gather_kernel<<<np, 256, 0, st>>>(Aw, Dp, round, nblk, S, done);
                eigh_kernel<<<np, EIGH_THREADS, lds, st>>>(S, S2, 0, nullptr, evs, U, done);
                eb_rows_kernel<<<dim3(chunks, np, 2), 256, 0, st>>>(Aw, Vt, U, Dp, round, nblk, done);
                eb_cols_kernel<<<dim3(chunks, np), 256, 0, st>>>(Aw, U, Dp, round, nblk, done);
            }
        }
        eb_finish_kernel<<<imin(cdiv((size_t)D * D, 256), 2048), 256, 0, st>>>(Aw, Vt, D, Dp, fn, eigvals + (size_t)b * D,
                                                                              fn == 3 ? out + (size_t)b * D * D : Vd, Td);
        if (fn == 1 || fn == 2)  // out = Vd^T * Td
            gemm_f64_launch(1, 0, 1, D, D, D, 1.0, Vd, 0, Td, 0, 0.0, out + (size_t)b * D * D, st);
        OTVAE_CHECK_LAUNCH("otvae_eigh_fn(block)");
    }
    return OTVAE_OK;
}

__global__ __launch_bounds__(256) void w2_tail_kernel(const double* __restrict__ ms, const double* __restrict__ mt,
                                                      const double* __restrict__ cs, const double* __restrict__ ct,
                                                      const double* __restrict__ sq, int D, double* __restrict__ out) {
    __shared__ double red[4];
    const int b = blockIdx.x;
    double s = 0.0;
    for (int i = threadIdx.x; i < D; i += 256) {
        const double d = ms[(size_t)b * D + i] - mt[(size_t)b * D + i];
        const size_t dd = ((size_t)b * D + i) * D + i;
        s += d * d + (cs[dd] + ct[dd] - 2.0 * sq[dd]);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[b] = (red[0] + red[1]) + (red[2] + red[3]);
}

extern "C" int otvae_w2_tail(const double* ms, const double* mt, const double* cs, const double* ct, const double* sqrt_mix,
                             int nb, int D, double* out, void* stream) {
    OTVAE_REQUIRE(ms && mt && cs && ct && sqrt_mix && out && nb > 0 && D > 0, "otvae_w2_tail: bad argument");
    w2_tail_kernel<<<nb, 256, 0, (hipStream_t)stream>>>(ms, mt, cs, ct, sqrt_mix, D, out);
    OTVAE_CHECK_LAUNCH("otvae_w2_tail");
    return OTVAE_OK;
}

// ---- w2_gaussian + the eq. 17 operator of the same pair of Gaussians in one call -----------------------------------------------
// (reference ot/w2_utils.py:40-80 and 756-769 with their 'spd' argument validation, :661-669).  Inputs: means, covariances and the
// spectra (eigvals, Vt rows = eigenvectors) of both covariances.  Everything between the two rounds of eigendecompositions --
// definiteness shifts, V f(lambda) V^T for three functions, the two inner products, the symmetry test, the tail, the operator --
// runs here without a host read: the Python composition of the same steps was ~90 small launches and four synchronisations
// (2.3 ms of a 6.5 ms call at D = 128).  flags[0..2] = {some source / target covariance is not positive definite, the inner
// product is not symmetric}: read ONCE by the caller, who raises the reference's errors.
__global__ __launch_bounds__(256) void w2t_shift_kernel(const double* __restrict__ lam_s, const double* __restrict__ lam_t, int nb, int D,
                                                        int make_pd, double* __restrict__ shift, int* __restrict__ flags) {
    __shared__ int s_bad[2];
    if (threadIdx.x < 2) s_bad[threadIdx.x] = 0;
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * nb; e += 256) {
        const double* l = (e < nb ? lam_s : lam_t) + (size_t)(e % nb) * D;
        double mn = INFINITY;
        for (int k = 0; k < D; ++k) mn = fmin(mn, l[k]);
        shift[e] = fabs(fmin(mn, 0.0)) + 1e-8;   // psd_shift(strict=True)
        if (!(mn > 0.0)) s_bad[e < nb ? 0 : 1] = 1;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * nb; e += 256)   // only_if_needed: no shift unless a matrix of that side's batch fails
        if (!make_pd || !s_bad[e < nb ? 0 : 1]) shift[e] = 0.0;
    if (threadIdx.x < 2) flags[threadIdx.x] = s_bad[threadIdx.x];
    if (threadIdx.x == 2) flags[2] = 0;
}

__global__ __launch_bounds__(256) void w2t_prep_kernel(const double* __restrict__ cs, const double* __restrict__ ct,
                                                       const double* __restrict__ lam_s, const double* __restrict__ vt_s,
                                                       const double* __restrict__ lam_t, const double* __restrict__ vt_t, int nb, int D,
                                                       const double* __restrict__ shift, double* __restrict__ cs_v, double* __restrict__ ct_v,
                                                       double* __restrict__ f_rt, double* __restrict__ f_rs, double* __restrict__ f_irs) {
    const size_t total = (size_t)nb * D * D;
    for (size_t e = blockIdx.x * (size_t)256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int b = (int)(e / ((size_t)D * D));
        const int r = (int)((e / D) % D), c = (int)(e % D);
        const double ss = shift[b], st = shift[nb + b];
        cs_v[e] = cs[e] + (r == c ? ss : 0.0);
        ct_v[e] = ct[e] + (r == c ? st : 0.0);
        const double ls = lam_s[(size_t)b * D + r] + ss, lt = lam_t[(size_t)b * D + r] + st;
        f_rt[e] = sqrt(lt) * vt_t[e];
        f_rs[e] = sqrt(ls) * vt_s[e];
        f_irs[e] = vt_s[e] / sqrt(ls + 1e-8);   // (lambda + STABILITY_CONST)^-1/2
    }
}

__global__ __launch_bounds__(256) void w2t_sym_kernel(const double* __restrict__ m, int D, int* __restrict__ flags) {
    __shared__ double red[4];
    const double* mb = m + (size_t)blockIdx.x * D * D;
    double s = 0.0;
    for (size_t e = threadIdx.x; e < (size_t)D * D; e += 256) {
        const int r = (int)(e / D), c = (int)(e % D);
        const double d = mb[e] - mb[(size_t)c * D + r];
        s = fma(d, d, s);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0 && !(((red[0] + red[1]) + (red[2] + red[3])) < 1e-8)) flags[2] = 1;   // is_symmetric: sum (m - m^T)^2 < 1e-8
}

__global__ __launch_bounds__(256) void w2t_diag_add_kernel(double* __restrict__ T, int nb, int D, double v) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < nb * D) T[(size_t)(e / D) * D * D + (size_t)(e % D) * (D + 1)] += v;
}

extern "C" int64_t otvae_w2_transport_ws(int nb, int D) {
    if (nb <= 0 || D <= 0) return -1;
    // shifts | cs_v ct_v f_rt f_rs f_irs rt rs irs tmp | mixcat[2 nb] roots[2 nb] | eigenvalue scratch [2 nb][D]
    return (int64_t)(2 * nb + 64) * 8 + (int64_t)13 * nb * D * D * 8 + (int64_t)2 * nb * D * 8;
}

extern "C" int otvae_w2_transport(const double* ms, const double* mt, const double* cs, const double* ct, const double* lam_s,
                                  const double* vt_s, const double* lam_t, const double* vt_t, int nb, int D, double pg_star, int make_pd,
                                  void* ws, void* eigh_ws, double* w2, double* T, int* flags, void* stream) {
    OTVAE_REQUIRE(ms && mt && cs && ct && lam_s && vt_s && lam_t && vt_t && ws && eigh_ws && w2 && T && flags && nb > 0 && D > 0,
                  "otvae_w2_transport: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const size_t mat = (size_t)nb * D * D, one = (size_t)D * D;
    double* shift = (double*)ws;
    double* cs_v = shift + 2 * nb + 64;
    double *ct_v = cs_v + mat, *f_rt = ct_v + mat, *f_rs = f_rt + mat, *f_irs = f_rs + mat, *rt = f_irs + mat, *rs = rt + mat,
           *irs = rs + mat, *tmp = irs + mat, *mixcat = tmp + mat, *roots = mixcat + 2 * mat, *evs = roots + 2 * mat;
    w2t_shift_kernel<<<1, 256, 0, st>>>(lam_s, lam_t, nb, D, make_pd, shift, flags);
    w2t_prep_kernel<<<imin(cdiv(mat, 256), 2048), 256, 0, st>>>(cs, ct, lam_s, vt_s, lam_t, vt_t, nb, D, shift, cs_v, ct_v, f_rt, f_rs, f_irs);
    OTVAE_CHECK_LAUNCH("otvae_w2_transport(prep)");
    // V f(lambda) V^T = Vt^T (f Vt)
    gemm_f64_launch(1, 0, nb, D, D, D, 1.0, vt_t, one, f_rt, one, 0.0, rt, st);
    gemm_f64_launch(1, 0, nb, D, D, D, 1.0, vt_s, one, f_rs, one, 0.0, rs, st);
    gemm_f64_launch(1, 0, nb, D, D, D, 1.0, vt_s, one, f_irs, one, 0.0, irs, st);
    // mix = Ct^1/2 Cs Ct^1/2 (validated covariances); inner = Cs^1/2 Ct Cs^1/2 (Ct as given)
    gemm_f64_launch(0, 0, nb, D, D, D, 1.0, rt, one, cs_v, one, 0.0, tmp, st);
    gemm_f64_launch(0, 0, nb, D, D, D, 1.0, tmp, one, rt, one, 0.0, mixcat, st);
    gemm_f64_launch(0, 0, nb, D, D, D, 1.0, rs, one, ct, one, 0.0, tmp, st);
    gemm_f64_launch(0, 0, nb, D, D, D, 1.0, tmp, one, rs, one, 0.0, mixcat + mat, st);
    w2t_sym_kernel<<<nb, 256, 0, st>>>(mixcat, D, flags);
    OTVAE_CHECK_LAUNCH("otvae_w2_transport(products)");
    int rc = otvae_eigh_fn(mixcat, 2 * nb, D, 1, roots, evs, eigh_ws, stream);   // both inner square roots side by side
    if (rc) return rc;
    w2_tail_kernel<<<nb, 256, 0, st>>>(ms, mt, cs_v, ct_v, roots, D, w2);
    // T = (1 - pg) Cs^-1/2 inner^1/2 Cs^-1/2 + pg I
    gemm_f64_launch(0, 0, nb, D, D, D, 1.0, irs, one, roots + mat, one, 0.0, tmp, st);
    gemm_f64_launch(0, 0, nb, D, D, D, 1.0 - pg_star, tmp, one, irs, one, 0.0, T, st);
    if (pg_star != 0.0) w2t_diag_add_kernel<<<cdiv((size_t)nb * D, 256), 256, 0, st>>>(T, nb, D, pg_star);
    OTVAE_CHECK_LAUNCH("otvae_w2_transport(tail)");
    return OTVAE_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void apply_transport_kernel(const T* __restrict__ x, const double* __restrict__ ms,
                                                              const double* __restrict__ mt, const double* __restrict__ Tm,
                                                              int B, int D, T* __restrict__ y) {
    extern __shared__ double xc[];  // centred sample, D doubles
    const int b = blockIdx.y, r = blockIdx.x;
    const T* xr = x + ((size_t)b * B + r) * D;
    for (int j = threadIdx.x; j < D; j += 256) xc[j] = (double)xr[j] - ms[(size_t)b * D + j];
    __syncthreads();
    const double* Tb = Tm + (size_t)b * D * D;
    for (int i = threadIdx.x; i < D; i += 256) {
        double s = 0.0;
        for (int j = 0; j < D; ++j) s = fma(Tb[(size_t)i * D + j], xc[j], s);
        y[((size_t)b * B + r) * D + i] = (T)(s + mt[(size_t)b * D + i]);
    }
}

extern "C" int otvae_apply_transport(int dtype, const void* x, const double* ms, const double* mt, const double* T, int nb,
                                     int B, int D, void* y, void* stream) {
    OTVAE_REQUIRE(x && ms && mt && T && y && nb > 0 && B > 0 && D > 0, "otvae_apply_transport: bad argument");
    OTVAE_REQUIRE(dtype == 0 || dtype == 1, "otvae_apply_transport: dtype must be 0 or 1");
    OTVAE_REQUIRE(D <= 4096, "otvae_apply_transport: D too large");
    dim3 grid(B, nb);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0)
        apply_transport_kernel<float><<<grid, 256, D * sizeof(double), st>>>((const float*)x, ms, mt, T, B, D, (float*)y);
    else
        apply_transport_kernel<double><<<grid, 256, D * sizeof(double), st>>>((const double*)x, ms, mt, T, B, D, (double*)y);
    OTVAE_CHECK_LAUNCH("otvae_apply_transport");
    return OTVAE_OK;
}

// ================================================================================================ codebook
// idx = argmax_k 1/(|x - c_k|_2 + 1e-8) / temperature  (softmax is monotone, so the arg-max of the weights);
// first index wins ties like torch.argmax.  One wave per sample, atoms strided over lanes.
__global__ __launch_bounds__(256) void codebook_assign_kernel(const float* __restrict__ x, const float* __restrict__ cb, int B,
                                                              int K, int d, float inv_temp, int64_t* __restrict__ idx,
                                                              float* __restrict__ enc) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.y;
    if (r >= B) return;
    const float* xr = x + ((size_t)b * B + r) * d;
    const float* cbb = cb + (size_t)b * K * d;
    float best = -INFINITY;
    int besti = 0x7fffffff;
    for (int k = lane; k < K; k += 64) {
        float s = 0.f;
        for (int j = 0; j < d; ++j) {
            const float t = xr[j] - cbb[(size_t)k * d + j];
            s = fmaf(t, t, s);
        }
        const float e = (1.f / (sqrtf(s) + 1e-8f)) * inv_temp;
        if (e > best) {
            best = e;
            besti = k;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(besti, o, 64);
        if (ob > best || (ob == best && oi < besti)) {
            best = ob;
            besti = oi;
        }
    }
    if (lane == 0) idx[(size_t)b * B + r] = besti;
    for (int j = lane; j < d; j += 64) enc[((size_t)b * B + r) * d + j] = cbb[(size_t)besti * d + j];
}

extern "C" int otvae_codebook_assign(const float* x, const float* codebook, int nb, int B, int K, int d, float temperature,
                                     int64_t* idx, float* enc, void* stream) {
    OTVAE_REQUIRE(x && codebook && idx && enc && nb > 0 && B > 0 && K > 0 && d > 0, "otvae_codebook_assign: bad argument");
    OTVAE_REQUIRE(temperature > 0.f, "otvae_codebook_assign: temperature must be positive");
    codebook_assign_kernel<<<dim3(cdiv(B, 4), nb), 256, 0, (hipStream_t)stream>>>(x, codebook, B, K, d, 1.f / temperature, idx, enc);
    OTVAE_CHECK_LAUNCH("otvae_codebook_assign");
    return OTVAE_OK;
}

// probs[b][r][k] = softmax_k((1 / (|x_r - c_k|_2 + 1e-8)) / temperature) (MixtureMixin.assign, base.py:216-224) and,
// optionally, its entropy per sample (CodebookPrior 'kl' loss, prior/codebook.py:81-82).  One wave per sample.
__global__ __launch_bounds__(256) void codebook_probs_kernel(const float* __restrict__ x, const float* __restrict__ cb, int B,
                                                             int K, int d, float inv_temp, float* __restrict__ probs,
                                                             float* __restrict__ entropy) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.y;
    if (r >= B) return;
    const float* xr = x + ((size_t)b * B + r) * d;
    const float* cbb = cb + (size_t)b * K * d;
    float* pr = probs + ((size_t)b * B + r) * K;
    float mx = -INFINITY;
    for (int k = lane; k < K; k += 64) {
        float s = 0.f;
        for (int j = 0; j < d; ++j) {
            const float t = xr[j] - cbb[(size_t)k * d + j];
            s = fmaf(t, t, s);
        }
        const float e = (1.f / (sqrtf(s) + 1e-8f)) * inv_temp;
        pr[k] = e;  // energies first, normalised below (same lane re-reads its own stores)
        mx = fmaxf(mx, e);
    }
    mx = wave_max(mx);
    float z = 0.f;
    for (int k = lane; k < K; k += 64) z += __expf(pr[k] - mx);
    z = wave_sum(z);
    const float inv_z = 1.f / z, logz = __logf(z);
    float h = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float t = pr[k] - mx;
        const float p = __expf(t) * inv_z;
        pr[k] = p;
        h -= p * (t - logz);  // -sum p log p with log p = t - log z
    }
    h = wave_sum(h);
    if (entropy != nullptr && lane == 0) entropy[(size_t)b * B + r] = h;
}

extern "C" int otvae_codebook_probs(const float* x, const float* codebook, int nb, int B, int K, int d, float temperature,
                                    float* probs, float* entropy, void* stream) {
    OTVAE_REQUIRE(x && codebook && probs && nb > 0 && B > 0 && K > 0 && d > 0, "otvae_codebook_probs: bad argument");
    OTVAE_REQUIRE(temperature > 0.f, "otvae_codebook_probs: temperature must be positive");
    codebook_probs_kernel<<<dim3(cdiv(B, 4), nb), 256, 0, (hipStream_t)stream>>>(x, codebook, B, K, d, 1.f / temperature, probs,
                                                                               entropy);
    OTVAE_CHECK_LAUNCH("otvae_codebook_probs");
    return OTVAE_OK;
}

// Backward of codebook_probs with respect to the samples (the codebook is a frozen parameter in the reference unless
// update_with_autograd, codebook_model.py:84-86): upstream gradients gp[b][r][k] of the probabilities and gh[b][r] of the
// entropy (either may be NULL).  With e_k = inv_temp / (dist_k + 1e-8), p = softmax(e), H = -sum p log p:
//   g_k   = gp_k - gh (log p_k + 1)
//   de_k  = p_k (g_k - sum_j g_j p_j)
//   gx    = sum_k de_k * (-inv_temp / (dist_k + 1e-8)^2) * (x - c_k) / dist_k
// One wave per sample; the per-atom coefficients of a sample live in LDS (K floats per wave).
__global__ __launch_bounds__(256) void codebook_probs_bwd_kernel(const float* __restrict__ x, const float* __restrict__ cb,
                                                                 const float* __restrict__ probs, const float* __restrict__ gp,
                                                                 const float* __restrict__ gh, int B, int K, int d, float inv_temp,
                                                                 float* __restrict__ gx, float* __restrict__ coef_out) {
    extern __shared__ float coef_lds[];  // [4][K]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wv;
    const int b = blockIdx.y;
    if (r >= B) return;
    float* coef = coef_lds + (size_t)wv * K;
    const size_t row = (size_t)b * B + r;
    const float* xr = x + row * d;
    const float* cbb = cb + (size_t)b * K * d;
    const float* pr = probs + row * K;
    const float ghr = gh ? gh[row] : 0.f;
    float dot = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float p = pr[k];
        float g = gp ? gp[row * K + k] : 0.f;
        if (gh) g -= ghr * (__logf(fmaxf(p, 1e-38f)) + 1.f);
        coef[k] = g;
        dot = fmaf(g, p, dot);
    }
    dot = wave_sum(dot);
    for (int k = lane; k < K; k += 64) {
        float s = 0.f;
        for (int j = 0; j < d; ++j) {
            const float t = xr[j] - cbb[(size_t)k * d + j];
            s = fmaf(t, t, s);
        }
        const float dist = sqrtf(s), den = dist + 1e-8f;
        const float de = pr[k] * (coef[k] - dot);
        coef[k] = dist > 0.f ? de * (-inv_temp / (den * den)) / dist : 0.f;
        if (coef_out) coef_out[row * K + k] = coef[k];  // for the gradient of the atoms (codebook_probs_bwd_atoms_kernel)
    }
    // the wave's LDS writes above are read by other lanes below: LDS operations of one wave execute in order
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int j = 0; j < d; ++j) {
        float acc = 0.f;
        const float xj = xr[j];
        for (int k = lane; k < K; k += 64) acc = fmaf(coef[k], xj - cbb[(size_t)k * d + j], acc);
        acc = wave_sum(acc);
        if (lane == 0) gx[row * d + j] = acc;
    }
}

// gc[k][:] = -sum_r coef[r][k] (x_r - c_k): the same per-(sample, atom) coefficients as gx, summed over the samples in increasing r
// (fixed order).  One block per (atom, problem); thread j owns coordinate j.
__global__ __launch_bounds__(256) void codebook_probs_bwd_atoms_kernel(const float* __restrict__ x, const float* __restrict__ cb,
                                                                       const float* __restrict__ coef, int B, int K, int d,
                                                                       float* __restrict__ gc) {
    const int k = blockIdx.x, b = blockIdx.y;
    const float* xb = x + (size_t)b * B * d;
    const float* cf = coef + (size_t)b * B * K + k;
    for (int j = threadIdx.x; j < d; j += 256) {
        const float ckj = cb[((size_t)b * K + k) * d + j];
        float acc = 0.f;
        for (int r = 0; r < B; ++r) acc = fmaf(cf[(size_t)r * K], ckj - xb[(size_t)r * d + j], acc);
        gc[((size_t)b * K + k) * d + j] = acc;
    }
}

static int codebook_probs_bwd_launch(const float* x, const float* codebook, const float* probs, const float* gprobs, const float* gentropy,
                                     int nb, int B, int K, int d, float temperature, float* gx, float* coef_ws, float* gc, void* stream) {
    OTVAE_REQUIRE(x && codebook && probs && gx && (gprobs || gentropy) && nb > 0 && B > 0 && K > 0 && d > 0,
                  "otvae_codebook_probs_bwd: bad argument");
    OTVAE_REQUIRE(temperature > 0.f, "otvae_codebook_probs_bwd: temperature must be positive");
    OTVAE_REQUIRE((size_t)K * 16 <= 64 * 1024, "otvae_codebook_probs_bwd: K = %d atoms exceed the LDS budget (4096)", K);
    OTVAE_REQUIRE((gc == nullptr) == (coef_ws == nullptr), "otvae_codebook_probs_bwd_atoms: the coefficient workspace goes with gc");
    codebook_probs_bwd_kernel<<<dim3(cdiv(B, 4), nb), 256, (size_t)K * 16, (hipStream_t)stream>>>(x, codebook, probs, gprobs, gentropy,
                                                                                                B, K, d, 1.f / temperature, gx, coef_ws);
    OTVAE_CHECK_LAUNCH("otvae_codebook_probs_bwd");
    if (gc) {
        codebook_probs_bwd_atoms_kernel<<<dim3(K, nb), 256, 0, (hipStream_t)stream>>>(x, codebook, coef_ws, B, K, d, gc);
        OTVAE_CHECK_LAUNCH("otvae_codebook_probs_bwd_atoms");
    }
    return OTVAE_OK;
}

extern "C" int otvae_codebook_probs_bwd(const float* x, const float* codebook, const float* probs, const float* gprobs,
                                        const float* gentropy, int nb, int B, int K, int d, float temperature, float* gx, void* stream) {
    return codebook_probs_bwd_launch(x, codebook, probs, gprobs, gentropy, nb, B, K, d, temperature, gx, nullptr, nullptr, stream);
}

// the same with the gradient of the ATOMS as well (CodebookModel(update_with_autograd=True): the codebook is a trained parameter,
// reference codebook_model.py:89): coef_ws [nb][B][K] floats of workspace, gc [nb][K][d]
extern "C" int otvae_codebook_probs_bwd_atoms(const float* x, const float* codebook, const float* probs, const float* gprobs,
                                              const float* gentropy, int nb, int B, int K, int d, float temperature, float* gx,
                                              float* coef_ws, float* gc, void* stream) {
    OTVAE_REQUIRE(coef_ws && gc, "otvae_codebook_probs_bwd_atoms: NULL workspace / output");
    return codebook_probs_bwd_launch(x, codebook, probs, gprobs, gentropy, nb, B, K, d, temperature, gx, coef_ws, gc, stream);
}

// ---- CodebookModel.energy for the other metrics (reference ot/distribution_models/codebook_model.py:155-168) ------------------------
//   metric 0 ('euclidean', any p > 0):  E = 1 / (cdist_p(x, c) + 1e-8),           cdist_p = (sum_j |x_j - c_j|^p)^(1/p)
//   metric 1 ('cosine'):                E = |x . c| / ((sum_j |x_j|^p)(sum_j |c_j|^p) + 1e-8)^(1/p)
// E [nb][B][K] fp32, no temperature (MixtureMixin.assign divides afterwards: topk masking sits in between, base.py:216-224).
// The p = 2 euclidean energies of the hot path never materialise (otvae_codebook_assign / _probs); this is the general route.
__device__ __forceinline__ float cb_powabs(float t, float p) { return p == 2.f ? t * t : (p == 1.f ? fabsf(t) : __powf(fabsf(t), p)); }

__global__ __launch_bounds__(256) void codebook_energy_kernel(const float* __restrict__ x, const float* __restrict__ cb, int B, int K,
                                                              int d, int metric, float p, float* __restrict__ E) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * K) return;
    const int r = i / K, k = i - r * K;
    const float* xr = x + ((size_t)b * B + r) * d;
    const float* ck = cb + ((size_t)b * K + k) * d;
    float e;
    if (metric == 0) {
        float s = 0.f;
        for (int j = 0; j < d; ++j) s += cb_powabs(xr[j] - ck[j], p);
        const float dist = p == 2.f ? sqrtf(s) : (p == 1.f ? s : __powf(s, 1.f / p));
        e = 1.f / (dist + 1e-8f);
    } else {
        float nx = 0.f, nc = 0.f, dot = 0.f;
        for (int j = 0; j < d; ++j) {
            nx += cb_powabs(xr[j], p);
            nc += cb_powabs(ck[j], p);
            dot = fmaf(xr[j], ck[j], dot);
        }
        e = fabsf(dot) / __powf(nx * nc + 1e-8f, 1.f / p);
    }
    E[((size_t)b * B + r) * K + k] = e;
}

// d E[r][k] / d x_r[j] (which = 0) or / d c_k[j] (which = 1), the factor shared by the two backward kernels.  Conventions at the
// kinks follow torch's backward formulas: cdist gives 0 where the distance is 0 (or, for p < 1, where the coordinate difference
// is 0); |t|^p differentiates to 0 at t = 0 (torch: sign(0) = 0).
__device__ __forceinline__ float cb_dabs_pow(float t, float p) {  // d |t|^p / d t
    if (t == 0.f) return 0.f;
    const float a = fabsf(t);
    const float m = p == 2.f ? 2.f * a : (p == 1.f ? 1.f : p * __powf(a, p - 1.f));
    return t > 0.f ? m : -m;
}

struct CbPair {  // what depends on the (sample, atom) pair only
    float a, b, c;
};
__device__ __forceinline__ CbPair cb_pair(const float* __restrict__ xr, const float* __restrict__ ck, int d, int metric, float p) {
    CbPair o = {0.f, 0.f, 0.f};
    if (metric == 0) {
        float s = 0.f;
        for (int j = 0; j < d; ++j) s += cb_powabs(xr[j] - ck[j], p);
        const float dist = p == 2.f ? sqrtf(s) : (p == 1.f ? s : __powf(s, 1.f / p));
        // dE/d dist = -1 / (dist + eps)^2;  d dist / d diff_j = (1 / p) s^(1/p - 1) d|diff_j|^p
        o.a = dist > 0.f ? -1.f / ((dist + 1e-8f) * (dist + 1e-8f)) * (1.f / p) * (p == 1.f ? 1.f : __powf(s, 1.f / p - 1.f)) : 0.f;
    } else {
        float nx = 0.f, nc = 0.f, dot = 0.f;
        for (int j = 0; j < d; ++j) {
            nx += cb_powabs(xr[j], p);
            nc += cb_powabs(ck[j], p);
            dot = fmaf(xr[j], ck[j], dot);
        }
        const float D = nx * nc + 1e-8f, Dp = __powf(D, -1.f / p);
        o.a = (dot > 0.f ? 1.f : (dot < 0.f ? -1.f : 0.f)) * Dp;   // d E / d dot
        const float t = -fabsf(dot) * (1.f / p) * Dp / D;           // d E / d D
        o.b = t * nc;                                               // d E / d nx
        o.c = t * nx;                                               // d E / d nc
    }
    return o;
}

__global__ __launch_bounds__(256) void codebook_energy_bwd_x_kernel(const float* __restrict__ x, const float* __restrict__ cb,
                                                                    const float* __restrict__ gE, int B, int K, int d, int metric,
                                                                    float p, float* __restrict__ gx) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
    if (r >= B) return;
    const float* xr = x + ((size_t)b * B + r) * d;
    for (int j = 0; j < d; ++j) {
        float acc = 0.f;
        for (int k = lane; k < K; k += 64) {   // fixed order: lanes stride over the atoms, then the wave tree
            const float* ck = cb + ((size_t)b * K + k) * d;
            const CbPair q = cb_pair(xr, ck, d, metric, p);
            const float g = gE[((size_t)b * B + r) * K + k];
            float v;
            if (metric == 0) {
                const float diff = xr[j] - ck[j];
                v = (p < 1.f && diff == 0.f) ? 0.f : q.a * cb_dabs_pow(diff, p);
            } else {
                v = q.a * ck[j] + q.b * cb_dabs_pow(xr[j], p);
            }
            acc = fmaf(g, v, acc);
        }
        acc = wave_sum(acc);
        if (lane == 0) gx[((size_t)b * B + r) * d + j] = acc;
    }
}

__global__ __launch_bounds__(256) void codebook_energy_bwd_c_kernel(const float* __restrict__ x, const float* __restrict__ cb,
                                                                    const float* __restrict__ gE, int B, int K, int d, int metric,
                                                                    float p, float* __restrict__ gc) {
    const int k = blockIdx.x, b = blockIdx.y;
    const float* ck = cb + ((size_t)b * K + k) * d;
    for (int j = threadIdx.x; j < d; j += 256) {
        float acc = 0.f;
        for (int r = 0; r < B; ++r) {          // samples in increasing order (fixed)
            const float* xr = x + ((size_t)b * B + r) * d;
            const CbPair q = cb_pair(xr, ck, d, metric, p);
            const float g = gE[((size_t)b * B + r) * K + k];
            float v;
            if (metric == 0) {
                const float diff = xr[j] - ck[j];
                v = (p < 1.f && diff == 0.f) ? 0.f : -q.a * cb_dabs_pow(diff, p);
            } else {
                v = q.a * xr[j] + q.c * cb_dabs_pow(ck[j], p);
            }
            acc = fmaf(g, v, acc);
        }
        gc[((size_t)b * K + k) * d + j] = acc;
    }
}

static int codebook_energy_check(const char* who, const void* a, const void* b, const void* c, int nb, int B, int K, int d, int metric,
                                 float p) {
    OTVAE_REQUIRE(a && b && c && nb > 0 && B > 0 && K > 0 && d > 0, "%s: bad argument", who);
    OTVAE_REQUIRE((metric == 0 || metric == 1) && p > 0.f, "%s: metric %d (0 euclidean, 1 cosine), p = %g", who, metric, (double)p);
    return OTVAE_OK;
}

extern "C" int otvae_codebook_energy(const float* x, const float* codebook, int nb, int B, int K, int d, int metric, float p, float* E,
                                     void* stream) {
    if (int rc = codebook_energy_check("otvae_codebook_energy", x, codebook, E, nb, B, K, d, metric, p)) return rc;
    codebook_energy_kernel<<<dim3(cdiv((int64_t)B * K, 256), nb), 256, 0, (hipStream_t)stream>>>(x, codebook, B, K, d, metric, p, E);
    OTVAE_CHECK_LAUNCH("otvae_codebook_energy");
    return OTVAE_OK;
}

// gx [nb][B][d] and / or gc [nb][K][d] (either may be NULL) from gE [nb][B][K]
extern "C" int otvae_codebook_energy_bwd(const float* x, const float* codebook, const float* gE, int nb, int B, int K, int d, int metric,
                                         float p, float* gx, float* gc, void* stream) {
    if (int rc = codebook_energy_check("otvae_codebook_energy_bwd", x, codebook, gE, nb, B, K, d, metric, p)) return rc;
    OTVAE_REQUIRE(gx || gc, "otvae_codebook_energy_bwd: nothing to compute");
    if (gx) {
        codebook_energy_bwd_x_kernel<<<dim3(cdiv(B, 4), nb), 256, 0, (hipStream_t)stream>>>(x, codebook, gE, B, K, d, metric, p, gx);
        OTVAE_CHECK_LAUNCH("otvae_codebook_energy_bwd(samples)");
    }
    if (gc) {
        codebook_energy_bwd_c_kernel<<<dim3(K, nb), 256, 0, (hipStream_t)stream>>>(x, codebook, gE, B, K, d, metric, p, gc);
        OTVAE_CHECK_LAUNCH("otvae_codebook_energy_bwd(atoms)");
    }
    return OTVAE_OK;
}

// One-hot k-means accumulation (MixtureMixin.kmean_iteration with 'argmax' weights, base.py:241-252):
// counts[b][k] = #{r : idx_r = k}, sums[b][k][:] = sum_{r : idx_r = k} x_r, members added in increasing r (fixed order).
// One block per (atom, problem): the block scans the index vector once, then its threads own the d coordinates.
__global__ __launch_bounds__(256) void codebook_kmeans_kernel(const float* __restrict__ x, const int64_t* __restrict__ idx, int B,
                                                              int K, int d, float* __restrict__ counts, float* __restrict__ sums) {
    extern __shared__ int members[];  // rows assigned to this atom, ascending
    __shared__ int s_n;
    const int k = blockIdx.x, b = blockIdx.y;
    const int64_t* ib = idx + (size_t)b * B;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    // ordered compaction, 256 rows per round: ballot per wave, wave offsets through LDS
    __shared__ int wcount[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r0 = 0; r0 < B; r0 += 256) {
        const int r = r0 + threadIdx.x;
        const bool hit = r < B && ib[r] == (int64_t)k;
        const unsigned long long bal = __ballot(hit);
        if (lane == 0) wcount[wave] = __popcll(bal);
        __syncthreads();
        int base = s_n;
        for (int w = 0; w < wave; ++w) base += wcount[w];
        if (hit) members[base + __popcll(bal & ((1ull << lane) - 1ull))] = r;
        __syncthreads();
        if (threadIdx.x == 0) s_n += wcount[0] + wcount[1] + wcount[2] + wcount[3];
        __syncthreads();
    }
    const int n = s_n;
    if (threadIdx.x == 0) counts[(size_t)b * K + k] = (float)n;
    const float* xb = x + (size_t)b * B * d;
    for (int j = threadIdx.x; j < d; j += 256) {
        float s = 0.f;
        for (int m = 0; m < n; ++m) s += xb[(size_t)members[m] * d + j];
        sums[((size_t)b * K + k) * d + j] = s;
    }
}

extern "C" int otvae_codebook_kmeans(const float* x, const int64_t* idx, int nb, int B, int K, int d, float* counts, float* sums,
                                     void* stream) {
    OTVAE_REQUIRE(x && idx && counts && sums && nb > 0 && B > 0 && K > 0 && d > 0, "otvae_codebook_kmeans: bad argument");
    OTVAE_REQUIRE((size_t)B * sizeof(int) <= 60 * 1024, "otvae_codebook_kmeans: more than 15360 samples per call unsupported");
    codebook_kmeans_kernel<<<dim3(K, nb), 256, (size_t)B * sizeof(int), (hipStream_t)stream>>>(x, idx, B, K, d, counts, sums);
    OTVAE_CHECK_LAUNCH("otvae_codebook_kmeans");
    return OTVAE_OK;
}


// ---- Gaussian-mixture energy (reference ot/distribution_models/gassian_mixture_model.py:91-99, diagonal covariances) ----
// energy[nb][B][K] = log N(x_b; mean_k, diag var_k) + log w_k
//                  = -1/2 sum_c ((x_bc - mean_kc)^2 / var_kc + log var_kc) - d/2 log(2 pi) + log w_k
// One lane per (sample, component); a block's samples and one component's parameters stay in L1/L2.
template <typename T>
__global__ __launch_bounds__(256) void gmm_diag_energy_kernel(const T* __restrict__ x, const T* __restrict__ mean,
                                                              const T* __restrict__ var, const T* __restrict__ logw, int B, int K,
                                                              int d, T* __restrict__ out) {
    const int nbi = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * K) return;
    const int b = i / K, k = i - b * K;
    const T* xp = x + ((size_t)nbi * B + b) * d;
    const T* mp = mean + ((size_t)nbi * K + k) * d;
    const T* vp = var + ((size_t)nbi * K + k) * d;
    T acc = (T)0;
    for (int c = 0; c < d; ++c) {
        const T diff = xp[c] - mp[c];
        acc += diff * diff / vp[c] + log(vp[c]);
    }
    out[((size_t)nbi * B + b) * K + k] = (T)-0.5 * (acc + (T)d * (T)1.8378770664093454835606594728112) + logw[(size_t)nbi * K + k];
}

extern "C" int otvae_gmm_diag_energy(int dtype, const void* x, const void* mean, const void* var, const void* logw, int nb,
                                     int B, int K, int d, void* energy, void* stream) {
    OTVAE_REQUIRE(x && mean && var && logw && energy && nb > 0 && B > 0 && K > 0 && d > 0, "otvae_gmm_diag_energy: bad argument");
    OTVAE_REQUIRE(dtype == 0 || dtype == 1, "otvae_gmm_diag_energy: dtype must be 0 (fp32) or 1 (fp64)");
    const dim3 grid(cdiv((int64_t)B * K, 256), nb);
    if (dtype == 0)
        gmm_diag_energy_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>((const float*)x, (const float*)mean, (const float*)var,
                                                                         (const float*)logw, B, K, d, (float*)energy);
    else
        gmm_diag_energy_kernel<double><<<grid, 256, 0, (hipStream_t)stream>>>((const double*)x, (const double*)mean,
                                                                          (const double*)var, (const double*)logw, B, K, d,
                                                                          (double*)energy);
    OTVAE_CHECK_LAUNCH("otvae_gmm_diag_energy");
    return OTVAE_OK;
}
