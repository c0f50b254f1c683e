// Symmetric eigensolver for D <= 128 (the latent sizes of the CNN configurations): one-sided (Hestenes) Jacobi.
// Replaces torch.linalg.eigh inside the reference's sqrtm / invsqrtm / min_eig (ot/matrix_utils.py:37-46,91-109).
//
// Why one-sided: the two-sided form (gaussian_ot.hip: eigh_kernel) rotates rows AND columns of A in LDS and the rows of V^T in
// global memory every step -- three barriers and a dependent global round trip per step, 8.6 us per step, 10 ms per
// 128 x 128 matrix.  Here G = A V is the only matrix the iteration touches, one barrier per step, everything on chip:
//   * odd-even ordering with unconditional swaps: the n = 2 ceil(D/2) columns sit at positions 0..n-1; even steps orthogonalise
//     the position pairs (0,1)(2,3)..., odd steps (1,2)(3,4)...; after its rotation a pair's two columns trade places, so
//     that n steps reverse the order and every two columns have met exactly once (a sweep);
//   * a group of L lanes owns pair slot i and KEEPS the column at position 2i+1 in registers across steps; the other column
//     of its pair comes from LDS and one result goes back to LDS for the neighbouring slot: one column read + one written
//     per slot and step -- half the LDS traffic of a scheme that keeps both columns in LDS, and LDS bandwidth is what
//     bounds the iteration (fp64, 128 KB per pass at D = 128);
//   * the rotations are NOT applied to V inside the loop: they are logged (16 bytes per pair and step) and a second kernel
//     replays the log on the rows of V = I, which are independent of each other -- D waves, each with its row in LDS, the
//     64 lanes take the 64 disjoint pairs of a step;
//   * eigenvalues are lambda_k = v_k . g_k (signed: indefinite and singular matrices are fine), f(A) = V f(Lambda) V^T is
//     one product.  A matrix with a negative diagonal entry (certainly indefinite) is solved as A + |A|_inf I and the shift
//     taken off the eigenvalues: one-sided Jacobi sees A^2, in which +lambda and -lambda of equal size are a double eigenvalue.
//     An indefinite matrix whose diagonal is non-negative (A = [[0,1],[1,0]], a bipartite adjacency matrix) is caught AFTER the
//     iteration: within such a +-lambda pair the columns of G are orthogonal for ANY rotation of the two eigenvectors, so the
//     iteration may stop at mixtures v with |v . A v| < |A v|.  hj_finish_kernel tests every pair for |v_k . g_k| = |g_k| and,
//     when one fails, the three kernels run a second time on A + |A|_inf I (pass 1; they return at once otherwise).
//   * round 3: a positive definite A (the covariances this solver exists for) is first FACTORED in place in LDS, A = U^T U (upper
//     Cholesky, right-looking, ~80 us at D = 128), and the iteration runs on the columns of L = U^T -- position j starts as row j of
//     U.  One-sided Jacobi on L computes L V = Q S, so A = L L^T = Q S^2 Q^T: eigenvalues are the squared column norms and the
//     eigenvectors are the NORMALISED FINAL COLUMNS themselves -- no rotation replay on V at all -- and the iteration sees
//     cond(A)^1/2 instead of cond(A)^2: 9 sweeps instead of 16 on covariance-like 128 x 128 matrices (numpy restatement of this
//     ordering; columns of L^T would need 11).  A non-positive pivot (semi-definite or indefinite input) puts A back and takes the
//     path above.  OTVAE_EIGH_NO_CHOL=1 switches the factorisation off (A/B).
// Arithmetic is fp64 throughout; the iteration ends after the first sweep without a rotation.  A matrix that is still rotating
// after HJ_MAX_SWEEPS sweeps gets NaN eigenvalues (loud, like a starved Sinkhorn solve) instead of an unconverged answer.
#include <type_traits>
#include "common.h"
#include <mutex>

void gemm_f64_launch(int transA, int transB, int nb, int m, int n, int k, double alpha, const double* A, size_t sA, const double* B,
                     size_t sB, double beta, double* C, hipStream_t st);  // gaussian_ot.hip
int cholesky_blocked(const double* A, int nb, int D, double* L, size_t ls, int* info, size_t is, hipStream_t st);  // gaussian_ot.hip

#define HJ_MAX_SWEEPS 24
// A pair is rotated whenever |g_p . g_q| > 1e-15 |g_p| |g_q| (every rotation refines), but only a pair whose cosine is above
// 1e-8 BEFORE its rotation keeps the iteration going.  Jacobi converges quadratically: a sweep that met no cosine above
// 1e-8 has rotated all of them away and leaves ~1e-16 behind, so it is the last one -- a level of 1e-13 (round 2's first
// setting) only confirmed that with one more, quiet sweep (measured on a numpy restatement of this ordering: one sweep fewer on
// covariance-like, graded, clustered and rank-deficient spectra alike, the same final cosines of 1e-15).  The computed inner
// product of two 128-vectors carries rounding noise of up to D eps = 3e-14 of |g_p| |g_q|: a stop level down there is
// re-triggered sweep after sweep and the solver never sees a quiet sweep.
#define HJ_TOL2 1e-30
#define HJ_STOP2 1e-16

struct HjCtl {
    int steps;   // steps whose rotations are in the log
    int sweeps;
    double shift;  // added to the diagonal before the iteration (0 unless some diagonal entry was negative, or pass 1)
    int redo;         // written by hj_finish_kernel (pass 0): an eigenpair failed |v . g| = |g|: pass 1 runs on the shifted matrix
    int unconverged;  // the last sweep of the budget still rotated above the stop level
    int chol;         // the iteration ran on the Cholesky factor's columns: eigenvalue = |g|^2, eigenvector = g / |g|
};

static __host__ __device__ inline size_t hj_per_matrix(int D) {
    const size_t n = (size_t)((D + 1) & ~1), half = n / 2;
    return ((4 * n * n * 8 + (size_t)HJ_MAX_SWEEPS * n * half * 16 + 256) + 255) & ~(size_t)255;
}

extern "C" int64_t otvae_eigh_onesided_ws(int nb, int D) {
    if (nb <= 0 || D <= 0) return -1;
    return (int64_t)nb * (int64_t)hj_per_matrix(D);
}

struct HjWs {
    double *G, *V, *W, *T;  // G[pos][D] (n x D), V[i][pos] and W[i][pos] (D x n), T[k][D]
    double2* log;
    HjCtl* ctl;
};
__host__ __device__ static inline HjWs hj_ws(void* ws, int b, int D) {
    const size_t n = (size_t)((D + 1) & ~1), half = n / 2;
    char* w = (char*)ws + (size_t)b * hj_per_matrix(D);
    HjWs r;
    r.G = (double*)w;
    r.V = r.G + n * n;
    r.W = r.V + n * n;
    r.T = r.W + n * n;
    r.log = (double2*)(r.T + n * n);
    r.ctl = (HjCtl*)((char*)r.log + (size_t)HJ_MAX_SWEEPS * n * half * 16);
    return r;
}

// all-reduce over the L (8 or 16) consecutive lanes that share a column pair, on the DPP network (no LDS round trip):
// xor 1 and xor 2 inside a quad, then the mirrored half-row (lane i <-> 7 - i: the other quad), then the mirrored row
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xF, 0xF, false);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo));
}
template <int L>
__device__ __forceinline__ double group_sum(double v) {
    v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);  // row_half_mirror
    if (L == 16) v += dpp_mov<0x140>(v);  // row_mirror
    return v;
}

// all-reduce over the whole wave without the LDS crossbar (the 64-bit __shfl_xor tree is twelve ds_bpermute round trips):
// the four DPP stages leave every lane of a 16-lane row with its row's sum, four v_readlane pairs fetch the row sums
__device__ __forceinline__ double wave_allsum_dpp(double v) {
    v = group_sum<16>(v);
    double tot = 0.0;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const unsigned long long u = (unsigned long long)__double_as_longlong(v);
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, 16 * rr);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), 16 * rr);
        tot += __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    }
    return tot;
}

// 1/x and 1/sqrt(x) from the hardware seeds + two Newton steps each (the compiler's IEEE division / square root expand to
// ~30 instructions apiece; a rotation angle need not be exact -- only c^2 + s^2 = 1 must hold to rounding, and it does
// because s = c t and c = rsqrt(1 + t^2) is refined to full precision)
__device__ __forceinline__ double nr_rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = fma(fma(-x, y, 1.0), y, y);
    return fma(fma(-x, y, 1.0), y, y);
}
__device__ __forceinline__ double nr_rsq(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    y = y * fma(-h * y, y, 1.5);
    return y * fma(-h * y, y, 1.5);
}

// one workgroup of 512 threads per matrix: L lanes per pair slot (8 for D > 64: 64 slots; 16 below), rows strided by L
template <int L>
__global__ __launch_bounds__(512) void hj_sweep_kernel(const double* __restrict__ Ain, int D, void* __restrict__ ws,
                                                        const double* __restrict__ G0in, const int* __restrict__ warm, int pass,
                                                        int use_chol) {
    if (pass == 1 && hj_ws(ws, blockIdx.x, D).ctl->redo == 0) return;  // uniform over the workgroup
    // the start basis counts only once the caller's flag says it holds one; the second pass starts cold on the shifted matrix
    const double* __restrict__ G0 = (pass == 0 && G0in && (!warm || warm[0])) ? G0in : nullptr;
    extern __shared__ __align__(16) double hj_lds[];
    __shared__ int s_rot;
    __shared__ double s_red[8], s_shift;
    __shared__ double s_nrm[130];  // squared norms of the columns in the even positions (the kept columns' live in registers)
    constexpr int R = 128 / L;  // rows per lane at most (D <= 128 for L = 8; D <= 64 for L = 16 uses 4 of its 8)
    const int n = (D + 1) & ~1, half = n / 2, LD = D + 1;
    const HjWs w = hj_ws(ws, blockIdx.x, D);
    const double* Ab = Ain + (size_t)blockIdx.x * D * D;
    double* G = hj_lds;  // published columns by POSITION: G[pos * LD + row]
    const int tid = threadIdx.x, grp = tid / L, r = tid % L;
    // Warm start (G0 != NULL): position j starts as column j of A V0 for an orthonormal V0 handed over by the caller (the
    // eigenvectors of a nearby matrix: the previous training step's covariance), G0[j][:] = A v_j; hj_vectors_kernel then starts
    // from V0 instead of the identity.  Nearly orthogonal columns converge in 2-4 sweeps instead of ~9.
    for (int e = tid; e < n * D; e += 512) {
        const int j = e / D, i = e - j * D;  // position j (= column j at the start), row i; lower triangle, like eigh(UPLO='L')
        double v = 0.0;                      // odd D: a zero dummy column
        if (j < D) v = G0 ? G0[(size_t)blockIdx.x * D * D + e] : ((i >= j) ? Ab[(size_t)i * D + j] : Ab[(size_t)j * D + i]);
        G[j * LD + i] = v;
    }
    if (tid == 0) {
        s_rot = 0;
        s_shift = 0.0;
    }
    __syncthreads();
    // certainly indefinite (a negative diagonal entry): shift by the infinity norm, so that the spectrum becomes non-negative
    if (!G0) {
        double mind = INFINITY, rsum = 0.0;
        for (int i = tid; i < D; i += 512) {
            mind = fmin(mind, G[i * LD + i]);
            double a1 = 0.0;
            for (int j = 0; j < D; ++j) a1 += fabs(G[j * LD + i]);
            rsum = fmax(rsum, a1);
        }
        mind = wave_min(mind);
        rsum = wave_max(rsum);
        if ((tid & 63) == 0) s_red[tid >> 6] = mind;
        __syncthreads();
        double m8 = s_red[0];
        for (int q8 = 1; q8 < 8; ++q8) m8 = fmin(m8, s_red[q8]);
        __syncthreads();
        if ((tid & 63) == 0) s_red[tid >> 6] = rsum;
        __syncthreads();
        if (tid == 0) {
            double r8 = s_red[0];
            for (int q8 = 1; q8 < 8; ++q8) r8 = fmax(r8, s_red[q8]);
            s_shift = (m8 < 0.0 || pass == 1) ? r8 : 0.0;
        }
        __syncthreads();
        const double sh = s_shift;
        if (sh != 0.0)
            for (int i = tid; i < D; i += 512) G[i * LD + i] += sh;
        __syncthreads();
    }
    // ---- positive definite input: factor in place, A = U^T U, and iterate on the columns of L = U^T (see the header).  M[r][c] =
    // G[r * LD + c] is row-major A; afterwards position j holds row j of U = column j of L, zeros below its diagonal entry.
    __shared__ int s_chol;
    if (tid == 0) s_chol = 0;
    __syncthreads();
    if (use_chol && !G0 && pass == 0 && s_shift == 0.0) {
        bool ok = true;
        for (int j = 0; j < D; ++j) {
            const double piv = G[j * LD + j];   // every thread reads the same value (behind the barrier of the step before)
            if (!(piv > 0.0) || !(piv < INFINITY)) {
                ok = false;
                break;                          // uniform
            }
            const double ujj = sqrt(piv), inv = 1.0 / ujj;
            __syncthreads();                    // all have read the pivot before it is overwritten
            for (int k = j + tid; k < D; k += 512) G[j * LD + k] = (k == j) ? ujj : G[j * LD + k] * inv;   // row j of U
            __syncthreads();
            // trailing upper triangle: M[i][k] -= U[j][i] U[j][k], j < i <= k
            // (16 rows x 32 columns of threads: no integer division; LD is odd, so the two rows of a wave hit different banks)
            const double* uj = G + j * LD;
            for (int a = j + 1 + (tid >> 5); a < D; a += 16) {
                const double ua = uj[a];
                double* ma = G + a * LD;
                for (int b = a + (tid & 31); b < D; b += 32) ma[b] = fma(-ua, uj[b], ma[b]);
            }
            __syncthreads();
        }
        if (ok) {
            for (int e = tid; e < D * D; e += 512) {  // zeros below the diagonal of U (the strict lower part still holds A)
                const int rr = e / D, cc = e - rr * D;
                if (cc < rr) G[rr * LD + cc] = 0.0;
            }
            if (tid == 0) s_chol = 1;
        } else {
            for (int e = tid; e < D * D; e += 512) {  // not positive definite: A again (lower triangle, like eigh(UPLO='L'))
                const int j = e / D, i = e - j * D;
                G[j * LD + i] = (i >= j) ? Ab[(size_t)i * D + j] : Ab[(size_t)j * D + i];
            }
        }
        __syncthreads();
    }
    const int nrow = (D + L - 1) / L;
    const bool active = grp < half;
    const int kpos = 2 * grp + 1;  // the position whose column this slot keeps in registers
    double keep[R];
#pragma unroll
    for (int u = 0; u < R; ++u) {
        const int i = r + L * u;
        keep[u] = (active && u < nrow && i < D) ? G[kpos * LD + i] : 0.0;
    }
    int gstep = 0, sweep = 0;
    // squared norms are carried from step to step (|p'|^2 = |p|^2 - t gamma, |q'|^2 = |q|^2 + t gamma) and recounted at the start of
    // every sweep: a step needs ONE dot product (p . q) instead of three
    double nk = 0.0;
    auto recount_norms = [&]() {
        if (active) {
            double a = 0.0, e2 = 0.0;
            const double* ge = G + (kpos - 1) * LD;
#pragma unroll
            for (int u = 0; u < R; ++u) {
                const int i = r + L * u;
                const double ev = (u < nrow && i < D) ? ge[i] : 0.0;
                a = fma(keep[u], keep[u], a);
                e2 = fma(ev, ev, e2);
            }
            nk = group_sum<L>(a);
            e2 = group_sum<L>(e2);
            if (r == 0) s_nrm[kpos - 1] = e2;
        }
        __syncthreads();
    };
    // one step; ODD is a compile-time constant so that the roles of the kept / fetched column cost no selects:
    // even step: pair (2g, 2g+1): the kept column is the UPPER one (q), the lower (p) comes from LDS position 2g;
    // odd step:  pair (2g+1, 2g+2): the kept column is the LOWER one (p), the upper (q) comes from LDS position 2g+2.
    // p' = c p - s q, q' = s p + c q, then the two trade places: position j <- q', position j + 1 <- p'.
    // even: the kept position 2g+1 = j+1 takes p', q' is published at 2g;  odd: the kept position 2g+1 = j takes q', p' at 2g+2.
    auto step = [&](auto odd_tag) {
        constexpr bool ODD = decltype(odd_tag)::value;
        const int opos = ODD ? kpos + 1 : kpos - 1;
        if (active && opos < n) {
            double other[R], gamma = 0.0;
            double* go = G + opos * LD;
#pragma unroll
            for (int u = 0; u < R; ++u) {
                const int i = r + L * u;
                other[u] = (u < nrow && i < D) ? go[i] : 0.0;
                gamma = fma(keep[u], other[u], gamma);
            }
            gamma = group_sum<L>(gamma);
            const double no = s_nrm[opos];
            const double alpha = ODD ? nk : no, beta = ODD ? no : nk;  // |p|^2 (lower position), |q|^2 (upper)
            double c = 1.0, s = 0.0, tg = 0.0;
            const double ab = alpha * beta;
            if (gamma * gamma > HJ_TOL2 * ab && ab > 1e-280) {
                const double zeta = (beta - alpha) * 0.5 * nr_rcp(gamma);
                const double az = fabs(zeta);
                double tt;
                if (az > 1e8) {
                    tt = 0.5 * nr_rcp(zeta);  // |zeta| + sqrt(1 + zeta^2) = 2 |zeta| to rounding
                } else {
                    const double q2 = fma(zeta, zeta, 1.0);
                    tt = nr_rcp(az + q2 * nr_rsq(q2));
                    tt = zeta >= 0.0 ? tt : -tt;
                }
                c = nr_rsq(fma(tt, tt, 1.0));
                s = c * tt;
                tg = tt * gamma;
                if (r == 0 && gamma * gamma > HJ_STOP2 * ab) s_rot = 1;  // benign race: every writer stores 1
            }
            // p' shrinks to |p|^2 - t gamma: recounted when the subtraction cancelled more than two digits (a column on its way to
            // zero in a rank-deficient matrix); p' is the published column in an odd step, the kept one in an even step
            const double n_p = alpha - tg, n_q = beta + tg;
            const bool recount = !(n_p > 0.01 * alpha) && tg != 0.0;  // uniform over the slot's lanes
            double n_pub = ODD ? n_p : n_q, acc2 = 0.0;
            nk = ODD ? n_q : n_p;
#pragma unroll
            for (int u = 0; u < R; ++u) {
                const int i = r + L * u;
                double pub;
                if (ODD) {  // p = keep, q = other
                    pub = c * keep[u] - s * other[u];      // p' -> position 2g+2
                    keep[u] = s * keep[u] + c * other[u];  // q' -> kept position
                } else {    // p = other, q = keep
                    pub = s * other[u] + c * keep[u];      // q' -> position 2g
                    keep[u] = c * other[u] - s * keep[u];  // p' -> kept position
                }
                if (u < nrow && i < D) go[i] = pub;
                if (recount) acc2 = ODD ? fma(pub, pub, acc2) : fma(keep[u], keep[u], acc2);
            }
            if (recount) {
                acc2 = group_sum<L>(acc2);
                if (ODD) n_pub = acc2;
                else nk = acc2;
            }
            if (r == 0) {
                s_nrm[opos] = n_pub;
                w.log[(size_t)gstep * half + grp] = make_double2(c, s);
            }
        }
        __syncthreads();
        ++gstep;
    };
    bool quiet = false;
    for (; sweep < HJ_MAX_SWEEPS; ++sweep) {
        recount_norms();
        for (int t = 0; t < n; t += 2) {
            step(std::false_type{});
            step(std::true_type{});
        }
        const int rotated = s_rot;  // read by every thread after the step's barrier
        __syncthreads();
        if (!rotated) {
            ++sweep;
            quiet = true;
            break;
        }
        if (tid == 0) s_rot = 0;
        __syncthreads();
    }
    // odd positions live in the slots' registers, even positions in LDS
    if (active) {
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int i = r + L * u;
            if (u < nrow && i < D) w.G[(size_t)kpos * D + i] = keep[u];
        }
    }
    for (int e = tid; e < half * D; e += 512) {
        const int k = e / D, i = e - k * D;
        w.G[(size_t)(2 * k) * D + i] = G[(2 * k) * LD + i];
    }
    if (tid == 0) {
        w.ctl->steps = gstep;
        w.ctl->sweeps = sweep;
        w.ctl->shift = s_shift;
        w.ctl->unconverged = quiet ? 0 : 1;
        w.ctl->chol = s_chol;
        if (pass == 0) w.ctl->redo = 0;
    }
}

// V = J_1 J_2 ... applied to the rows of the identity: a wave per row, its row (by position) in LDS, lane k replays slot k:
// the same rotation and the same exchange of places as the columns of G underwent
#define HJV_CHUNK 32  // steps of the rotation log staged in LDS at a time (32 KB)
// lane k of a wave holds the row's entries (2k, 2k + 1) in registers.  An even step rotates exactly such a pair: no communication.
// An odd step rotates (2k + 1, 2k + 2): lane k needs the first entry of lane k + 1 for its new second entry, lane k + 1 the second
// entry of lane k (and slot k's rotation) for its new first entry -- two whole-wave DPP shifts by one lane (gfx9 wave_shl1 /
// wave_shr1), no LDS round trip: 0.28 -> ~0.1 ms per 128 x 128 decomposition against the version that kept the row in LDS.
__device__ __forceinline__ double wave_take_next(double v) {  // lane k <- lane k + 1 (lane 63: 0)
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)u, 0x130, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), 0x130, 0xF, 0xF, true);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo));
}
__device__ __forceinline__ double wave_take_prev(double v) {  // lane k <- lane k - 1 (lane 0: 0)
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)u, 0x138, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), 0x138, 0xF, 0xF, true);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo));
}

__global__ __launch_bounds__(256) void hj_vectors_kernel(int D, void* __restrict__ ws, const double* __restrict__ Vin, const int* __restrict__ warm,
                                                         int pass) {
    if (pass == 1 && hj_ws(ws, blockIdx.y, D).ctl->redo == 0) return;
    const double* __restrict__ Vinit = (pass == 0 && Vin && (!warm || warm[0])) ? Vin : nullptr;
    __shared__ double2 slog[HJV_CHUNK][64];
    const int n = (D + 1) & ~1, half = n / 2;
    const HjWs w = hj_ws(ws, blockIdx.y, D);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wv;
    // row i of V0: the identity, or component i of the caller's start vectors (Vinit[k][:] = vector of position k)
    auto v0 = [&](int k) -> double {
        if (k >= n) return 0.0;
        if (Vinit && k < D && i < D) return Vinit[((size_t)blockIdx.y * D + k) * D + i];
        return k == i ? 1.0 : 0.0;
    };
    double a = v0(2 * lane), b = v0(2 * lane + 1);
    const bool slot = lane < half;
    if (w.ctl->chol) {
        // the iteration ran on the Cholesky factor's columns: the eigenvectors are the final columns themselves (no replay);
        // hj_finish_kernel normalises them
        if (i < D && slot) {
            const int k0 = 2 * lane;
            w.V[(size_t)i * n + k0] = w.G[(size_t)k0 * D + i];
            w.V[(size_t)i * n + k0 + 1] = w.G[(size_t)(k0 + 1) * D + i];
            w.W[(size_t)i * n + k0] = 0.0;
            w.W[(size_t)i * n + k0 + 1] = 0.0;
        }
        return;
    }
    const int steps = w.ctl->steps;
    for (int st0 = 0; st0 < steps; st0 += HJV_CHUNK) {
        __syncthreads();  // the previous chunk is consumed
        const int cnt = min(HJV_CHUNK, steps - st0);
        for (int e = threadIdx.x; e < cnt * half; e += 256) slog[e / half][e % half] = w.log[(size_t)st0 * half + e];
        __syncthreads();
        for (int s_ = 0; s_ < cnt; ++s_) {
            const double2 cur = slot ? slog[s_][lane] : make_double2(1.0, 0.0);
            if (!((st0 + s_) & 1)) {  // pair (2k, 2k + 1): (p, q) = (a, b) -> q' moves down, p' moves up
                if (slot) {
                    const double na = cur.y * a + cur.x * b, nb2 = cur.x * a - cur.y * b;
                    a = na;
                    b = nb2;
                }
            } else {                  // pair (2k + 1, 2k + 2) = (b of lane k, a of lane k + 1), slots k < half - 1
                const double an = wave_take_next(a), bp = wave_take_prev(b);
                const double2 prv = (lane >= 1 && lane < half) ? slog[s_][lane - 1] : make_double2(1.0, 0.0);
                const double nb2 = cur.y * b + cur.x * an;       // position 2k + 1 <- q' = s p + c q
                const double na = prv.x * bp - prv.y * a;        // position 2k     <- p' = c p - s q of slot k - 1
                if (lane < half - 1) b = nb2;
                if (lane >= 1 && lane < half) a = na;
            }
        }
    }
    if (i < D && slot) {
        const int k0 = 2 * lane;
        w.V[(size_t)i * n + k0] = a;
        w.V[(size_t)i * n + k0 + 1] = b;
        w.W[(size_t)i * n + k0] = a * w.G[(size_t)k0 * D + i];  // summed over i by hj_finish_kernel: lambda = v . g
        w.W[(size_t)i * n + k0 + 1] = b * w.G[(size_t)(k0 + 1) * D + i];
    }
}

// eigenvalues; fn 3: out[k][:] = v_k; fn 1 / 2: T[k][:] = f(lambda_k) v_k for the product out = V T.  Eigenpair k sits at position
// k, or k + 1 when D is odd and the zero dummy column has ended at position 0 (every sweep reverses the order of the positions)
__global__ __launch_bounds__(256) void hj_finish_kernel(int D, int fn, void* __restrict__ ws, double* __restrict__ eigvals,
                                                        double* __restrict__ out, int pass) {
    __shared__ double lam[128], vscale[128];
    __shared__ double s_max[4];
    __shared__ int s_bad;
    const int n = (D + 1) & ~1;
    const HjWs w = hj_ws(ws, blockIdx.x, D);
    if (pass == 1 && w.ctl->redo == 0) return;
    if (threadIdx.x == 0) s_bad = 0;
    const int off = ((D & 1) && (w.ctl->sweeps & 1)) ? 1 : 0;
    // lambda_k = +-|g_k|: at convergence the column g_k = lambda_k v_k, and its norm carries the eigenvalue with RELATIVE
    // accuracy, which v_k . g_k (absolute accuracy eps |A|) does not -- for a covariance with condition 1e17 the dot product of
    // the smallest pair comes out as -3e-12 and its square root as NaN.  The sign comes from the dot product; a dot product
    // within rounding noise of zero (64 D eps of the largest) counts as non-negative.
    double mx = 0.0;
    for (int k = threadIdx.x; k < D; k += 256) {
        double dot = 0.0, nrm = 0.0;
        for (int i = 0; i < D; ++i) {
            dot += w.W[(size_t)i * n + k + off];
            const double gki = w.G[(size_t)(k + off) * D + i];
            nrm = fma(gki, gki, nrm);
        }
        lam[k] = dot;                 // signed, absolute accuracy
        w.T[k] = sqrt(nrm);           // scratch: |lambda_k|, relative accuracy
        mx = fmax(mx, fabs(dot));
    }
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = mx;
    __syncthreads();
    const double lmax = fmax(fmax(s_max[0], s_max[1]), fmax(s_max[2], s_max[3]));
    const double noise = 64.0 * D * 2.220446049250313e-16 * lmax;
    const double shift = w.ctl->shift;
    const bool unconverged = w.ctl->unconverged != 0, chol = w.ctl->chol != 0;
    for (int k = threadIdx.x; k < D; k += 256) {
        // an eigenpair satisfies |v . g| = |g|; a mixture inside a +-lambda pair (equal columns norms in G, i.e. a double
        // eigenvalue of A^2) does not.  Testable where the dot product's absolute noise is below 1e-6 of the norm; only unshifted
        // runs can hold such pairs (the shifted spectrum is non-negative)
        if (!chol && pass == 0 && shift == 0.0 && w.T[k] > 1e6 * noise && fabs(lam[k]) < (1.0 - 1e-6) * w.T[k]) s_bad = 1;
        const double v = chol ? w.T[k] * w.T[k] : (lam[k] < -noise ? -w.T[k] : w.T[k]) - shift;
        lam[k] = v;
        vscale[k] = chol ? (w.T[k] > 0.0 ? 1.0 / w.T[k] : 0.0) : 1.0;
        eigvals[(size_t)blockIdx.x * D + k] = unconverged ? __longlong_as_double(0x7ff8000000000000LL) : v;
    }
    __syncthreads();
    if (pass == 0 && threadIdx.x == 0) w.ctl->redo = s_bad;
    if (fn == 0) return;
    double* dst = fn == 3 ? out + (size_t)blockIdx.x * D * D : w.T;
    for (int e = threadIdx.x; e < D * D; e += 256) {
        const int k = e / D, i = e - k * D;
        const double v = w.V[(size_t)i * n + k + off] * vscale[k];
        // (hj_product_kernel multiplies by the raw V: in chol mode the second 1 / |g_k| rides on T)
        dst[e] = fn == 3 ? v : (fn == 1 ? sqrt(lam[k]) : 1.0 / sqrt(lam[k])) * v * vscale[k];
    }
}

// out[i][j] = sum_k V[i][k] T[k][j]
__global__ __launch_bounds__(256) void hj_product_kernel(int D, void* __restrict__ ws, double* __restrict__ out) {
    __shared__ double as[16][17], bs[16][17];
    const HjWs w = hj_ws(ws, blockIdx.z, D);
    const int n = (D + 1) & ~1, off = ((D & 1) && (w.ctl->sweeps & 1)) ? 1 : 0;
    double* ob = out + (size_t)blockIdx.z * D * D;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int i = blockIdx.y * 16 + ty, j = blockIdx.x * 16 + tx;
    double acc = 0.0;
    for (int k0 = 0; k0 < D; k0 += 16) {
        as[ty][tx] = (i < D && k0 + tx < D) ? w.V[(size_t)i * n + k0 + tx + off] : 0.0;
        bs[ty][tx] = (k0 + ty < D && j < D) ? w.T[(size_t)(k0 + ty) * D + j] : 0.0;
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) acc = fma(as[ty][kk], bs[kk][tx], acc);
        __syncthreads();
    }
    if (i < D && j < D) ob[(size_t)i * D + j] = acc;
}

static bool g_hj_lds_set = false;

int eigh_onesided(const double* A, int nb, int D, int fn, double* out, double* eigvals, void* ws, hipStream_t st, const double* Vinit,
                  double* g0, const int* warm) {
    const size_t lds = (size_t)((D + 1) & ~1) * (D + 1) * sizeof(double);  // n positions x (D + 1) rows
    if (lds > 65536 && !g_hj_lds_set) {
        // the kernel also holds 4 bytes of static LDS: ask for what D = 128 needs, not for the whole 160 KiB
        if (hipFuncSetAttribute((const void*)hj_sweep_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 129 * 8) != hipSuccess) {
            otvae_set_error("otvae_eigh_fn: cannot raise dynamic LDS limit");
            return OTVAE_ELAUNCH;
        }
        g_hj_lds_set = true;
    }
    const double* G0 = nullptr;
    if (Vinit) {  // G0[b][j][:] = A_b v_j = row j of Vinit_b A_b (A symmetric)
        gemm_f64_launch(0, 0, nb, D, D, D, 1.0, Vinit, (size_t)D * D, A, (size_t)D * D, 0.0, g0, st);
        G0 = g0;
    }
    static const int use_chol = getenv("OTVAE_EIGH_NO_CHOL") ? 0 : 1;
    for (int pass = 0; pass < 2; ++pass) {  // pass 1: no-ops unless pass 0 met a +-lambda pair (see the header)
        if (D > 64)
            hj_sweep_kernel<8><<<nb, 512, lds, st>>>(A, D, ws, G0, warm, pass, use_chol);
        else
            hj_sweep_kernel<16><<<nb, 512, lds, st>>>(A, D, ws, G0, warm, pass, use_chol);
        OTVAE_CHECK_LAUNCH("otvae_eigh_fn(sweeps)");
        hj_vectors_kernel<<<dim3(cdiv(D, 4), nb), 256, 0, st>>>(D, ws, Vinit, warm, pass);
        OTVAE_CHECK_LAUNCH("otvae_eigh_fn(vectors)");
        hj_finish_kernel<<<nb, 256, 0, st>>>(D, fn, ws, eigvals, out, pass);
        OTVAE_CHECK_LAUNCH("otvae_eigh_fn(finish)");
    }
    if (fn == 1 || fn == 2) {
        hj_product_kernel<<<dim3(cdiv(D, 16), cdiv(D, 16), nb), 256, 0, st>>>(D, ws, out);
        OTVAE_CHECK_LAUNCH("otvae_eigh_fn(product)");
    }
    return OTVAE_OK;
}

// ================================================================================================ 128 < D <= 1024
// The same iteration across workgroups (the reference's tests/test_latent_transport.py transports 64x4x4 = 1024-dimensional
// latents).  G and V live in global memory (8 MiB each at D = 1024: L2 / Infinity-Cache resident), column by column.  The
// columns are cut into blocks of 8; a ROUND pairs the blocks round-robin and one workgroup per block pair
//   * loads its 16 columns of G into LDS (131 KB at D = 1024), runs the odd-even sweep of hj_sweep_kernel over the 16
//     positions (a wave per pair slot, the kept column in registers, 16 steps), logging the rotations in LDS;
//   * writes the columns back to their HOME slots (16 steps reverse the order of the positions, so position p goes home to
//     15 - p): a column of G never changes its slot, the round-robin over blocks therefore meets every pair of columns;
//   * loads the same 16 columns of V and replays the logged rotations on them.
// Positive definite input (the covariances this path exists for) is first factored, A = L L^T (otvae_cholesky), and the
// iteration runs on the columns of X = L: X V = Q S gives A = X X^T = Q S^2 Q^T -- eigenvalues |x_k|^2, eigenvectors the
// NORMALISED FINAL COLUMNS themselves, so V is neither kept nor updated (ctl->qmode; the update role of a launch returns at
// once).  The iteration sees cond(A)^1/2 instead of the cond(A)^2 that working on the columns of A itself implies (the Gram
// matrix of A's columns is A^2), which is what its convergence speed depends on: an ill-conditioned 1024 x 1024 latent
// covariance needed > 16 sweeps on A and needs ~10 on its factor.  Round 2 iterated on the columns of L^T (the rows of L:
// no transposed copy, but V had to be carried and the sweep count is ~20 % higher: 11 against 9 at D = 128 in a numpy
// restatement of the ordering); OTVAE_EIGH_BLOCK_LT=1 brings that variant back for an A/B.  When the factorisation
// meets a non-positive pivot (device flag) the columns of A are iterated as in the small solver.
// One launch per round (nblk - 1 rounds per sweep); the stop test is a device flag: a round that rotated above the stop level
// marks the sweep, a tiny kernel after each sweep turns the remaining launches into no-ops once a sweep stayed quiet.
#define HJB_B 8
#define HJB_MAX_SWEEPS 24
#define HJB_MAX_D 1024

struct HjbCtl {
    int done, rotated, sweeps;
    int chol_info;  // 0: the iteration runs on the Cholesky factor; else on A itself (written by otvae_cholesky before the init kernel)
    int log_stamp[2];  // id of the launch whose column rotations fill rotation log 0 / 1 (-1: none)
    int qmode;         // the iteration runs on the columns of L (not of L^T): the eigenvectors are the final columns, V is not kept
};
#define HJB_VSLABS 2  // row slabs the eigenvector update of a block pair is split over (when the launch still fits the chip)

static __host__ __device__ inline int hjb_dp(int D) { return (D + 2 * HJB_B - 1) / (2 * HJB_B) * (2 * HJB_B); }

extern "C" int64_t otvae_eigh_block_onesided_ws(int nb, int D) {
    if (nb <= 0 || D <= 0) return -1;
    const int64_t Dp = hjb_dp(D);
    // G[Dp][D], V[Dp][D] (column-major: a column is contiguous), lam[Dp], sign dots [Dp], T[D][D] for f(A), two rotation logs
    // (one (c, s) pair per column pair and step of a round: Dp * HJB_B of them), ctl
    return (int64_t)nb * ((2 * Dp * (int64_t)D + 2 * Dp + (int64_t)D * D) * 8 + 2 * Dp * HJB_B * 16 + 256);
}

struct HjbWs {
    double *G, *V, *nrm, *dot, *T;
    double2* log;  // [2][pairs = Dp / (2 HJB_B)][2 HJB_B steps][HJB_B pair slots]
    HjbCtl* ctl;
};
__host__ __device__ static inline HjbWs hjb_ws(void* ws, int b, int D) {
    const size_t Dp = (size_t)hjb_dp(D);
    const size_t per = (2 * Dp * D + 2 * Dp + (size_t)D * D) * 8 + 2 * Dp * HJB_B * 16 + 256;
    char* w = (char*)ws + (size_t)b * per;
    HjbWs r;
    r.G = (double*)w;
    r.V = r.G + Dp * D;
    r.nrm = r.V + Dp * D;
    r.dot = r.nrm + Dp;
    r.T = r.dot + Dp;
    r.log = (double2*)(r.T + (size_t)D * D);
    r.ctl = (HjbCtl*)(r.log + 2 * Dp * HJB_B);
    return r;
}

__global__ __launch_bounds__(256) void hjb_init_kernel(const double* __restrict__ Ain, int D, void* __restrict__ ws, int lcols) {
    const HjbWs w = hjb_ws(ws, blockIdx.y, D);
    const double* Ab = Ain + (size_t)blockIdx.y * D * D;
    const size_t total = (size_t)hjb_dp(D) * D;
    for (size_t e = blockIdx.x * (size_t)256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int c = (int)(e / D), i = (int)(e - (size_t)c * D);
        double v = 0.0;  // zero dummy columns
        if (c < D) v = w.ctl->chol_info == 0 ? (lcols ? w.T[(size_t)i * D + c]   // column c of X = L (zero above the diagonal)
                                                      : w.T[e])                   // column c of X = L^T is row c of L
                                             : ((i >= c) ? Ab[(size_t)i * D + c] : Ab[(size_t)c * D + i]);  // lower triangle of A
        w.G[e] = v;
        w.V[e] = (c == i) ? 1.0 : 0.0;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        w.ctl->done = 0;
        w.ctl->rotated = 0;
        w.ctl->sweeps = 0;
        w.ctl->log_stamp[0] = w.ctl->log_stamp[1] = -1;
        w.ctl->qmode = (lcols && w.ctl->chol_info == 0) ? 1 : 0;
    }
}

// round-robin pairing of nblk (even) blocks: pair k of `round`
__device__ __forceinline__ void hjb_pair(int round, int k, int nblk, int& I, int& J) {
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

// One launch = the column rotations of round `launch % nrounds` (blocks 0 .. pairs-1, one block pair each) AND the eigenvector update
// of the PREVIOUS launch's round (blocks pairs .., HJB_VSLABS row slabs per block pair): the update only replays logged rotations
// on the rows of V, so it neither feeds the next round's rotations nor needs whole columns -- it runs beside them on otherwise idle
// CUs instead of behind them in the same workgroup (70 -> 45 us per round at D = 1024).  The log is double-buffered in the
// workspace and stamped with the launch that filled it: a rotation role that found the solver converged leaves no stamp, and the
// update role of the next launch has nothing to replay.
template <int R>  // rows per lane: D <= 64 R
__global__ __launch_bounds__(512) void hjb_round_kernel(int D, int launch, int nrounds, int rotate, int vslabs, void* __restrict__ ws) {
    extern __shared__ __align__(16) double hj_lds[];
    __shared__ double2 s_log[2 * HJB_B][HJB_B];
    __shared__ double s_nrm[2 * HJB_B];  // squared norms of the columns in the even positions (the kept columns' live in registers)
    __shared__ int s_rot;
    const HjbWs w = hjb_ws(ws, blockIdx.y, D);
    constexpr int NP = 2 * HJB_B;  // positions
    const int nblk = hjb_dp(D) / HJB_B, npairs = nblk / 2;
    const bool vrole = (int)blockIdx.x >= npairs;
    int pair_idx, row0, nrows, my_launch;
    if (!vrole) {
        if (!rotate || w.ctl->done) return;
        pair_idx = blockIdx.x;
        row0 = 0;
        nrows = D;
        my_launch = launch;
    } else {
        my_launch = launch - 1;
        if (my_launch < 0 || w.ctl->qmode || w.ctl->log_stamp[my_launch & 1] != my_launch) return;
        const int v = blockIdx.x - npairs;
        pair_idx = v / vslabs;
        const int per = ((D + vslabs - 1) / vslabs + 63) / 64 * 64;  // rows per slab, whole lanes
        row0 = (v % vslabs) * per;
        nrows = min(per, D - row0);
        if (nrows <= 0) return;
    }
    const int LD = nrows + 1;
    int I, J;
    hjb_pair(my_launch % nrounds, pair_idx, nblk, I, J);
    const int tid = threadIdx.x, lane = tid & 63, g = tid >> 6;  // wave g = pair slot g
    const int kpos = 2 * g + 1;
    auto home = [&](int pos) { return pos < HJB_B ? I * HJB_B + pos : J * HJB_B + (pos - HJB_B); };
    double* P = hj_lds;  // P[pos * LD + row - row0]
    double2* glog = w.log + ((size_t)(my_launch & 1) * npairs + pair_idx) * NP * HJB_B;
    if (tid == 0) s_rot = 0;
    if (vrole)
        for (int e = tid; e < NP * HJB_B; e += 512) s_log[e / HJB_B][e % HJB_B] = glog[e];

    auto sweep16 = [&](double* __restrict__ M, bool replay) {
        // load the 16 columns (a wave per column, two columns each)
        for (int pos = g; pos < NP; pos += 8) {
            const double* src = M + (size_t)home(pos) * D + row0;
            for (int i = lane; i < nrows; i += 64) P[pos * LD + i] = src[i];
        }
        __syncthreads();
        double keep[R];
        double nk = 0.0;  // |kept column|^2, tracked through the round: a step then needs ONE dot product (the pair's) instead of three
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int i = lane + 64 * u;
            keep[u] = i < nrows ? P[kpos * LD + i] : 0.0;
        }
        if (!replay) {
            double ne = 0.0;
#pragma unroll
            for (int u = 0; u < R; ++u) {
                const int i = lane + 64 * u;
                const double e = i < nrows ? P[(kpos - 1) * LD + i] : 0.0;
                nk = fma(keep[u], keep[u], nk);
                ne = fma(e, e, ne);
            }
            nk = wave_allsum_dpp(nk);
            ne = wave_allsum_dpp(ne);
            if (lane == 0) s_nrm[kpos - 1] = ne;
            __syncthreads();
        }
        auto step = [&](auto odd_tag, int t) {
            constexpr bool ODD = decltype(odd_tag)::value;
            const int opos = ODD ? kpos + 1 : kpos - 1;
            if (opos < NP) {
                double other[R];
                double* go = P + opos * LD;
#pragma unroll
                for (int u = 0; u < R; ++u) {
                    const int i = lane + 64 * u;
                    other[u] = i < nrows ? go[i] : 0.0;
                }
                double c = 1.0, s = 0.0;
                if (replay) {
                    const double2 cs = s_log[t][g];
                    c = cs.x;
                    s = cs.y;
#pragma unroll
                    for (int u = 0; u < R; ++u) {
                        const int i = lane + 64 * u;
                        double pub;
                        if (ODD) {
                            pub = c * keep[u] - s * other[u];
                            keep[u] = s * keep[u] + c * other[u];
                        } else {
                            pub = s * other[u] + c * keep[u];
                            keep[u] = c * other[u] - s * keep[u];
                        }
                        if (i < nrows) go[i] = pub;
                    }
                } else {
                    double gamma = 0.0;
#pragma unroll
                    for (int u = 0; u < R; ++u) gamma = fma(keep[u], other[u], gamma);
                    gamma = wave_allsum_dpp(gamma);
                    const double no = s_nrm[opos];
                    // the pair is (a, b) = (keep, other) in an odd step, (other, keep) in an even one; it becomes
                    // (c a - s b, s a + c b) with squared norms alpha - t gamma and beta + t gamma
                    const double alpha = ODD ? nk : no, beta = ODD ? no : nk;
                    const double ab = alpha * beta;
                    double tg = 0.0;
                    if (gamma * gamma > HJ_TOL2 * ab && ab > 1e-280) {
                        const double zeta = (beta - alpha) * 0.5 * nr_rcp(gamma);
                        const double az = fabs(zeta);
                        double tt;
                        if (az > 1e8) {
                            tt = 0.5 * nr_rcp(zeta);
                        } else {
                            const double q2 = fma(zeta, zeta, 1.0);
                            tt = nr_rcp(az + q2 * nr_rsq(q2));
                            tt = zeta >= 0.0 ? tt : -tt;
                        }
                        c = nr_rsq(fma(tt, tt, 1.0));
                        s = c * tt;
                        tg = tt * gamma;
                        if (lane == 0 && gamma * gamma > HJ_STOP2 * ab) s_rot = 1;
                    }
                    if (lane == 0) s_log[t][g] = make_double2(c, s);
                    // tracked norms of the two new columns; the shrinking one (alpha - t gamma) is recomputed below when the
                    // subtraction cancelled more than two digits (rank-deficient input: a column on its way to zero)
                    const double n_x = alpha - tg, n_y = beta + tg;
                    const bool recount = !(n_x > 0.01 * alpha) && tg != 0.0;
                    double n_pub = ODD ? n_x : n_y;
                    nk = ODD ? n_y : n_x;
                    double acc2 = 0.0;
#pragma unroll
                    for (int u = 0; u < R; ++u) {
                        const int i = lane + 64 * u;
                        double pub;
                        if (ODD) {
                            pub = c * keep[u] - s * other[u];
                            keep[u] = s * keep[u] + c * other[u];
                        } else {
                            pub = s * other[u] + c * keep[u];
                            keep[u] = c * other[u] - s * keep[u];
                        }
                        if (i < nrows) go[i] = pub;
                        if (recount) acc2 = ODD ? fma(pub, pub, acc2) : fma(keep[u], keep[u], acc2);
                    }
                    if (recount) {  // wave-uniform
                        acc2 = wave_allsum_dpp(acc2);
                        if (ODD) n_pub = acc2;
                        else nk = acc2;
                    }
                    if (lane == 0) s_nrm[opos] = n_pub;
                }
            } else if (!replay && lane == 0) {
                s_log[t][g] = make_double2(1.0, 0.0);  // the idle slot of an odd step: its log entry is never read, but it is copied
            }
            __syncthreads();
        };
        for (int t = 0; t < NP; t += 2) {
            step(std::false_type{}, t);
            step(std::true_type{}, t + 1);
        }
        // 16 steps reversed the order: position p holds what started at 15 - p and goes back there
        {
            double* dst = M + (size_t)home(NP - 1 - kpos) * D + row0;
#pragma unroll
            for (int u = 0; u < R; ++u) {
                const int i = lane + 64 * u;
                if (i < nrows) dst[i] = keep[u];
            }
        }
        for (int pos = 2 * g; pos < NP; pos += 16) {  // the even positions: wave g writes position 2g
            double* dst = M + (size_t)home(NP - 1 - pos) * D + row0;
            for (int i = lane; i < nrows; i += 64) dst[i] = P[pos * LD + i];
        }
        __syncthreads();
    };
    if (vrole) {
        sweep16(w.V, true);
        return;
    }
    sweep16(w.G, false);
    for (int e = tid; e < NP * HJB_B; e += 512) glog[e] = s_log[e / HJB_B][e % HJB_B];
    if (tid == 0) {
        if (s_rot) w.ctl->rotated = 1;  // benign race: every writer stores 1
        if (blockIdx.x == 0) w.ctl->log_stamp[launch & 1] = launch;
    }
}

// ---- the same round from the block pair's GRAM matrix ----------------------------------------------------------------------
// The 16 sequential column steps above each reduce over all D rows; here the rotation workgroup forms W = B^T B (16 x 16, the
// 16 columns B of the block pair; one v_mfma_f64_16x16x4_f64 per 4 rows: a lane's element of B is both operands), runs ONE
// cyclic sweep of two-sided Jacobi on W in LDS (15 steps of 8 disjoint rotations on 16-vectors, accumulated in Q) and applies
// B <- B Q as a product; the update role applies the same Q to the rows of V.  A numpy restatement (block pairs of 16 columns,
// one inner sweep, the Cholesky factor of covariance-like and graded matrices) needs the same number of outer sweeps as the
// column steps and ends at the same cosines (1e-15).  Measured at D = 1024 (phases switched off one at a time): a launch is
// ~31 us instead of ~40 -- ~15 us of it the launch itself plus streaming the 16 columns in, 2.6 us the Gram matrix, ~6 us the 8
// rotation steps, ~7 us writing the rotated columns back: every round moves the whole of X and V (33 MB) through HBM.
// Columns keep their slots (no exchange of places): the log is the 16 x 16 matrix Q of the block pair.
typedef double double4v_hj __attribute__((ext_vector_type(4)));

// B <- B Q for the 16 columns of a block pair, as out^T = Q^T B^T on the matrix cores: one 16-row tile per wave and MFMA group;
// lane (i = l % 16, k = l / 16) feeds A[i][k] = Q[4 ks + k][i] and B[k][j] = x[row 16 T + j][column 4 ks + k], and ends up with
// out[row 16 T + l % 16][column 4 r + l / 16] in accumulator register r (layout probed on gfx950, tools/probe/mfma_f64.hip): its four
// stores run along the rows.  From the LDS copy of the columns (rotation role), or in place on M (update role: a wave reads a
// tile's 16 x 16 entries before the MFMAs and writes them after, tiles of different waves share no row).
__device__ __forceinline__ void hjg_apply_lds(const double* __restrict__ P, int LD, double* __restrict__ M, int D, int I, int J,
                                              const double (*Qs)[17]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l16 = lane & 15, lk = lane >> 4;
    auto home = [&](int pos) { return pos < HJB_B ? I * HJB_B + pos : J * HJB_B + (pos - HJB_B); };
    double qa[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qa[ks] = Qs[4 * ks + lk][l16];
    const int ntiles = (D + 15) / 16;
    for (int T = wave; T < ntiles; T += 8) {
        double4v_hj acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[ks], P[(4 * ks + lk) * LD + 16 * T + l16], acc, 0, 0, 0);
        const int row = 16 * T + l16;
        if (row < D) {
#pragma unroll
            for (int r = 0; r < 4; ++r) M[(size_t)home(4 * r + lk) * D + row] = acc[r];
        }
    }
}

__device__ __forceinline__ void hjg_apply_inplace(double* __restrict__ M, int D, int row0, int nrows, int I, int J, const double (*Qs)[17]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l16 = lane & 15, lk = lane >> 4;
    auto home = [&](int pos) { return pos < HJB_B ? I * HJB_B + pos : J * HJB_B + (pos - HJB_B); };
    double qa[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qa[ks] = Qs[4 * ks + lk][l16];
    const int ntiles = (nrows + 15) / 16;
    for (int T = wave; T < ntiles; T += 8) {
        const int row = 16 * T + l16;
        double b[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) b[ks] = row < nrows ? M[(size_t)home(4 * ks + lk) * D + row0 + row] : 0.0;
        double4v_hj acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[ks], b[ks], acc, 0, 0, 0);
        if (row < nrows) {
#pragma unroll
            for (int r = 0; r < 4; ++r) M[(size_t)home(4 * r + lk) * D + row0 + row] = acc[r];
        }
    }
}

__global__ __launch_bounds__(512) void hjg_round_kernel(int D, int launch, int nrounds, int rotate, int vslabs, void* __restrict__ ws) {
    extern __shared__ __align__(16) double hj_lds[];
    __shared__ double s_W[16][17], s_Q[16][17], s_part[8][16][16];
    __shared__ double2 s_cs[8];
    __shared__ int s_pq[8][2];
    __shared__ int s_rot;
    const HjbWs w = hjb_ws(ws, blockIdx.y, D);
    constexpr int NP = 2 * HJB_B;
    const int nblk = hjb_dp(D) / HJB_B, npairs = nblk / 2;
    const bool vrole = (int)blockIdx.x >= npairs;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int pair_idx, my_launch;
    if (!vrole) {
        if (!rotate || w.ctl->done) return;
        pair_idx = blockIdx.x;
        my_launch = launch;
    } else {
        my_launch = launch - 1;
        if (my_launch < 0 || w.ctl->qmode || w.ctl->log_stamp[my_launch & 1] != my_launch) return;
        pair_idx = (blockIdx.x - npairs) / vslabs;
    }
    int I, J;
    hjb_pair(my_launch % nrounds, pair_idx, nblk, I, J);
    auto home = [&](int pos) { return pos < HJB_B ? I * HJB_B + pos : J * HJB_B + (pos - HJB_B); };
    double* glog = (double*)(w.log + ((size_t)(my_launch & 1) * npairs + pair_idx) * NP * HJB_B);  // 256 doubles: Q row-major
    if (vrole) {
        const int v = blockIdx.x - npairs;
        const int per = ((D + vslabs - 1) / vslabs + 63) / 64 * 64;
        const int row0 = (v % vslabs) * per, nrows = min(per, D - row0);
        if (nrows <= 0) return;
        if (tid < 256) s_Q[tid >> 4][tid & 15] = glog[tid];
        __syncthreads();
        hjg_apply_inplace(w.V, D, row0, nrows, I, J, s_Q);
        return;
    }
    // ---- rotation role
    const int Dr = (D + 15) & ~15, LD = Dr + 1;
    double* P = hj_lds;  // P[pos * LD + row], rows D .. Dr zero (whole 16-row tiles for the products)
    {   // a wave streams in its two columns with every load of a batch in flight before the first LDS store: 8 x 16 bytes per lane
        // and column cover 1024 rows, so that D = 1024 costs one memory round trip instead of one per 64 rows (D even: the rows of a
        // column start 16-byte aligned; odd D takes the 8-byte path)
        const double* src0 = w.G + (size_t)home(wave) * D;
        const double* src1 = w.G + (size_t)home(wave + 8) * D;
        double* d0 = P + wave * LD;
        double* d1 = P + (wave + 8) * LD;
        if ((D & 1) == 0) {
            for (int base = 0; base < Dr; base += 1024) {
                double2 v0[8], v1[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = base + 2 * lane + 128 * u;  // rows i, i + 1 (D even: both inside or both outside)
                    v0[u] = i < D ? *(const double2*)(src0 + i) : make_double2(0.0, 0.0);
                    v1[u] = i < D ? *(const double2*)(src1 + i) : make_double2(0.0, 0.0);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = base + 2 * lane + 128 * u;
                    if (i < Dr) {
                        d0[i] = v0[u].x;
                        d0[i + 1] = v0[u].y;
                        d1[i] = v1[u].x;
                        d1[i + 1] = v1[u].y;
                    }
                }
            }
        } else {
            for (int i = lane; i < Dr; i += 64) {
                d0[i] = i < D ? src0[i] : 0.0;
                d1[i] = i < D ? src1[i] : 0.0;
            }
        }
    }
    if (tid == 0) s_rot = 0;
    __syncthreads();
    {   // W = B^T B on the matrix cores: row group rg covers rows 4 rg .. 4 rg + 3; lane (c = l % 16, k = l / 16) holds B[4 rg + k][c]
        double4v_hj acc = {0.0, 0.0, 0.0, 0.0};
        const int c = lane & 15, k = lane >> 4;
        for (int rg = wave; rg < Dr / 4; rg += 8) {
            const double a = P[c * LD + 4 * rg + k];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) s_part[wave][4 * r + k][c] = acc[r];  // D[4 r + l / 16][l % 16]
    }
    __syncthreads();
    if (tid < 256) {
        const int i = tid >> 4, j = tid & 15;
        double t = 0.0;
#pragma unroll
        for (int g8 = 0; g8 < 8; ++g8) t += s_part[g8][i][j];
        s_W[i][j] = t;
        s_Q[i][j] = i == j ? 1.0 : 0.0;
    }
    __syncthreads();
    if (tid < 256) {  // does this pair still hold a cosine above the stop level?
        const int i = tid >> 4, j = tid & 15;
        const double gij = s_W[i][j], ab = s_W[i][i] * s_W[j][j];
        if (i < j && gij * gij > HJ_STOP2 * ab && ab > 1e-280) s_rot = 1;  // benign race
    }
    __syncthreads();  // the sweep below rewrites W
    if (wave == 0) {  // two-sided Jacobi on W in LDS, 8 disjoint rotations per step, accumulated in Q
        volatile double(*W)[17] = s_W;
        volatile double(*Q)[17] = s_Q;
        // Every round rotates the 64 pairs ACROSS the two blocks (8 steps: column i against column 8 + (i + t) % 8); the pairs inside
        // a block, which every round of a sweep would meet again, are only rotated by the first round of a sweep (a full round-robin
        // of 15 steps).  Same final cosines, at most one outer sweep more (numpy restatement), half the steps per round.
        const bool full = (my_launch % nrounds) == 0;
        const int nsteps = full ? NP - 1 : HJB_B;
        for (int t = 0; t < nsteps; ++t) {
            if (lane < 8) {
                int p, q;
                if (full) {
                    hjb_pair(t, lane, NP, p, q);
                } else {
                    p = lane;
                    q = HJB_B + ((lane + t) & (HJB_B - 1));
                }
                const double alpha = W[p][p], beta = W[q][q], gamma = W[p][q];
                double c = 1.0, s_ = 0.0;
                const double ab = alpha * beta;
                if (gamma * gamma > HJ_TOL2 * ab && ab > 1e-280) {
                    // the tangent need not be exact (an inexact angle only leaves a little more for the next visit): hardware
                    // seeds + one Newton step; c and s are made orthonormal to rounding below
                    double y = __builtin_amdgcn_rcp(gamma);
                    y = fma(fma(-gamma, y, 1.0), y, y);
                    const double zeta = (beta - alpha) * 0.5 * y;
                    const double az = fabs(zeta);
                    double tt;
                    if (az > 1e8) {
                        tt = 0.5 * __builtin_amdgcn_rcp(zeta);
                    } else {
                        const double q2 = fma(zeta, zeta, 1.0);
                        double rq = __builtin_amdgcn_rsq(q2);
                        rq = rq * fma(-0.5 * q2 * rq, rq, 1.5);
                        const double den = az + q2 * rq;
                        double rd = __builtin_amdgcn_rcp(den);
                        rd = fma(fma(-den, rd, 1.0), rd, rd);
                        tt = zeta >= 0.0 ? rd : -rd;
                    }
                    c = nr_rsq(fma(tt, tt, 1.0));
                    s_ = c * tt;
                }
                s_cs[lane] = make_double2(c, s_);
                s_pq[lane][0] = p;
                s_pq[lane][1] = q;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // The 8 pairs of a step cover all 16 indices, so W falls into 8 x 8 blocks of 2 x 2 entries, block (a, b) = rows of pair a x
            // columns of pair b, and J^T W J maps every block onto itself: W_ab <- J_a^T W_ab J_b.  One lane per block, one pass (the
            // column update followed by the row update of the first version was two LDS round trips and a barrier more per step).
            {
                const int a = lane >> 3, b = lane & 7;
                const int pa = s_pq[a][0], qa = s_pq[a][1], pb = s_pq[b][0], qb = s_pq[b][1];
                const double ca = s_cs[a].x, sa = s_cs[a].y, cb = s_cs[b].x, sb = s_cs[b].y;
                const double w00 = W[pa][pb], w01 = W[pa][qb], w10 = W[qa][pb], w11 = W[qa][qb];
                // columns: (p, q) <- (c p - s q, s p + c q) with pair b's rotation
                const double t00 = cb * w00 - sb * w01, t01 = sb * w00 + cb * w01;
                const double t10 = cb * w10 - sb * w11, t11 = sb * w10 + cb * w11;
                // rows, with pair a's rotation
                W[pa][pb] = ca * t00 - sa * t10;
                W[pa][qb] = ca * t01 - sa * t11;
                W[qa][pb] = sa * t00 + ca * t10;
                W[qa][qb] = sa * t01 + ca * t11;
                // Q <- Q J: rows 2a, 2a + 1 against pair b
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    const int r = 2 * a + rr;
                    const double qp = Q[r][pb], qq = Q[r][qb];
                    Q[r][pb] = cb * qp - sb * qq;
                    Q[r][qb] = sb * qp + cb * qq;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
    if (tid < 256) glog[tid] = s_Q[tid >> 4][tid & 15];
    hjg_apply_lds(P, LD, w.G, D, I, J, s_Q);
    if (tid == 0) {
        if (s_rot) w.ctl->rotated = 1;
        if (blockIdx.x == 0) w.ctl->log_stamp[launch & 1] = launch;
    }
}

// after every sweep: one thread closes the sweep of every matrix of the batch; `host_done` (mapped host memory, or NULL) learns
// whether the whole batch has converged -- the host stops enqueueing rounds two sweeps later (see eigh_block_onesided)
__global__ void hjb_check_kernel(int D, int nb, void* __restrict__ ws, volatile int* host_done) {
    if (threadIdx.x != 0) return;
    int all = 1;
    for (int b = 0; b < nb; ++b) {
        const HjbWs w = hjb_ws(ws, b, D);
        if (!w.ctl->done) {
            w.ctl->sweeps += 1;
            if (!w.ctl->rotated) w.ctl->done = 1;
            w.ctl->rotated = 0;
        }
        all &= w.ctl->done;
    }
    if (host_done) {
        *host_done = all;
        __threadfence_system();
    }
}

// |lambda_k| = |g_k| and the sign's dot product v_k . g_k: a wave per column
__global__ __launch_bounds__(256) void hjb_norms_kernel(int D, void* __restrict__ ws) {
    const HjbWs w = hjb_ws(ws, blockIdx.y, D);
    const int lane = threadIdx.x & 63, k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= D) return;
    const double* gk = w.G + (size_t)k * D;
    const double* vk = w.V + (size_t)k * D;
    double nn = 0.0, dd = 0.0;
    for (int i = lane; i < D; i += 64) {
        nn = fma(gk[i], gk[i], nn);
        dd = fma(gk[i], vk[i], dd);
    }
    nn = wave_sum(nn);
    dd = wave_sum(dd);
    if (lane == 0) {
        w.nrm[k] = sqrt(nn);
        w.dot[k] = dd;
    }
}

// eigenvalues; fn 3: out[k][:] = v_k (a copy: V is stored one eigenvector per row already); fn 1 / 2: T[k][:] = f(lambda_k) v_k
__global__ __launch_bounds__(256) void hjb_finish_kernel(int D, int fn, void* __restrict__ ws, double* __restrict__ eigvals,
                                                         double* __restrict__ out) {
    __shared__ double s_max[4];
    __shared__ double s_noise;
    const HjbWs w = hjb_ws(ws, blockIdx.y, D);
    double mx = 0.0;
    for (int k = threadIdx.x; k < D; k += 256) mx = fmax(mx, fabs(w.dot[k]));
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) s_noise = 64.0 * D * 2.220446049250313e-16 * fmax(fmax(s_max[0], s_max[1]), fmax(s_max[2], s_max[3]));
    __syncthreads();
    const double noise = s_noise;
    const bool chol = w.ctl->chol_info == 0, qmode = w.ctl->qmode != 0;
    // the columns of an indefinite A were iterated unshifted: a +-lambda pair of equal size is a double eigenvalue of A^2 and the
    // iteration may have stopped at a mixture of its two eigenvectors (|v . g| < |g|, see the header of this file).  The small
    // solver repeats such a matrix on A + |A|_inf I; here (a hundred launches per sweep) the matrix gets NaN eigenvalues instead
    // of a silently wrong answer.  Covariances -- what this path exists for -- are never indefinite beyond rounding.
    int bad = 0;
    if (!chol)
        for (int k = threadIdx.x; k < D; k += 256)
            if (w.nrm[k] > 1e6 * noise && fabs(w.dot[k]) < (1.0 - 1e-6) * w.nrm[k]) bad = 1;
    bad = __syncthreads_or(bad);
    const double qnan = __longlong_as_double(0x7ff8000000000000LL);
    auto lam = [&](int k) { return bad ? qnan : chol ? w.nrm[k] * w.nrm[k] : (w.dot[k] < -noise ? -w.nrm[k] : w.nrm[k]); };
    if (blockIdx.x == 0)
        for (int k = threadIdx.x; k < D; k += 256) eigvals[(size_t)blockIdx.y * D + k] = lam(k);
    if (fn == 0) return;
    double* dst = fn == 3 ? out + (size_t)blockIdx.y * D * D : w.T;
    const size_t total = (size_t)D * D;
    for (size_t e = blockIdx.x * (size_t)256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int k = (int)(e / D);
        double v;
        if (qmode) {  // the eigenvectors are the normalised final columns; V (one eigenvector per row) is filled for the product below
            v = w.nrm[k] > 0.0 ? w.G[e] / w.nrm[k] : 0.0;
            if (fn != 3) w.V[e] = v;
        } else {
            v = w.V[e];
        }
        if (fn == 3) {
            dst[e] = v;
        } else {
            const double l = lam(k);
            dst[e] = (fn == 1 ? sqrt(l) : 1.0 / sqrt(l)) * v;
        }
    }
}


static bool g_hjb_lds_set[2] = {false, false};
static bool g_hjg_lds_set = false;

int eigh_block_onesided(const double* A, int nb, int D, int fn, double* out, double* eigvals, void* ws, hipStream_t st) {
    const size_t lds = (size_t)2 * HJB_B * (D + 1) * sizeof(double);
    const bool big = D > 512;
    if (lds > 65536 && !g_hjb_lds_set[big]) {
        const hipError_t e = big ? hipFuncSetAttribute((const void*)hjb_round_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * HJB_B * (HJB_MAX_D + 1) * 8))
                                 : hipFuncSetAttribute((const void*)hjb_round_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * HJB_B * 513 * 8));
        if (e != hipSuccess) {
            otvae_set_error("otvae_eigh_fn: cannot raise dynamic LDS limit");
            return OTVAE_ELAUNCH;
        }
        g_hjb_lds_set[big] = true;
    }
    if (!g_hjg_lds_set) {
        if (hipFuncSetAttribute((const void*)hjg_round_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(2 * HJB_B * (HJB_MAX_D + 1) * 8)) != hipSuccess) {
            otvae_set_error("otvae_eigh_fn: cannot raise dynamic LDS limit");
            return OTVAE_ELAUNCH;
        }
        g_hjg_lds_set = true;
    }
    const int Dp = hjb_dp(D), nblk = Dp / HJB_B;
    {  // L into the T area (free until the end), the pivot flag into the control block: the whole batch in the same launches
        const HjbWs w0 = hjb_ws(ws, 0, D);
        const size_t per = (size_t)((char*)hjb_ws(ws, 1, D).G - (char*)w0.G);  // workspace pitch per matrix (a multiple of 8 bytes)
        const int rc = cholesky_blocked(A, nb, D, w0.T, per / sizeof(double), &w0.ctl->chol_info, per / sizeof(int), st);
        if (rc) return rc;
    }
    static const int lcols = getenv("OTVAE_EIGH_BLOCK_LT") ? 0 : 1;  // A/B switch: the round-2 iteration on the columns of L^T
    hjb_init_kernel<<<dim3(imin(cdiv((size_t)Dp * D, 256), 1024), nb), 256, 0, st>>>(A, D, ws, lcols);
    const int nrounds = nblk - 1;
    // one workgroup per CU at these LDS sizes: split the eigenvector update over row slabs only while every block of a launch is
    // still resident at once (two matrices of D = 1024 side by side are 128 rotation + 128 update blocks already)
    const int vslabs = (nblk / 2 * (1 + HJB_VSLABS) * nb <= 256) ? HJB_VSLABS : 1;
    const dim3 grid(nblk / 2 * (1 + vslabs), nb);  // rotation blocks, then the eigenvector-update blocks of the launch before
    static const bool column_steps = getenv("OTVAE_EIGH_COLUMN_STEPS") != nullptr;  // A/B switch: the first-generation rounds
    const size_t lds_g = (size_t)2 * HJB_B * (((D + 15) & ~15) + 1) * sizeof(double);
    auto round_launch = [&](int launch, int rotate) {
        if (!column_steps)
            hjg_round_kernel<<<grid, 512, lds_g, st>>>(D, launch, nrounds, rotate, vslabs, ws);
        else if (big)
            hjb_round_kernel<16><<<grid, 512, lds, st>>>(D, launch, nrounds, rotate, vslabs, ws);
        else
            hjb_round_kernel<8><<<grid, 512, lds, st>>>(D, launch, nrounds, rotate, vslabs, ws);
    };
    // The launches of a converged solver are no-ops, but a no-op still costs its ~2 us of queue time and a sweep is 31 ... 127 of
    // them: the 24-sweep budget, issued blindly, spent 1 ms (D = 256) ... 4 ms (D = 1024) behind the ~10 sweeps that did the
    // work.  Outside a stream capture the host therefore FOLLOWS the device two sweeps behind: the check kernel of sweep s writes
    // "all converged" to mapped host memory and an event marks it; before enqueueing sweep s + 2 the host waits for that event (the
    // device is busy with sweep s + 1 meanwhile: no bubble) and stops when the flag is up -- at most one sweep of no-ops is left.
    // Under capture (nothing may synchronise) the whole budget is recorded as before.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess) cap = hipStreamCaptureStatusNone;
    static const bool no_follow = getenv("OTVAE_EIGH_NO_FOLLOW") != nullptr;  // A/B switch
    // The pinned flags, the two events and the lock are PER DEVICE: events are bound to the device that was current when
    // they were made, so one process-wide pair failed hipEventRecord on a second GPU's stream, and one lock serialised the solves
    // of all devices.  Indexed by the calling thread's current device, which is the stream's device under torch.
    struct FollowState {
        std::mutex mu;
        int* flags = nullptr;  // [HJB_MAX_SWEEPS], mapped host memory
        hipEvent_t ev[2];
    };
    static FollowState follow_tab[16];
    int follow_dev = -1;
    bool follow = cap == hipStreamCaptureStatusNone && !no_follow;
    if (follow && (hipGetDevice(&follow_dev) != hipSuccess || follow_dev < 0 || follow_dev >= 16)) follow = false;
    FollowState& fs = follow_tab[follow ? follow_dev : 0];
    std::unique_lock<std::mutex> follow_lock(fs.mu, std::defer_lock);
    if (follow) {
        follow_lock.lock();
        if (!fs.flags) {
            int* f = nullptr;
            if (hipHostMalloc((void**)&f, HJB_MAX_SWEEPS * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
                hipEventCreateWithFlags(&fs.ev[0], hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&fs.ev[1], hipEventDisableTiming) == hipSuccess)
                fs.flags = f;
            else
                follow = false;  // (no pinned memory: the blind budget still gives the right answer)
        }
    }
    int* const follow_flags = fs.flags;
    hipEvent_t* const follow_ev = fs.ev;
    int issued = 0;
    for (int sweep = 0; sweep < HJB_MAX_SWEEPS; ++sweep) {
        if (follow && sweep >= 2) {
            if (hipEventSynchronize(follow_ev[sweep & 1]) != hipSuccess) {
                otvae_set_error("otvae_eigh_fn: the device failed while the host followed the sweeps");
                return OTVAE_ELAUNCH;
            }
            if (((volatile int*)follow_flags)[sweep - 2]) break;
        }
        for (int round = 0; round < nrounds; ++round) round_launch(sweep * nrounds + round, 1);
        if (follow) follow_flags[sweep] = 0;
        hjb_check_kernel<<<1, 64, 0, st>>>(D, nb, ws, follow ? follow_flags + sweep : nullptr);
        if (follow && hipEventRecord(follow_ev[sweep & 1], st) != hipSuccess) {
            otvae_set_error("otvae_eigh_fn: hipEventRecord failed");
            return OTVAE_ELAUNCH;
        }
        issued = sweep + 1;
    }
    round_launch(issued * nrounds, 0);  // the eigenvector update of the very last round, if the last sweep issued still rotated
    OTVAE_CHECK_LAUNCH("otvae_eigh_fn(block rounds)");
    hjb_norms_kernel<<<dim3(cdiv(D, 4), nb), 256, 0, st>>>(D, ws);
    hjb_finish_kernel<<<dim3(imin(cdiv((size_t)D * D, 2048), 256), nb), 256, 0, st>>>(D, fn, ws, eigvals, out);
    if (fn == 1 || fn == 2)  // out = V^T T (V holds one eigenvector per row): the fp64 matrix-core product of gaussian_ot.hip
        for (int b = 0; b < nb; ++b) {
            const HjbWs w = hjb_ws(ws, b, D);
            gemm_f64_launch(1, 0, 1, D, D, D, 1.0, w.V, 0, w.T, 0, 0.0, out + (size_t)b * D * D, st);
        }
    OTVAE_CHECK_LAUNCH("otvae_eigh_fn(block finish)");
    return OTVAE_OK;
}
