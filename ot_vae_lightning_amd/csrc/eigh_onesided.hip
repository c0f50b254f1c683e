// Symmetric eigensolver for D <= 128 (the latent sizes of the CNN configurations): one-sided (Hestenes) Jacobi.
// Replaces torch.linalg.eigh inside the reference's sqrtm / invsqrtm / min_eig (ot/matrix_utils.py:37-46,91-109).
//
// Why one-sided: the two-sided form (gaussian_ot.hip: eigh_kernel) rotates rows AND columns of A in LDS and the rows of V^T in
// global memory every step -- three barriers and a dependent global round trip per step, 8.6 us per step, 11 ms per
// 128 x 128 matrix.  Here G = A V is the only matrix the iteration touches: a step orthogonalises De/2 disjoint column
// pairs of G (round-robin tournament), one barrier per step, everything in LDS (G is 128 KB at D = 128).  The rotations are
// NOT applied to V inside the loop: they are logged (16 bytes per pair and step), and a second kernel replays the log on
// the rows of V = I, which are independent of each other -- D waves, each with its row in LDS, 64 disjoint rotations per
// step handled by the 64 lanes.  Eigenvalues are lambda_k = v_k . g_k (signed: indefinite and singular matrices are fine),
// f(A) = V f(Lambda) V^T is one product.  Arithmetic is fp64 throughout; the iteration ends after the first sweep without a
// rotation.  A matrix with a negative diagonal entry (certainly indefinite) is solved as A + |A|_inf I and the shift taken
// off the eigenvalues: one-sided Jacobi sees A^2, in which +lambda and -lambda of equal magnitude are a double eigenvalue.
#include "common.h"

#define HJ_MAX_SWEEPS 24
// a pair is rotated while |g_p . g_q| > tol |g_p| |g_q| with tol = 2e-14 (LAPACK's one-sided Jacobi, dgesvj, uses sqrt(M) eps =
// 2.5e-15 at M = 128; at 1e-15 rounding noise re-triggers rotations sweep after sweep and the solver never sees a quiet sweep)
#define HJ_TOL2 4e-28

struct HjCtl {
    int steps;   // steps whose rotations are in the log
    int sweeps;
    double shift;  // added to the diagonal before the iteration (0 unless some diagonal entry was negative)
};

// seats (unordered) of slot k at `step`: slot 0 keeps player De - 1 and meets step % m; slot k > 0 holds (step + k) % m and
// (step - k) % m.  From one step to the next every moving seat advances by one (mod m).
__device__ __forceinline__ void hj_seats(int step, int k, int De, int& a, int& b) {
    const int m = De - 1;
    if (k == 0) {
        a = m;
        b = step % m;
    } else {
        a = (step + k) % m;
        b = (step - k + m) % m;
    }
}

__device__ __forceinline__ void hj_pair(int step, int k, int De, int& p, int& q) {
    const int m = De - 1;
    if (k == 0) {
        p = m;
        q = step % m;
    } else {
        p = (step + k) % m;
        q = (step - k + m) % m;
    }
    if (p > q) {
        const int t = p;
        p = q;
        q = t;
    }
}

extern "C" int64_t otvae_eigh_onesided_ws(int nb, int D) {
    const int De = (D + 1) & ~1, half = De / 2;
    const int64_t per = 4 * (int64_t)D * D * 8 + (int64_t)HJ_MAX_SWEEPS * (De - 1) * half * 16 + 256;
    if (nb <= 0 || D <= 0) return -1;
    return (int64_t)nb * ((per + 255) & ~(int64_t)255);
}

struct HjWs {
    double *G, *V, *W, *T;
    double2* log;
    HjCtl* ctl;
};
__host__ __device__ static inline HjWs hj_ws(void* ws, int b, int D) {
    const int De = (D + 1) & ~1, half = De / 2;
    const size_t per = ((4 * (size_t)D * D * 8 + (size_t)HJ_MAX_SWEEPS * (De - 1) * half * 16 + 256) + 255) & ~(size_t)255;
    char* w = (char*)ws + (size_t)b * per;
    HjWs r;
    r.G = (double*)w;
    r.V = r.G + (size_t)D * D;
    r.W = r.V + (size_t)D * D;
    r.T = r.W + (size_t)D * D;
    r.log = (double2*)(r.T + (size_t)D * D);
    r.ctl = (HjCtl*)((char*)r.log + (size_t)HJ_MAX_SWEEPS * (De - 1) * half * 16);
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

// one workgroup of 512 threads per matrix: L lanes per column pair (8 for D > 64: 64 pairs; 16 below), rows strided by L
template <int L>
__global__ __launch_bounds__(512) void hj_sweep_kernel(const double* __restrict__ Ain, int D, void* __restrict__ ws) {
    extern __shared__ __align__(16) double hj_lds[];
    __shared__ int s_rot;
    constexpr int R = 128 / L;  // rows per lane at most (D <= 128 for L = 8, D <= 64 for L = 16 -> R = 16 / 4 used)
    const int De = (D + 1) & ~1, half = De / 2, LD = D + 1;
    const HjWs w = hj_ws(ws, blockIdx.x, D);
    const double* Ab = Ain + (size_t)blockIdx.x * D * D;
    double* G = hj_lds;  // column major: G[col * LD + row]
    const int tid = threadIdx.x, grp = tid / L, r = tid % L;
    for (int e = tid; e < D * D; e += 512) {
        const int i = e / D, j = e - i * D;  // reads the lower triangle, like torch.linalg.eigh(UPLO='L')
        G[j * LD + i] = (i >= j) ? Ab[(size_t)i * D + j] : Ab[(size_t)j * D + i];
    }
    if (tid == 0) s_rot = 0;
    __syncthreads();
    // certainly indefinite (a negative diagonal entry): shift by the infinity norm, so that the spectrum becomes non-negative
    __shared__ double s_red[8], s_shift;
    {
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
            s_shift = m8 < 0.0 ? r8 : 0.0;
        }
        __syncthreads();
        const double sh = s_shift;
        if (sh != 0.0)
            for (int i = tid; i < D; i += 512) G[i * LD + i] += sh;
        __syncthreads();
    }
    const int nrow = (D + L - 1) / L;
    int gstep = 0, sweep = 0;
    // round-robin seats of this slot without a modulo per step: both move one seat forward (mod De - 1) every step
    const int m = De - 1;
    int sp, sq;
    hj_seats(0, grp < half ? grp : 0, De, sp, sq);
    for (; sweep < HJ_MAX_SWEEPS; ++sweep) {
        for (int step = 0; step < De - 1; ++step, ++gstep) {
            const int p = sp < sq ? sp : sq, q = sp < sq ? sq : sp;
            if (grp != 0) sp = sp + 1 == m ? 0 : sp + 1;
            sq = sq + 1 == m ? 0 : sq + 1;
            if (grp < half) {
                double c = 1.0, s = 0.0;
                if (q < D) {
                    double a[R], b[R], alpha = 0.0, beta = 0.0, gamma = 0.0;
                    double* gp = G + p * LD;
                    double* gq = G + q * LD;
#pragma unroll
                    for (int u = 0; u < R; ++u) {
                        const int i = r + L * u;
                        const bool ok = u < nrow && i < D;
                        a[u] = ok ? gp[i] : 0.0;
                        b[u] = ok ? gq[i] : 0.0;
                        alpha = fma(a[u], a[u], alpha);
                        beta = fma(b[u], b[u], beta);
                        gamma = fma(a[u], b[u], gamma);
                    }
                    alpha = group_sum<L>(alpha);
                    beta = group_sum<L>(beta);
                    gamma = group_sum<L>(gamma);
                    const double ab = alpha * beta;
                    if (gamma * gamma > HJ_TOL2 * ab && ab > 1e-280) {
                        const double zeta = (beta - alpha) * 0.5 * nr_rcp(gamma);
                        const double az = fabs(zeta);
                        double t;
                        if (az > 1e8) {
                            t = 0.5 * nr_rcp(zeta);  // |zeta| + sqrt(1 + zeta^2) = 2 |zeta| to rounding
                        } else {
                            const double q2 = fma(zeta, zeta, 1.0);
                            t = nr_rcp(az + q2 * nr_rsq(q2));
                            t = zeta >= 0.0 ? t : -t;
                        }
                        c = nr_rsq(fma(t, t, 1.0));
                        s = c * t;
#pragma unroll
                        for (int u = 0; u < R; ++u) {
                            const int i = r + L * u;
                            if (u < nrow && i < D) {
                                gp[i] = c * a[u] - s * b[u];
                                gq[i] = s * a[u] + c * b[u];
                            }
                        }
                        if (r == 0) s_rot = 1;  // benign race: every writer stores 1
                    }
                }
                if (r == 0) w.log[(size_t)gstep * half + grp] = make_double2(c, s);
            }
            __syncthreads();
        }
        const int rotated = s_rot;  // read by every thread after the step's barrier
        __syncthreads();
        if (!rotated) {
            ++sweep;
            break;
        }
        if (tid == 0) s_rot = 0;
        __syncthreads();
    }
    for (int e = tid; e < D * D; e += 512) {
        const int k = e / D, i = e - k * D;
        w.G[e] = G[k * LD + i];  // column k contiguous
    }
    if (tid == 0) {
        w.ctl->steps = gstep;
        w.ctl->sweeps = sweep;
        w.ctl->shift = s_shift;
    }
}

// V = J_1 J_2 ... applied to the rows of the identity: a wave per row, its row in LDS, lane k replays pair slot k
__global__ __launch_bounds__(256) void hj_vectors_kernel(int D, void* __restrict__ ws) {
    __shared__ double rows[4][130];
    const int De = (D + 1) & ~1, half = De / 2;
    const HjWs w = hj_ws(ws, blockIdx.y, D);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wv;
    double* row = rows[wv];
    for (int k = lane; k < De; k += 64) row[k] = (k == i) ? 1.0 : 0.0;
    const int steps = w.ctl->steps;
    __syncthreads();
    double2 cs = (lane < half && steps > 0) ? w.log[lane] : make_double2(1.0, 0.0);
    const int m = De - 1;
    int sp, sq;
    hj_seats(0, lane < half ? lane : 0, De, sp, sq);
    volatile double* vrow = row;  // the wave's own row: LDS operations of one wave execute in order, no workgroup barrier
    for (int st = 0; st < steps; ++st) {
        const double2 cur = cs;
        if (lane < half && st + 1 < steps) cs = w.log[(size_t)(st + 1) * half + lane];  // next step's pair, in flight
        const int p = sp < sq ? sp : sq, q = sp < sq ? sq : sp;
        if (lane != 0) sp = sp + 1 == m ? 0 : sp + 1;
        sq = sq + 1 == m ? 0 : sq + 1;
        if (lane < half && cur.y != 0.0) {
            const double vp = vrow[p], vq = vrow[q];
            vrow[p] = cur.x * vp - cur.y * vq;
            vrow[q] = cur.y * vp + cur.x * vq;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    if (i < D)
        for (int k = lane; k < D; k += 64) {
            const double v = row[k];
            w.V[(size_t)i * D + k] = v;
            w.W[(size_t)i * D + k] = v * w.G[(size_t)k * D + i];  // summed over i by hj_finish_kernel: lambda_k = v_k . g_k
        }
}

// eigenvalues; fn 3: out[k][:] = v_k; fn 1 / 2: T[k][:] = f(lambda_k) v_k for the product out = V T
__global__ __launch_bounds__(256) void hj_finish_kernel(int D, int fn, void* __restrict__ ws, double* __restrict__ eigvals,
                                                        double* __restrict__ out) {
    __shared__ double lam[128];
    const HjWs w = hj_ws(ws, blockIdx.x, D);
    for (int k = threadIdx.x; k < D; k += 256) {
        double s = 0.0;
        for (int i = 0; i < D; ++i) s += w.W[(size_t)i * D + k];
        s -= w.ctl->shift;
        lam[k] = s;
        eigvals[(size_t)blockIdx.x * D + k] = s;
    }
    __syncthreads();
    if (fn == 0) return;
    double* dst = fn == 3 ? out + (size_t)blockIdx.x * D * D : w.T;
    for (int e = threadIdx.x; e < D * D; e += 256) {
        const int k = e / D, i = e - k * D;
        const double v = w.V[(size_t)i * D + k];
        dst[e] = fn == 3 ? v : (fn == 1 ? sqrt(lam[k]) : 1.0 / sqrt(lam[k])) * v;
    }
}

// out[i][j] = sum_k V[i][k] T[k][j]
__global__ __launch_bounds__(256) void hj_product_kernel(int D, void* __restrict__ ws, double* __restrict__ out) {
    __shared__ double as[16][17], bs[16][17];
    const HjWs w = hj_ws(ws, blockIdx.z, D);
    double* ob = out + (size_t)blockIdx.z * D * D;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int i = blockIdx.y * 16 + ty, j = blockIdx.x * 16 + tx;
    double acc = 0.0;
    for (int k0 = 0; k0 < D; k0 += 16) {
        as[ty][tx] = (i < D && k0 + tx < D) ? w.V[(size_t)i * D + k0 + tx] : 0.0;
        bs[ty][tx] = (k0 + ty < D && j < D) ? w.T[(size_t)(k0 + ty) * D + j] : 0.0;
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) acc = fma(as[ty][kk], bs[kk][tx], acc);
        __syncthreads();
    }
    if (i < D && j < D) ob[(size_t)i * D + j] = acc;
}

static bool g_hj_lds_set = false;

int eigh_onesided(const double* A, int nb, int D, int fn, double* out, double* eigvals, void* ws, hipStream_t st) {
    const size_t lds = (size_t)D * (D + 1) * sizeof(double);
    if (lds > 65536 && !g_hj_lds_set) {
        // the kernel also holds 4 bytes of static LDS: ask for what D = 128 needs, not for the whole 160 KiB
        if (hipFuncSetAttribute((const void*)hj_sweep_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 129 * 8) != hipSuccess) {
            otvae_set_error("otvae_eigh_fn: cannot raise dynamic LDS limit");
            return OTVAE_ELAUNCH;
        }
        g_hj_lds_set = true;
    }
    if (D > 64)
        hj_sweep_kernel<8><<<nb, 512, lds, st>>>(A, D, ws);
    else
        hj_sweep_kernel<16><<<nb, 512, lds, st>>>(A, D, ws);
    OTVAE_CHECK_LAUNCH("otvae_eigh_fn(sweeps)");
    hj_vectors_kernel<<<dim3(cdiv(D, 4), nb), 256, 0, st>>>(D, ws);
    OTVAE_CHECK_LAUNCH("otvae_eigh_fn(vectors)");
    hj_finish_kernel<<<nb, 256, 0, st>>>(D, fn, ws, eigvals, out);
    OTVAE_CHECK_LAUNCH("otvae_eigh_fn(finish)");
    if (fn == 1 || fn == 2) {
        hj_product_kernel<<<dim3(cdiv(D, 16), cdiv(D, 16), nb), 256, 0, st>>>(D, ws, out);
        OTVAE_CHECK_LAUNCH("otvae_eigh_fn(product)");
    }
    return OTVAE_OK;
}
