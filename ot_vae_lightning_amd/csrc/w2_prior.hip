// Gaussian W2 with empirical covariance as a differentiable prior term (BASELINE north_star): the tail of the
// forward pass and the backward pass of
//   L(z) = |mu - mu_t|^2 + tr(S + S_t - 2 (S_t^1/2 S S_t^1/2)^1/2),   mu, S = mean_cov(GaussianModel._stats(z))
// (reference ot/w2_utils.py:40-80, ot/matrix_utils.py:37-46,145-158, ot/distribution_models/gaussian_model.py:144-157).
// The statistics, the covariance, the products with S_t^1/2 and the eigendecomposition of M = S_t^1/2 S S_t^1/2 are the
// library's existing fp64 kernels (gaussian_ot.hip); here:
//   tail:     L = |mu - mu_t|^2 + tr S + tr S_t - 2 sum_k sqrt(lambda_k(M))  and  Q = diag(lambda^-1/4) V^T  (so that
//             M^-1/2 = Q^T Q is one product)
//   backward: dL/dz_i = (2/n) [ (mu - mu_t) + (z_i - mu) - W (z_i - mu) ],  W = S_t^1/2 M^-1/2 S_t^1/2
//             -- the eigh-based sqrtm adjoint (Daleckii-Krein) of the reference's autograd path collapses to this closed
//             form under the trace (d tr f(M) = tr f'(M) dM); parity with the reference under torch.autograd is held to 1e-7
//             by tests/golden/w2_prior.npz.
#include "common.h"

// make_pd semantics of w2_gaussian's argument validation (w2_utils.py:661-669): a covariance that fails the strict eigenvalue
// test is shifted by |min(lambda_min, 0)| + 1e-8.  With the identity target M = S and lambda are S's eigenvalues, so the
// shift (if any) applies to S itself: its trace and M's spectrum move together.  With a general target the inner product is
// validated as 'spsd', which the reference never shifts.
__global__ __launch_bounds__(256) void w2_prior_tail_kernel(const double* __restrict__ mu, const double* __restrict__ mut,
                                                            const double* __restrict__ cov, const double* __restrict__ covt,
                                                            const double* __restrict__ lam, const double* __restrict__ vt, int D,
                                                            int source_is_m, double scale, int rep_n, float* __restrict__ loss_out,
                                                            double* __restrict__ q) {
    __shared__ double red[4], red2[4];
    __shared__ double s_shift, s_floor;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // lambda_min -> shift (every block computes it: D reads); general target: the mean target variance for the floor below
    double lo = INFINITY, tv = 0.0;
    for (int i = threadIdx.x; i < D; i += 256) {
        lo = fmin(lo, lam[i]);
        if (covt) tv += covt[(size_t)i * D + i];
    }
    lo = wave_min(lo);
    tv = wave_sum(tv);
    if (lane == 0) {
        red[wv] = lo;
        red2[wv] = tv;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double m = fmin(fmin(red[0], red[1]), fmin(red[2], red[3]));
        s_shift = (source_is_m && !(m > 0.0)) ? fabs(fmin(m, 0.0)) + 1e-8 : 0.0;
        // General target: lambda are the eigenvalues of covt^1/2 S covt^1/2.  The reference validates S with make_pd (a rank-
        // deficient batch covariance, B <= D, gets |lambda_min| + 1e-8 on its diagonal), which lifts them by ~1e-8 x the target's
        // variances; without it lambda^-1/4 below is infinite and the captured step trains to NaN.  Clamped at that size.
        s_floor = source_is_m ? 0.0 : 1e-8 * ((red2[0] + red2[1]) + (red2[2] + red2[3])) / (double)D;
    }
    __syncthreads();
    const double shift = s_shift, floor_ = s_floor;
    // Q[k][:] = (lambda_k + shift)^-1/4 V^T[k][:]
    const size_t total = (size_t)D * D;
    for (size_t e = blockIdx.x * (size_t)256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int k = (int)(e / D);
        q[e] = vt[e] / sqrt(sqrt(fmax(lam[k] + shift, floor_)));
    }
    if (blockIdx.x != 0) return;
    double s = 0.0;
    for (int i = threadIdx.x; i < D; i += 256) {
        const double dm = mu[i] - (mut ? mut[i] : 0.0);
        const size_t dd = (size_t)i * D + i;
        s += dm * dm + (cov[dd] + shift) + (covt ? covt[dd] : 1.0) - 2.0 * sqrt(fmax(lam[i] + shift, floor_));
    }
    s = wave_sum(s);
    __syncthreads();
    if (lane == 0) red[wv] = s;
    __syncthreads();
    const float val = (float)(scale * ((red[0] + red[1]) + (red[2] + red[3])));
    for (int i = threadIdx.x; i < rep_n; i += 256) loss_out[i] = val;
}

extern "C" int otvae_w2_prior_tail(const double* mu, const double* mut, const double* cov, const double* covt, const double* lam,
                                   const double* vt, int D, double scale, int rep, float* loss, double* q, void* stream) {
    OTVAE_REQUIRE(mu && cov && lam && vt && loss && q && D > 0 && rep > 0, "otvae_w2_prior_tail: bad argument");
    w2_prior_tail_kernel<<<imin(cdiv((size_t)D * D, 2048), 256), 256, 0, (hipStream_t)stream>>>(mu, mut, cov, covt, lam, vt, D,
                                                                                            covt == nullptr, scale, rep, loss, q);
    OTVAE_CHECK_LAUNCH("otvae_w2_prior_tail");
    return OTVAE_OK;
}

// gz[i][d] = gadd[i][d] + c [ (mu_d - mut_d) + (z_id - mu_d) - sum_e W[d][e] (z_ie - mu_e) ],  c = 2 scale (sum_q g[q]) / B.
// 32 samples x 32 dimensions per workgroup, fp64 accumulation, W and the centred samples staged through LDS in slices of 32.
template <typename T>
__global__ __launch_bounds__(256) void w2_prior_bwd_kernel(const T* __restrict__ z, int B, int D, const double* __restrict__ mu,
                                                           const double* __restrict__ mut, const double* __restrict__ W,
                                                           const float* __restrict__ g, int ng, double scale,
                                                           const T* __restrict__ gadd, T* __restrict__ gz) {
    __shared__ double zs[32][33], ws[32][33];
    __shared__ double s_g;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int i0 = blockIdx.y * 32, d0 = blockIdx.x * 32;
    if (threadIdx.x < 64) {
        float gs = 0.f;
        for (int qd = threadIdx.x; qd < ng; qd += 64) gs += g[qd];
        gs = wave_sum(gs);
        if (threadIdx.x == 0) s_g = (double)gs;
    }
    double acc[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
    for (int e0 = 0; e0 < D; e0 += 32) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = threadIdx.x + 256 * u, r = idx >> 5, c = idx & 31;
            zs[r][c] = (i0 + r < B && e0 + c < D) ? (double)z[(size_t)(i0 + r) * D + e0 + c] - mu[e0 + c] : 0.0;   // [sample][e]
            ws[r][c] = (d0 + r < D && e0 + c < D) ? W[(size_t)(d0 + r) * D + e0 + c] : 0.0;                       // [d][e]
        }
        __syncthreads();
#pragma unroll 8
        for (int e = 0; e < 32; ++e) {
            const double z0 = zs[ty][e], z1 = zs[ty + 16][e], w0 = ws[tx][e], w1 = ws[tx + 16][e];
            acc[0][0] += z0 * w0;
            acc[0][1] += z0 * w1;
            acc[1][0] += z1 * w0;
            acc[1][1] += z1 * w1;
        }
    }
    const double c = 2.0 * scale * s_g / (double)B;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int i = i0 + ty + 16 * a, d = d0 + tx + 16 * b;
            if (i < B && d < D) {
                const size_t e = (size_t)i * D + d;
                const double v = c * ((mu[d] - (mut ? mut[d] : 0.0)) + ((double)z[e] - mu[d]) - acc[a][b]);
                gz[e] = (T)(gadd ? (double)gadd[e] + v : v);
            }
        }
}

extern "C" int otvae_w2_prior_bwd(int dtype, const void* z, int B, int D, const double* mu, const double* mut, const double* W,
                                  const float* g, int ng, double scale, const void* gadd, void* gz, void* stream) {
    OTVAE_REQUIRE(z && mu && W && g && gz && B > 0 && D > 0 && ng > 0, "otvae_w2_prior_bwd: bad argument");
    OTVAE_REQUIRE(dtype == 0 || dtype == 1, "otvae_w2_prior_bwd: dtype must be 0 (fp32) or 1 (fp64)");
    const dim3 grid(cdiv(D, 32), cdiv(B, 32));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0)
        w2_prior_bwd_kernel<float><<<grid, 256, 0, st>>>((const float*)z, B, D, mu, mut, W, g, ng, scale, (const float*)gadd, (float*)gz);
    else
        w2_prior_bwd_kernel<double><<<grid, 256, 0, st>>>((const double*)z, B, D, mu, mut, W, g, ng, scale, (const double*)gadd,
                                                          (double*)gz);
    OTVAE_CHECK_LAUNCH("otvae_w2_prior_bwd");
    return OTVAE_OK;
}
