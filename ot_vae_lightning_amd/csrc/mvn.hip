// Log-density of samples under N(mean, L L^T) (L lower triangular with a positive diagonal) or N(mean, diag(sigma^2)): the
// forward of GaussianModel(update_with_autograd=True) (reference ot/distribution_models/gaussian_model.py:52-55,76-93,125-128:
// nn.Parameters trained through the negative log-likelihood, covariance held as its Cholesky factor by the ExpScaleTril
// parametrisation) -- torch.distributions.MultivariateNormal(scale_tril=) / Independent(Normal) in the reference.
//   y = L^-1 (x - mean)            forward substitution, one lane per sample, L in LDS
//   lp = -1/2 |y|^2 - sum_i log L_ii - D/2 log 2 pi
// backward (g = d loss / d lp): qg_b = g_b L^-T y_b (back substitution); the caller forms
//   d mean = sum_b qg_b,  d x_b = -qg_b,  d L = tril(sum_b qg_b y_b^T) - (sum_b g_b) diag(1 / L_ii)
// with the library's GEMM.  fp64; problems [nb] x samples [B] x D <= 128.
#include "common.h"

#define MVN_MAX_D 128

template <bool DIAG, bool BWD>
__global__ __launch_bounds__(256) void mvn_solve_kernel(const double* __restrict__ xin, const double* __restrict__ mean,
                                                        const double* __restrict__ Lm, const double* __restrict__ gin, int B, int D,
                                                        double* __restrict__ out, double* __restrict__ lp) {
    extern __shared__ __align__(16) double sL[];  // L [D][D] (or sigma [D]) + mean [D]
    const int p = blockIdx.y;
    const double* Lp = Lm + (size_t)p * (DIAG ? D : D * D);
    const int nl = DIAG ? D : D * D;
    for (int e = threadIdx.x; e < nl; e += 256) sL[e] = Lp[e];
    double* sm = sL + nl;
    if (!BWD)
        for (int e = threadIdx.x; e < D; e += 256) sm[e] = mean[(size_t)p * D + e];
    __syncthreads();
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const double* xb = xin + ((size_t)p * B + b) * D;
    double* ob = out + ((size_t)p * B + b) * D;
    if constexpr (!BWD) {
        double ss = 0.0, ld = 0.0;
        if constexpr (DIAG) {
            for (int i = 0; i < D; ++i) {
                const double y = (xb[i] - sm[i]) / sL[i];
                ob[i] = y;
                ss = fma(y, y, ss);
                ld += log(sL[i]);
            }
        } else {
            for (int i = 0; i < D; ++i) {      // y_i = (r_i - sum_{j<i} L_ij y_j) / L_ii; y kept in the output row (L1-resident)
                double acc = xb[i] - sm[i];
                const double* Li = sL + (size_t)i * D;
                for (int j = 0; j < i; ++j) acc = fma(-Li[j], ob[j], acc);
                const double y = acc / Li[i];
                ob[i] = y;
                ss = fma(y, y, ss);
                ld += log(Li[i]);
            }
        }
        lp[(size_t)p * B + b] = -0.5 * ss - ld - 0.5 * (double)D * 1.8378770664093454835606594728112;  // log(2 pi)
    } else {
        const double g = gin[(size_t)p * B + b];
        if constexpr (DIAG) {
            for (int i = 0; i < D; ++i) ob[i] = g * xb[i] / sL[i];
        } else {
            for (int i = D - 1; i >= 0; --i) {  // q_i = (y_i - sum_{j>i} L_ji q_j) / L_ii
                double acc = xb[i];
                for (int j = i + 1; j < D; ++j) acc = fma(-sL[(size_t)j * D + i], ob[j], acc);
                ob[i] = acc / sL[(size_t)i * D + i];
            }
            for (int i = 0; i < D; ++i) ob[i] *= g;
        }
    }
}

static bool g_mvn_lds_set[2] = {false, false};

template <bool BWD>
static int mvn_launch(const double* x, const double* mean, const double* L, const double* g, int nb, int B, int D, int diag,
                      double* out, double* lp, hipStream_t st, const char* who) {
    OTVAE_REQUIRE(x && L && out && nb > 0 && B > 0 && D > 0, "%s: bad argument", who);
    if (D > MVN_MAX_D) {
        otvae_set_error("%s: D = %d > %d is not implemented", who, D, MVN_MAX_D);
        return OTVAE_EUNSUPPORTED;
    }
    const size_t lds = ((size_t)(diag ? D : D * D) + D) * sizeof(double);
    if (lds > 65536 && !g_mvn_lds_set[BWD]) {
        if (hipFuncSetAttribute((const void*)mvn_solve_kernel<false, BWD>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)((MVN_MAX_D * MVN_MAX_D + MVN_MAX_D) * sizeof(double))) != hipSuccess) {
            otvae_set_error("%s: cannot raise the dynamic LDS limit", who);
            return OTVAE_ELAUNCH;
        }
        g_mvn_lds_set[BWD] = true;
    }
    const dim3 grid(cdiv(B, 256), nb);
    if (diag)
        mvn_solve_kernel<true, BWD><<<grid, 256, lds, st>>>(x, mean, L, g, B, D, out, lp);
    else
        mvn_solve_kernel<false, BWD><<<grid, 256, lds, st>>>(x, mean, L, g, B, D, out, lp);
    OTVAE_CHECK_LAUNCH(who);
    return OTVAE_OK;
}

extern "C" int otvae_mvn_logprob_fwd(const double* x, const double* mean, const double* L, int nb, int B, int D, int diag, double* y,
                                     double* lp, void* stream) {
    OTVAE_REQUIRE(mean && lp, "otvae_mvn_logprob_fwd: bad argument");
    return mvn_launch<false>(x, mean, L, nullptr, nb, B, D, diag, y, lp, (hipStream_t)stream, "otvae_mvn_logprob_fwd");
}

extern "C" int otvae_mvn_logprob_bwd(const double* g, const double* y, const double* L, int nb, int B, int D, int diag, double* qg,
                                     void* stream) {
    OTVAE_REQUIRE(g, "otvae_mvn_logprob_bwd: bad argument");
    return mvn_launch<true>(y, nullptr, L, g, nb, B, D, diag, qg, nullptr, (hipStream_t)stream, "otvae_mvn_logprob_bwd");
}
