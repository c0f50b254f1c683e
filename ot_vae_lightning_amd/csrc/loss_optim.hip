// GaussianPrior re-parametrisation + closed-form KL (reference prior/gaussian.py:63-96, prior/base.py:74-78),
// the nelbo reduction of VAE.nelbo (model/vae.py:158-176) and the Adam update configured by
// VAE.configure_optimizers (model/vae.py:148-151), each fused into one pass over its data.
#include "common.h"

// ---- GaussianPrior ------------------------------------------------------------------------------------------
// h [B][S][2D]: mu = h[..., :D], log_var = h[..., D:]; one workgroup per sample.
__global__ __launch_bounds__(256) void gaussian_prior_fwd_kernel(const float* __restrict__ h, const float* __restrict__ eps,
                                                                 int S, int D, float coeff, float* __restrict__ z,
                                                                 float* __restrict__ loss) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    const int n = S * D;
    const float* hb = h + (size_t)b * S * 2 * D;
    float kl = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int s = i / D, d = i - s * D;
        const float mu = hb[(size_t)s * 2 * D + d];
        const float lv = hb[(size_t)s * 2 * D + D + d];
        const float sd = __expf(0.5f * lv);
        const float var = sd * sd;
        z[(size_t)b * n + i] = fmaf(eps[(size_t)b * n + i], sd, mu);
        kl += 0.5f * (mu * mu - __logf(var) + var - 1.f);
    }
    kl = wave_sum(kl);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = kl;
    __syncthreads();
    if (threadIdx.x == 0) loss[b] = coeff * ((red[0] + red[1]) + (red[2] + red[3]));
}

__global__ __launch_bounds__(256) void gaussian_prior_bwd_kernel(const float* __restrict__ h, const float* __restrict__ eps,
                                                                 const float* __restrict__ gz, const float* __restrict__ gloss,
                                                                 int S, int D, float coeff, float* __restrict__ gh) {
    const int b = blockIdx.x;
    const int n = S * D;
    const float* hb = h + (size_t)b * S * 2 * D;
    float* gb = gh + (size_t)b * S * 2 * D;
    const float gl = (gloss ? gloss[b] : 0.f) * coeff;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int s = i / D, d = i - s * D;
        const float mu = hb[(size_t)s * 2 * D + d];
        const float lv = hb[(size_t)s * 2 * D + D + d];
        const float sd = __expf(0.5f * lv);
        const float g = gz ? gz[(size_t)b * n + i] : 0.f;
        gb[(size_t)s * 2 * D + d] = fmaf(gl, mu, g);
        // d z/d lv = eps*sd/2 ; d KL/d lv = (var - 1)/2
        gb[(size_t)s * 2 * D + D + d] = 0.5f * (g * eps[(size_t)b * n + i] * sd + gl * (sd * sd - 1.f));
    }
}

extern "C" int otvae_gaussian_prior_fwd(const float* h, const float* eps, int B, int S, int D, float coeff, float* z,
                                        float* loss, void* stream) {
    OTVAE_REQUIRE(h && eps && z && loss && B > 0 && S > 0 && D > 0, "otvae_gaussian_prior_fwd: bad argument");
    gaussian_prior_fwd_kernel<<<B, 256, 0, (hipStream_t)stream>>>(h, eps, S, D, coeff, z, loss);
    OTVAE_CHECK_LAUNCH("otvae_gaussian_prior_fwd");
    return OTVAE_OK;
}

extern "C" int otvae_gaussian_prior_bwd(const float* h, const float* eps, const float* gz, const float* gloss, int B, int S,
                                        int D, float coeff, float* gh, void* stream) {
    OTVAE_REQUIRE(h && eps && gh && B > 0 && S > 0 && D > 0, "otvae_gaussian_prior_bwd: bad argument");
    gaussian_prior_bwd_kernel<<<B, 256, 0, (hipStream_t)stream>>>(h, eps, gz, gloss, S, D, coeff, gh);
    OTVAE_CHECK_LAUNCH("otvae_gaussian_prior_bwd");
    return OTVAE_OK;
}

// ---- GaussianPrior options (prior/gaussian.py:63-96, prior/base.py:65-68): mode bit 0 = empirical_kl (the Monte-Carlo estimate
// sum log q(z) - log p(z) at the drawn z instead of the closed form), bit 1 = fixed_var (q = N(h, s), s = 1 or the per-sample
// temperature + 1e-8; h carries no log-variance half: [B][S][D]).
//   closed form:  0.5 (mu^2 - log s^2 + s^2 - 1)            empirical:  0.5 z^2 - 0.5 eps^2 - log s,   z = mu + s eps
__global__ __launch_bounds__(256) void gaussian_prior_ex_fwd_kernel(const float* __restrict__ h, const float* __restrict__ eps,
                                                                    const float* __restrict__ temp, int S, int D, float coeff, int mode,
                                                                    float* __restrict__ z, float* __restrict__ loss) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    const int n = S * D;
    const bool emp = mode & 1, fixed = mode & 2;
    const float* hb = h + (size_t)b * S * (fixed ? D : 2 * D);
    const float sfix = temp ? temp[b] + 1e-8f : 1.f;
    float kl = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int s = i / D, d = i - s * D;
        float mu, sd, lsd;  // mean, standard deviation, its logarithm
        if (fixed) {
            mu = hb[i];
            sd = sfix;
            lsd = logf(sfix);
        } else {
            mu = hb[(size_t)s * 2 * D + d];
            lsd = 0.5f * hb[(size_t)s * 2 * D + D + d];
            sd = __expf(lsd);
        }
        const float e = eps[(size_t)b * n + i];
        const float zz = fmaf(e, sd, mu);
        z[(size_t)b * n + i] = zz;
        kl += emp ? 0.5f * (zz * zz - e * e) - lsd : 0.5f * (mu * mu + sd * sd - 1.f) - lsd;
    }
    kl = wave_sum(kl);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = kl;
    __syncthreads();
    if (threadIdx.x == 0) loss[b] = coeff * ((red[0] + red[1]) + (red[2] + red[3]));
}

__global__ __launch_bounds__(256) void gaussian_prior_ex_bwd_kernel(const float* __restrict__ h, const float* __restrict__ eps,
                                                                    const float* __restrict__ temp, const float* __restrict__ gz,
                                                                    const float* __restrict__ gloss, int S, int D, float coeff,
                                                                    int mode, float* __restrict__ gh) {
    const int b = blockIdx.x;
    const int n = S * D;
    const bool emp = mode & 1, fixed = mode & 2;
    const size_t row = (size_t)S * (fixed ? D : 2 * D);
    const float* hb = h + (size_t)b * row;
    float* gb = gh + (size_t)b * row;
    const float sfix = temp ? temp[b] + 1e-8f : 1.f;
    const float gl = (gloss ? gloss[b] : 0.f) * coeff;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int s = i / D, d = i - s * D;
        const float e = eps[(size_t)b * n + i];
        const float g = gz ? gz[(size_t)b * n + i] : 0.f;
        if (fixed) {
            const float mu = hb[i];
            gb[i] = fmaf(gl, emp ? fmaf(e, sfix, mu) : mu, g);
        } else {
            const float mu = hb[(size_t)s * 2 * D + d];
            const float sd = __expf(0.5f * hb[(size_t)s * 2 * D + D + d]);
            const float zz = fmaf(e, sd, mu);
            // d z / d lv = eps sd / 2;  closed form: d KL / d mu = mu, d KL / d lv = (sd^2 - 1) / 2
            // empirical: d L / d mu = z,  d L / d lv = z eps sd / 2 - 1/2
            gb[(size_t)s * 2 * D + d] = fmaf(gl, emp ? zz : mu, g);
            gb[(size_t)s * 2 * D + D + d] = 0.5f * (g * e * sd + gl * (emp ? zz * e * sd - 1.f : sd * sd - 1.f));
        }
    }
}

extern "C" int otvae_gaussian_prior_ex_fwd(const float* h, const float* eps, const float* temp, int B, int S, int D, float coeff,
                                           int mode, float* z, float* loss, void* stream) {
    OTVAE_REQUIRE(h && eps && z && loss && B > 0 && S > 0 && D > 0 && mode >= 0 && mode <= 3, "otvae_gaussian_prior_ex_fwd: bad argument");
    OTVAE_REQUIRE(!temp || (mode & 2), "otvae_gaussian_prior_ex_fwd: a temperature goes with fixed_var (mode bit 1)");
    gaussian_prior_ex_fwd_kernel<<<B, 256, 0, (hipStream_t)stream>>>(h, eps, temp, S, D, coeff, mode, z, loss);
    OTVAE_CHECK_LAUNCH("otvae_gaussian_prior_ex_fwd");
    return OTVAE_OK;
}

extern "C" int otvae_gaussian_prior_ex_bwd(const float* h, const float* eps, const float* temp, const float* gz, const float* gloss,
                                           int B, int S, int D, float coeff, int mode, float* gh, void* stream) {
    OTVAE_REQUIRE(h && eps && gh && B > 0 && S > 0 && D > 0 && mode >= 0 && mode <= 3, "otvae_gaussian_prior_ex_bwd: bad argument");
    OTVAE_REQUIRE(!temp || (mode & 2), "otvae_gaussian_prior_ex_bwd: a temperature goes with fixed_var (mode bit 1)");
    gaussian_prior_ex_bwd_kernel<<<B, 256, 0, (hipStream_t)stream>>>(h, eps, temp, gz, gloss, S, D, coeff, mode, gh);
    OTVAE_CHECK_LAUNCH("otvae_gaussian_prior_ex_bwd");
    return OTVAE_OK;
}

// ---- ConditionalGaussianPrior (prior/conditional_gaussian.py:84-93): KL(q || p_y) against a per-sample diagonal prior ----
// h [B][2n] (mu | log_var), eps / z [B][n], prior mean pm and log standard deviation pl [B][n] (rows gathered by label):
//   KL = sum_i  pl_i - lv_i/2 + (exp(lv_i) + (mu_i - pm_i)^2) / (2 exp(2 pl_i)) - 1/2
__global__ __launch_bounds__(256) void gaussian_prior_cond_fwd_kernel(const float* __restrict__ h, const float* __restrict__ eps,
                                                                      const float* __restrict__ pm, const float* __restrict__ pl,
                                                                      int n, float coeff, float* __restrict__ z,
                                                                      float* __restrict__ loss) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    const float* hb = h + (size_t)b * 2 * n;
    float kl = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float mu = hb[i], lv = hb[n + i];
        const float sd = __expf(0.5f * lv), var = sd * sd;
        const float dm = mu - pm[(size_t)b * n + i], l = pl[(size_t)b * n + i];
        z[(size_t)b * n + i] = fmaf(eps[(size_t)b * n + i], sd, mu);
        kl += l - 0.5f * lv + 0.5f * (var + dm * dm) * __expf(-2.f * l) - 0.5f;
    }
    kl = wave_sum(kl);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = kl;
    __syncthreads();
    if (threadIdx.x == 0) loss[b] = coeff * ((red[0] + red[1]) + (red[2] + red[3]));
}

__global__ __launch_bounds__(256) void gaussian_prior_cond_bwd_kernel(const float* __restrict__ h, const float* __restrict__ eps,
                                                                      const float* __restrict__ pm, const float* __restrict__ pl,
                                                                      const float* __restrict__ gz, const float* __restrict__ gloss,
                                                                      int n, float coeff, float* __restrict__ gh,
                                                                      float* __restrict__ gpm, float* __restrict__ gpl) {
    const int b = blockIdx.x;
    const float* hb = h + (size_t)b * 2 * n;
    float* gb = gh + (size_t)b * 2 * n;
    const float gl = (gloss ? gloss[b] : 0.f) * coeff;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float mu = hb[i], lv = hb[n + i];
        const float sd = __expf(0.5f * lv), var = sd * sd;
        const float dm = mu - pm[(size_t)b * n + i], l = pl[(size_t)b * n + i];
        const float ip = __expf(-2.f * l);  // 1 / sigma_p^2
        const float g = gz ? gz[(size_t)b * n + i] : 0.f;
        gb[i] = fmaf(gl, dm * ip, g);
        gb[n + i] = 0.5f * (g * eps[(size_t)b * n + i] * sd + gl * (var * ip - 1.f));
        if (gpm) gpm[(size_t)b * n + i] = -gl * dm * ip;
        if (gpl) gpl[(size_t)b * n + i] = gl * (1.f - (var + dm * dm) * ip);
    }
}

extern "C" int otvae_gaussian_prior_cond_fwd(const float* h, const float* eps, const float* prior_mean, const float* prior_log_std,
                                             int B, int n, float coeff, float* z, float* loss, void* stream) {
    OTVAE_REQUIRE(h && eps && prior_mean && prior_log_std && z && loss && B > 0 && n > 0, "otvae_gaussian_prior_cond_fwd: bad argument");
    gaussian_prior_cond_fwd_kernel<<<B, 256, 0, (hipStream_t)stream>>>(h, eps, prior_mean, prior_log_std, n, coeff, z, loss);
    OTVAE_CHECK_LAUNCH("otvae_gaussian_prior_cond_fwd");
    return OTVAE_OK;
}

extern "C" int otvae_gaussian_prior_cond_bwd(const float* h, const float* eps, const float* prior_mean, const float* prior_log_std,
                                             const float* gz, const float* gloss, int B, int n, float coeff, float* gh,
                                             float* g_prior_mean, float* g_prior_log_std, void* stream) {
    OTVAE_REQUIRE(h && eps && prior_mean && prior_log_std && gh && B > 0 && n > 0, "otvae_gaussian_prior_cond_bwd: bad argument");
    gaussian_prior_cond_bwd_kernel<<<B, 256, 0, (hipStream_t)stream>>>(h, eps, prior_mean, prior_log_std, gz, gloss, n, coeff, gh,
                                                                       g_prior_mean, g_prior_log_std);
    OTVAE_CHECK_LAUNCH("otvae_gaussian_prior_cond_bwd");
    return OTVAE_OK;
}

// ---- ConditionalGaussianPrior with the options it inherits (prior/conditional_gaussian.py:44-93 over prior/gaussian.py:58-96,
// prior/base.py:65-68): empirical_kl (mode bit 0), fixed_var (bit 1) and a re-parametrisation dimension other than 1.  h is
// [B][S][2 D] (fixed_var: [B][S][D]): within each of the S slices the first D entries are the means, the next D the log-variances
// (torch.chunk on dimension r of a contiguous tensor: S = the sizes in front of r, D = half of r's size times the sizes behind it).
// eps / z / pm / pl are [B][S * D].  With l = log sigma_p, ip = exp(-2 l), dm = mu - mu_p, lsd = log sigma_q (0 when fixed):
//   closed form   KL = l - lsd + (sigma_q^2 + dm^2) ip / 2 - 1/2
//   empirical     KL = log q(z) - log p(z) = ((z - mu_p)^2 ip - eps^2) / 2 - lsd + l,   z = mu + eps sigma_q
__global__ __launch_bounds__(256) void gaussian_prior_cond_ex_fwd_kernel(const float* __restrict__ h, const float* __restrict__ eps,
                                                                         const float* __restrict__ pm, const float* __restrict__ pl,
                                                                         int S, int D, float coeff, int mode, float* __restrict__ z,
                                                                         float* __restrict__ loss) {
    __shared__ float red[4];
    const int b = blockIdx.x, n = S * D;
    const bool emp = mode & 1, fixed = mode & 2;
    const float* hb = h + (size_t)b * S * (fixed ? D : 2 * D);
    float kl = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int s = i / D, d = i - s * D;
        const float mu = fixed ? hb[i] : hb[(size_t)s * 2 * D + d];
        const float lsd = fixed ? 0.f : 0.5f * hb[(size_t)s * 2 * D + D + d];
        const float sd = fixed ? 1.f : __expf(lsd);
        const float e = eps[(size_t)b * n + i];
        const float zz = fmaf(e, sd, mu);
        z[(size_t)b * n + i] = zz;
        const float l = pl[(size_t)b * n + i], ip = __expf(-2.f * l), mp = pm[(size_t)b * n + i];
        if (emp) {
            const float dz = zz - mp;
            kl += 0.5f * (dz * dz * ip - e * e) - lsd + l;
        } else {
            const float dm = mu - mp;
            kl += l - lsd + 0.5f * (sd * sd + dm * dm) * ip - 0.5f;
        }
    }
    kl = wave_sum(kl);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = kl;
    __syncthreads();
    if (threadIdx.x == 0) loss[b] = coeff * ((red[0] + red[1]) + (red[2] + red[3]));
}

__global__ __launch_bounds__(256) void gaussian_prior_cond_ex_bwd_kernel(const float* __restrict__ h, const float* __restrict__ eps,
                                                                         const float* __restrict__ pm, const float* __restrict__ pl,
                                                                         const float* __restrict__ gz, const float* __restrict__ gloss,
                                                                         int S, int D, float coeff, int mode, float* __restrict__ gh,
                                                                         float* __restrict__ gpm, float* __restrict__ gpl) {
    const int b = blockIdx.x, n = S * D;
    const bool emp = mode & 1, fixed = mode & 2;
    const size_t row = (size_t)S * (fixed ? D : 2 * D);
    const float* hb = h + (size_t)b * row;
    float* gb = gh + (size_t)b * row;
    const float gl = (gloss ? gloss[b] : 0.f) * coeff;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int s = i / D, d = i - s * D;
        const float mu = fixed ? hb[i] : hb[(size_t)s * 2 * D + d];
        const float lsd = fixed ? 0.f : 0.5f * hb[(size_t)s * 2 * D + D + d];
        const float sd = fixed ? 1.f : __expf(lsd);
        const float e = eps[(size_t)b * n + i];
        const float g = gz ? gz[(size_t)b * n + i] : 0.f;
        const float l = pl[(size_t)b * n + i], ip = __expf(-2.f * l), mp = pm[(size_t)b * n + i];
        float dmu, dlv, dmp, dl;   // d KL / d (mu, log_var, mu_p, l)
        if (emp) {
            const float dz = fmaf(e, sd, mu) - mp;
            dmu = dz * ip;
            dlv = 0.5f * (dz * ip * e * sd) - 0.5f;
            dmp = -dz * ip;
            dl = 1.f - dz * dz * ip;
        } else {
            const float dm = mu - mp;
            dmu = dm * ip;
            dlv = 0.5f * (sd * sd * ip - 1.f);
            dmp = -dm * ip;
            dl = 1.f - (sd * sd + dm * dm) * ip;
        }
        if (fixed) {
            gb[i] = fmaf(gl, dmu, g);
        } else {
            gb[(size_t)s * 2 * D + d] = fmaf(gl, dmu, g);
            gb[(size_t)s * 2 * D + D + d] = fmaf(gl, dlv, 0.5f * g * e * sd);
        }
        if (gpm) gpm[(size_t)b * n + i] = gl * dmp;
        if (gpl) gpl[(size_t)b * n + i] = gl * dl;
    }
}

extern "C" int otvae_gaussian_prior_cond_ex_fwd(const float* h, const float* eps, const float* prior_mean, const float* prior_log_std,
                                                int B, int S, int D, float coeff, int mode, float* z, float* loss, void* stream) {
    OTVAE_REQUIRE(h && eps && prior_mean && prior_log_std && z && loss && B > 0 && S > 0 && D > 0 && mode >= 0 && mode <= 3,
                  "otvae_gaussian_prior_cond_ex_fwd: bad argument");
    gaussian_prior_cond_ex_fwd_kernel<<<B, 256, 0, (hipStream_t)stream>>>(h, eps, prior_mean, prior_log_std, S, D, coeff, mode, z, loss);
    OTVAE_CHECK_LAUNCH("otvae_gaussian_prior_cond_ex_fwd");
    return OTVAE_OK;
}

extern "C" int otvae_gaussian_prior_cond_ex_bwd(const float* h, const float* eps, const float* prior_mean, const float* prior_log_std,
                                                const float* gz, const float* gloss, int B, int S, int D, float coeff, int mode,
                                                float* gh, float* g_prior_mean, float* g_prior_log_std, void* stream) {
    OTVAE_REQUIRE(h && eps && prior_mean && prior_log_std && gh && B > 0 && S > 0 && D > 0 && mode >= 0 && mode <= 3,
                  "otvae_gaussian_prior_cond_ex_bwd: bad argument");
    gaussian_prior_cond_ex_bwd_kernel<<<B, 256, 0, (hipStream_t)stream>>>(h, eps, prior_mean, prior_log_std, gz, gloss, S, D, coeff, mode,
                                                                          gh, g_prior_mean, g_prior_log_std);
    OTVAE_CHECK_LAUNCH("otvae_gaussian_prior_cond_ex_bwd");
    return OTVAE_OK;
}

// ---- nelbo ---------------------------------------------------------------------------------------------------
#define NELBO_PARTS 256
extern "C" int otvae_nelbo_ws(void) { return NELBO_PARTS + 8; }

__global__ __launch_bounds__(256) void nelbo_partial_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                            int64_t numel, double* __restrict__ ws) {
    __shared__ double red[4];
    double s = 0.0;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < numel; i += (int64_t)gridDim.x * 256) {
        const float d = pred[i] - target[i];
        s += (double)(d * d);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) ws[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void nelbo_final_kernel(const double* __restrict__ ws, int parts, int64_t numel,
                                                          const float* __restrict__ prior_loss, int B, float chw,
                                                          float* __restrict__ out) {
    __shared__ double red[4];
    __shared__ double pr[4];
    double s = 0.0, p = 0.0;
    for (int i = threadIdx.x; i < parts; i += 256) s += ws[i];
    if (prior_loss)
        for (int i = threadIdx.x; i < B; i += 256) p += (double)prior_loss[i];
    s = wave_sum(s);
    p = wave_sum(p);
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6] = s;
        pr[threadIdx.x >> 6] = p;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float recon = (float)(((red[0] + red[1]) + (red[2] + red[3])) / (double)numel);
        const float prior = (float)(((pr[0] + pr[1]) + (pr[2] + pr[3])) / (double)B) / chw;
        out[0] = recon + prior;
        out[1] = recon;
        out[2] = prior;
    }
}

extern "C" int otvae_nelbo_fwd(const float* pred, const float* target, int64_t numel, const float* prior_loss, int B,
                               float chw, double* ws, float* out, void* stream) {
    OTVAE_REQUIRE(pred && target && ws && out && numel > 0 && B > 0 && chw > 0, "otvae_nelbo_fwd: bad argument");
    const int parts = imin(NELBO_PARTS, cdiv(numel, 1024));
    hipStream_t st = (hipStream_t)stream;
    nelbo_partial_kernel<<<parts, 256, 0, st>>>(pred, target, numel, ws);
    OTVAE_CHECK_LAUNCH("otvae_nelbo_fwd(partial)");
    nelbo_final_kernel<<<1, 256, 0, st>>>(ws, parts, numel, prior_loss, B, chw, out);
    OTVAE_CHECK_LAUNCH("otvae_nelbo_fwd(final)");
    return OTVAE_OK;
}

__global__ __launch_bounds__(256) void nelbo_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                        int64_t numel, int B, float chw, const float* __restrict__ gout,
                                                        float* __restrict__ gpred, float* __restrict__ gprior) {
    // out = {total = recon + prior, recon, prior}: d/d recon = g[0]+g[1], d/d prior = g[0]+g[2]
    const float gr = gout ? gout[0] + gout[1] : 1.f;
    const float gp = gout ? gout[0] + gout[2] : 1.f;
    const float k = 2.f * gr / (float)numel;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < numel; i += (int64_t)gridDim.x * 256)
        gpred[i] = k * (pred[i] - target[i]);
    if (gprior && blockIdx.x == 0)
        for (int i = threadIdx.x; i < B; i += 256) gprior[i] = gp / ((float)B * chw);
}

extern "C" int otvae_nelbo_bwd(const float* pred, const float* target, int64_t numel, int B, float chw, const float* gout,
                               float* gpred, float* gprior, void* stream) {
    OTVAE_REQUIRE(pred && target && gpred && numel > 0 && B > 0, "otvae_nelbo_bwd: bad argument");
    nelbo_bwd_kernel<<<imin(cdiv(numel, 256), 2048), 256, 0, (hipStream_t)stream>>>(pred, target, numel, B, chw, gout, gpred,
                                                                                 gprior);
    OTVAE_CHECK_LAUNCH("otvae_nelbo_bwd");
    return OTVAE_OK;
}

// ---- gradients that reached their parameter through plain autograd (embeddings, learned tokens ...) into their slots of the flat
// gradient buffer: up to 32 (source, slot, length) triples per launch, passed by value (no device-side table: nothing to upload inside
// a captured step).  One launch instead of one copy per parameter (engine.HipTrainer._collect_loose_grads).
#define COPYB_MAX 32
struct CopyBatch {
    const float* src[COPYB_MAX];
    float* dst[COPYB_MAX];
    long long n[COPYB_MAX];
};

__global__ __launch_bounds__(256) void copy_batched_kernel(CopyBatch d) {
    const float* __restrict__ s = d.src[blockIdx.y];
    float* __restrict__ t = d.dst[blockIdx.y];
    const long long n = d.n[blockIdx.y];
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) t[i] = s[i];
}

extern "C" int otvae_copy_batched(int count, const float* const* src, float* const* dst, const int64_t* n, void* stream) {
    OTVAE_REQUIRE(count > 0 && src && dst && n, "otvae_copy_batched: bad argument");
    for (int i0 = 0; i0 < count; i0 += COPYB_MAX) {
        CopyBatch d;
        const int c = imin(COPYB_MAX, count - i0);
        long long longest = 0;
        for (int i = 0; i < c; ++i) {
            OTVAE_REQUIRE(src[i0 + i] && dst[i0 + i] && n[i0 + i] > 0, "otvae_copy_batched: entry %d", i0 + i);
            d.src[i] = src[i0 + i];
            d.dst[i] = dst[i0 + i];
            d.n[i] = n[i0 + i];
            longest = n[i0 + i] > longest ? n[i0 + i] : longest;
        }
        copy_batched_kernel<<<dim3(imin(cdiv(longest, 1024), 256), c), 256, 0, (hipStream_t)stream>>>(d);
        OTVAE_CHECK_LAUNCH("otvae_copy_batched");
    }
    return OTVAE_OK;
}

// ---- Adam ----------------------------------------------------------------------------------------------------
__global__ void step_begin_kernel(int32_t* step) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *step += 1;
}

extern "C" int otvae_step_begin(int32_t* step, void* stream) {
    OTVAE_REQUIRE(step, "otvae_step_begin: NULL step");
    step_begin_kernel<<<1, 64, 0, (hipStream_t)stream>>>(step);
    OTVAE_CHECK_LAUNCH("otvae_step_begin");
    return OTVAE_OK;
}

// torch_ema's three roundings, s - ((s - p) * omd), with contraction into a fused multiply-add switched off (HIP's round-to-nearest
// intrinsics are the plain operators, which -ffp-contract=fast fuses: one rounding fewer than the package's tensor operations make)
__device__ __forceinline__ float ema_step(float s, float p, float omd) {
#pragma clang fp contract(off)
    const float d = s - p;
    const float t = d * omd;
    return s - t;
}

// torch.optim.Adam (no weight decay / amsgrad): m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ;
// p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// Step guard, running buffers.  A NaN that reaches a BatchNorm's input does not stay a NaN: the next layer's ReLU (fmaxf) turns the
// NaN-normalised tensor into zeros, so later layers see finite -- and meaningless -- batch statistics and would fold them into
// their running buffers.  The guarded step therefore keeps a copy of all running buffers (one flat fp32 range) from the start
// of the step (otvae_step_begin_guarded) and the guarded Adam kernel puts it back when it refuses the step.
__global__ __launch_bounds__(256) void step_begin_guarded_kernel(int32_t* step, const float* __restrict__ state, float* __restrict__ backup,
                                                                 int64_t n) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *step += 1;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) backup[i] = state[i];
}

extern "C" int otvae_step_begin_guarded(int32_t* step, const float* state, float* backup, int64_t n, void* stream) {
    OTVAE_REQUIRE(step && (n == 0 || (state && backup)) && n >= 0, "otvae_step_begin_guarded: bad argument");
    step_begin_guarded_kernel<<<imax(1, imin(cdiv(n, 1024), 1024)), 256, 0, (hipStream_t)stream>>>(step, state, backup, n);
    OTVAE_CHECK_LAUNCH("otvae_step_begin_guarded");
    return OTVAE_OK;
}

// Step guard (guard != NULL): the update is applied only when every watched device scalar is finite -- the step's loss
// (a starved Sinkhorn solve poisons it with NaN, csrc/sinkhorn.hip: sk_finish) and, when the gradient norm was reduced
// (otvae_grad_clip_coef), that norm.  Every block evaluates the same two scalars, so the decision is uniform without a flag
// kernel.  A skipped step leaves p, m, v untouched, takes the step counter back (Adam's bias correction must not advance) and
// counts itself in guard[0]; guard[1] holds the step number of the last skip.
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, const float* __restrict__ hyper,
                                                   int32_t* __restrict__ step, float grad_scale,
                                                   const float* __restrict__ scale_dev, int32_t* __restrict__ guard,
                                                   const float* __restrict__ watch_loss, float* __restrict__ state,
                                                   const float* __restrict__ backup, int64_t n_state,
                                                   float* __restrict__ ema = nullptr, double ema_decay = 0.0) {
    if (scale_dev) grad_scale = *scale_dev;  // clip coefficient x 1/world, left by grad_clip_final_kernel
    if (guard) {
        bool ok = isfinite(grad_scale);
        if (scale_dev) ok = ok && isfinite(scale_dev[1]);      // the gradient norm
        if (watch_loss) ok = ok && isfinite(watch_loss[0]);
        if (!ok) {
            // every block has read *step-free state only; the counter is taken back by one thread of the grid
            if (blockIdx.x == 0 && threadIdx.x == 0) {
                guard[0] += 1;
                guard[1] = *step;
                *step -= 1;
            }
            for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n_state; i += (int64_t)gridDim.x * 256) state[i] = backup[i];
            return;
        }
    }
    const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3];
    const int t = *step;
    // Parameter moving average (the reference's `ema_decay`, model/base.py:153-190, kept there by the third-party torch_ema package:
    // shadow -= (1 - d) (shadow - p), d = min(decay, (1 + n) / (10 + n)), n = updates so far incl. this one) folded into the
    // optimizer's pass: one update per ACCEPTED optimizer step, so n is the device step counter t; the three roundings are
    // torch_ema's own (sub, mul by the fp32 image of 1 - d, sub: no fused multiply-add)
    float ema_omd = 0.f;
    if (ema) {
        const double d = fmin(ema_decay, (1.0 + (double)t) / (10.0 + (double)t));
        ema_omd = (float)(1.0 - d);
    }
    const float bc1 = 1.f - powf(b1, (float)t);
    const float bc2s = sqrtf(1.f - powf(b2, (float)t));
    const float step_size = lr / bc1;
    const int64_t n4 = n >> 2;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 pv = reinterpret_cast<float4*>(p)[i];
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        float4 mv = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
#define ADAM1(f)                                            \
    {                                                       \
        const float gg = gv.f * grad_scale;                 \
        mv.f = fmaf(b1, mv.f, (1.f - b1) * gg);             \
        vv.f = fmaf(b2, vv.f, (1.f - b2) * gg * gg);        \
        pv.f -= step_size * mv.f / (sqrtf(vv.f) / bc2s + eps); \
    }
        ADAM1(x) ADAM1(y) ADAM1(z) ADAM1(w)
#undef ADAM1
        reinterpret_cast<float4*>(p)[i] = pv;
        reinterpret_cast<float4*>(m)[i] = mv;
        reinterpret_cast<float4*>(v)[i] = vv;
        if (ema) {
            float4 sv = reinterpret_cast<float4*>(ema)[i];
            sv.x = ema_step(sv.x, pv.x, ema_omd);
            sv.y = ema_step(sv.y, pv.y, ema_omd);
            sv.z = ema_step(sv.z, pv.z, ema_omd);
            sv.w = ema_step(sv.w, pv.w, ema_omd);
            reinterpret_cast<float4*>(ema)[i] = sv;
        }
    }
    // tail
    for (int64_t i = (n4 << 2) + blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float gg = g[i] * grad_scale;
        const float mm = fmaf(b1, m[i], (1.f - b1) * gg);
        const float vv = fmaf(b2, v[i], (1.f - b2) * gg * gg);
        m[i] = mm;
        v[i] = vv;
        const float pn = p[i] - step_size * mm / (sqrtf(vv) / bc2s + eps);
        p[i] = pn;
        if (ema) ema[i] = ema_step(ema[i], pn, ema_omd);
    }
}

// shadow -= (1 - d) (shadow - p) with the caller's effective decay d (the host-driven route: a stock optimizer stepped by the
// reference's own loop; `ParamEMA.update`, engine/ema.py)
__global__ __launch_bounds__(256) void ema_update_kernel(float* __restrict__ shadow, const float* __restrict__ p, int64_t n, float omd) {
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        shadow[i] = ema_step(shadow[i], p[i], omd);
}

extern "C" int otvae_ema_update(float* shadow, const float* p, int64_t n, double decay, void* stream) {
    OTVAE_REQUIRE(shadow && p && n > 0 && decay >= 0.0 && decay <= 1.0, "otvae_ema_update: bad argument");
    ema_update_kernel<<<imin(cdiv(n, 1024), 2048), 256, 0, (hipStream_t)stream>>>(shadow, p, n, (float)(1.0 - decay));
    OTVAE_CHECK_LAUNCH("otvae_ema_update");
    return OTVAE_OK;
}

extern "C" int otvae_adam_step(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper,
                               const int32_t* step, float grad_scale, void* stream) {
    OTVAE_REQUIRE(p && g && m && v && hyper && step && n > 0, "otvae_adam_step: bad argument");
    OTVAE_REQUIRE(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)m % 16 == 0) && ((uintptr_t)v % 16 == 0),
                  "otvae_adam_step: buffers must be 16-byte aligned");
    adam_kernel<<<imin(cdiv(n, 1024), 2048), 256, 0, (hipStream_t)stream>>>(p, g, m, v, n, hyper, const_cast<int32_t*>(step), grad_scale,
                                                                            nullptr, nullptr, nullptr, nullptr, nullptr, 0);
    OTVAE_CHECK_LAUNCH("otvae_adam_step");
    return OTVAE_OK;
}

extern "C" int otvae_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper,
                                   const int32_t* step, const float* grad_scale_dev, void* stream) {
    OTVAE_REQUIRE(p && g && m && v && hyper && step && grad_scale_dev && n > 0, "otvae_adam_step_dev: bad argument");
    OTVAE_REQUIRE(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)m % 16 == 0) && ((uintptr_t)v % 16 == 0),
                  "otvae_adam_step_dev: buffers must be 16-byte aligned");
    adam_kernel<<<imin(cdiv(n, 1024), 2048), 256, 0, (hipStream_t)stream>>>(p, g, m, v, n, hyper, const_cast<int32_t*>(step), 1.f,
                                                                            grad_scale_dev, nullptr, nullptr, nullptr, nullptr, 0);
    OTVAE_CHECK_LAUNCH("otvae_adam_step_dev");
    return OTVAE_OK;
}

extern "C" int otvae_adam_step_ema(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, int32_t* step,
                                   float grad_scale, const float* grad_scale_dev, const float* watch_loss, int32_t* guard,
                                   float* state, const float* backup, int64_t n_state, float* ema_shadow, double ema_decay,
                                   void* stream) {
    OTVAE_REQUIRE(p && g && m && v && hyper && step && n > 0, "otvae_adam_step_ema: bad argument");
    OTVAE_REQUIRE(n_state >= 0 && (n_state == 0 || (state && backup && guard)), "otvae_adam_step_ema: state / backup / guard missing");
    OTVAE_REQUIRE(!watch_loss || guard, "otvae_adam_step_ema: a watched loss needs the guard counters");
    OTVAE_REQUIRE(!ema_shadow || (ema_decay >= 0.0 && ema_decay <= 1.0), "otvae_adam_step_ema: ema_decay must lie in [0, 1]");
    OTVAE_REQUIRE(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)m % 16 == 0) && ((uintptr_t)v % 16 == 0) &&
                      (!ema_shadow || (uintptr_t)ema_shadow % 16 == 0),
                  "otvae_adam_step_ema: buffers must be 16-byte aligned");
    adam_kernel<<<imin(cdiv(n, 1024), 2048), 256, 0, (hipStream_t)stream>>>(p, g, m, v, n, hyper, step, grad_scale, grad_scale_dev, guard,
                                                                            watch_loss, state, backup, n_state, ema_shadow, ema_decay);
    OTVAE_CHECK_LAUNCH("otvae_adam_step_ema");
    return OTVAE_OK;
}

extern "C" int otvae_adam_step_guarded(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, int32_t* step,
                                       float grad_scale, const float* grad_scale_dev, const float* watch_loss, int32_t* guard,
                                       float* state, const float* backup, int64_t n_state, void* stream) {
    OTVAE_REQUIRE(p && g && m && v && hyper && step && guard && n > 0, "otvae_adam_step_guarded: bad argument");
    OTVAE_REQUIRE(n_state >= 0 && (n_state == 0 || (state && backup)), "otvae_adam_step_guarded: state / backup missing");
    OTVAE_REQUIRE(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)m % 16 == 0) && ((uintptr_t)v % 16 == 0),
                  "otvae_adam_step_guarded: buffers must be 16-byte aligned");
    adam_kernel<<<imin(cdiv(n, 1024), 2048), 256, 0, (hipStream_t)stream>>>(p, g, m, v, n, hyper, step, grad_scale, grad_scale_dev, guard,
                                                                            watch_loss, state, backup, n_state);
    OTVAE_CHECK_LAUNCH("otvae_adam_step_guarded");
    return OTVAE_OK;
}

// ---- global-norm gradient clipping (reference configs/ddp.yaml:4 `gradient_clip_val: 1.0` -> Lightning ->
// torch.nn.utils.clip_grad_norm_: coef = min(1, max_norm / (|g|_2 + 1e-6)), g *= coef) over the flat gradient buffer.
// g holds the SUM over ranks; the gradient that is clipped is g * grad_scale (the mean).  The scale Adam then applies
// to g is grad_scale * coef: no pass that rewrites g.
#define CLIP_PARTS 512
__global__ __launch_bounds__(256) void grad_sqnorm_partial_kernel(const float* __restrict__ g, int64_t n, double* __restrict__ ws) {
    __shared__ double red[4];
    double s = 0.0;
    const int64_t n4 = n >> 2;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 q = reinterpret_cast<const float4*>(g)[i];
        s += (double)(q.x * q.x + q.y * q.y) + (double)(q.z * q.z + q.w * q.w);
    }
    for (int64_t i = (n4 << 2) + blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += (double)(g[i] * g[i]);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) ws[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void grad_clip_final_kernel(const double* __restrict__ ws, int parts, float grad_scale,
                                                              float max_norm, float* __restrict__ out) {
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < parts; i += 256) s += ws[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt((red[0] + red[1]) + (red[2] + red[3])) * grad_scale;
        float coef = max_norm / (norm + 1e-6f);
        coef = coef < 1.f ? coef : 1.f;
        if (!(max_norm > 0.f)) coef = 1.f;  // max_norm <= 0: report the norm only
        out[0] = grad_scale * coef;
        out[1] = norm;
    }
}

extern "C" int otvae_grad_clip_ws(void) { return CLIP_PARTS; }

extern "C" int otvae_grad_clip_coef(const float* g, int64_t n, float grad_scale, float max_norm, double* ws, float* out,
                                    void* stream) {
    OTVAE_REQUIRE(g && ws && out && n > 0, "otvae_grad_clip_coef: bad argument");
    OTVAE_REQUIRE((uintptr_t)g % 16 == 0, "otvae_grad_clip_coef: the gradient buffer must be 16-byte aligned");
    const int parts = imin(CLIP_PARTS, cdiv(n, 4096));
    hipStream_t st = (hipStream_t)stream;
    grad_sqnorm_partial_kernel<<<parts, 256, 0, st>>>(g, n, ws);
    OTVAE_CHECK_LAUNCH("otvae_grad_clip_coef(partial)");
    grad_clip_final_kernel<<<1, 256, 0, st>>>(ws, parts, grad_scale, max_norm, out);
    OTVAE_CHECK_LAUNCH("otvae_grad_clip_coef(final)");
    return OTVAE_OK;
}

// ---- start of a step, round 4: the step counter, the step guard's backup of the running state (n may be 0) and the zeroing of the
// BatchNorm statistic slots the step's kernels will add into (functional.SlotArena; zero_words int64 words, 16-byte aligned, may be 0)
// as ONE launch
__global__ __launch_bounds__(256) void step_begin_slots_kernel(int32_t* step, const float* __restrict__ state, float* __restrict__ backup,
                                                               int64_t n, int4* __restrict__ zero, int64_t zero_n16) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *step += 1;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) backup[i] = state[i];
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < zero_n16; i += (int64_t)gridDim.x * 256) zero[i] = make_int4(0, 0, 0, 0);
}

extern "C" int otvae_step_begin_slots(int32_t* step, const float* state, float* backup, int64_t n, void* zero, int64_t zero_words,
                                      void* stream) {
    OTVAE_REQUIRE(step && (n == 0 || (state && backup)) && n >= 0, "otvae_step_begin_slots: bad argument");
    OTVAE_REQUIRE(zero_words >= 0 && (zero_words == 0 || (zero && ((uintptr_t)zero & 15) == 0 && zero_words % 2 == 0)),
                  "otvae_step_begin_slots: the range to zero must be 16-byte aligned, an even number of int64 words");
    const int64_t work = n > zero_words / 2 ? n : zero_words / 2;
    step_begin_slots_kernel<<<imax(1, imin(cdiv(work, 1024), 1024)), 256, 0, (hipStream_t)stream>>>(step, state, backup, n, (int4*)zero,
                                                                                                    zero_words / 2);
    OTVAE_CHECK_LAUNCH("otvae_step_begin_slots");
    return OTVAE_OK;
}

__global__ __launch_bounds__(256) void zero_words_kernel(int4* __restrict__ zero, int64_t n16) {
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) zero[i] = make_int4(0, 0, 0, 0);
}

// zero_words int64 words (16-byte aligned, an even count): the slots of a pass that has no step-begin launch of its own
extern "C" int otvae_zero_words(void* zero, int64_t zero_words, void* stream) {
    OTVAE_REQUIRE(zero && zero_words > 0 && ((uintptr_t)zero & 15) == 0 && zero_words % 2 == 0,
                  "otvae_zero_words: the range must be 16-byte aligned, an even number of int64 words");
    zero_words_kernel<<<imin(cdiv(zero_words / 2, 1024), 1024), 256, 0, (hipStream_t)stream>>>((int4*)zero, zero_words / 2);
    OTVAE_CHECK_LAUNCH("otvae_zero_words");
    return OTVAE_OK;
}
