// Grouped and dilated convolutions of ConvLayer (reference networks/cnn.py:66-67,103-104: nn.Conv2d(..., dilation, groups)) on the
// DENSE convolution kernels: the layer's weight [Cout][Cin / groups][KH][KW] is expanded into the dense weight the kernels take,
//     dense[ky * dil][kx * dil][ci][co] = w[co][ci - first(ci's group)][ky][kx]   if ci and co belong to the same group,
//     0                                                                            elsewhere (other groups, the holes of the dilation),
// on HWIO memory [(KH - 1) dil + 1][(KW - 1) dil + 1][Cin][Cout].  A product with an exact zero adds nothing, so the dense
// convolution returns what the grouped / dilated one does (the non-zero terms accumulate in the same tap-major, channel-minor
// order); it costs groups * dil^2 times the multiplications, on layers that are launch- and latency-bound.
// The backward kernel gathers the gradient of the dense weight back: d w[co][cig][ky][kx] = d dense[ky dil][kx dil][g(co) Cig + cig][co].
#include "common.h"

__global__ __launch_bounds__(256) void weight_expand_fwd_kernel(const float* __restrict__ w, int Cout, int Cin, int groups, int KH,
                                                                int KW, int dil, float* __restrict__ dense) {
    const int KHd = (KH - 1) * dil + 1, KWd = (KW - 1) * dil + 1;
    const int Cig = Cin / groups, Cog = Cout / groups;
    const int64_t total = (int64_t)KHd * KWd * Cin * Cout;
    for (int64_t e = blockIdx.x * (int64_t)256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int co = (int)(e % Cout);
        int64_t t = e / Cout;
        const int ci = (int)(t % Cin);
        t /= Cin;
        const int kxd = (int)(t % KWd), kyd = (int)(t / KWd);
        float v = 0.f;
        if (kyd % dil == 0 && kxd % dil == 0 && ci / Cig == co / Cog)
            v = w[(((size_t)co * Cig + (ci - (ci / Cig) * Cig)) * KH + kyd / dil) * KW + kxd / dil];
        dense[e] = v;
    }
}

__global__ __launch_bounds__(256) void weight_expand_bwd_kernel(const float* __restrict__ gdense, int Cout, int Cin, int groups, int KH,
                                                                int KW, int dil, float* __restrict__ gw) {
    const int KWd = (KW - 1) * dil + 1;
    const int Cig = Cin / groups, Cog = Cout / groups;
    const int64_t total = (int64_t)Cout * Cig * KH * KW;
    for (int64_t e = blockIdx.x * (int64_t)256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int kx = (int)(e % KW);
        int64_t t = e / KW;
        const int ky = (int)(t % KH);
        t /= KH;
        const int cig = (int)(t % Cig), co = (int)(t / Cig);
        const int ci = (co / Cog) * Cig + cig;
        gw[e] = gdense[(((size_t)(ky * dil) * KWd + kx * dil) * Cin + ci) * Cout + co];
    }
}

static int weight_expand_check(const char* who, const void* a, const void* b, int Cout, int Cin, int groups, int KH, int KW, int dil) {
    OTVAE_REQUIRE(a && b, "%s: NULL tensor", who);
    OTVAE_REQUIRE(Cout > 0 && Cin > 0 && groups > 0 && Cin % groups == 0 && Cout % groups == 0, "%s: %d -> %d channels in %d groups", who,
                  Cin, Cout, groups);
    OTVAE_REQUIRE(KH >= 1 && KW >= 1 && dil >= 1 && (KH - 1) * dil + 1 <= 32 && (KW - 1) * dil + 1 <= 32,
                  "%s: a %d x %d kernel with dilation %d exceeds the 32 x 32 taps of the convolution kernels", who, KH, KW, dil);
    return OTVAE_OK;
}

extern "C" int otvae_weight_expand_fwd(const float* w, int Cout, int Cin, int groups, int KH, int KW, int dilation, float* dense,
                                       void* stream) {
    if (int rc = weight_expand_check("otvae_weight_expand_fwd", w, dense, Cout, Cin, groups, KH, KW, dilation)) return rc;
    const int64_t total = (int64_t)((KH - 1) * dilation + 1) * ((KW - 1) * dilation + 1) * Cin * Cout;
    weight_expand_fwd_kernel<<<imin(cdiv(total, 256), 2048), 256, 0, (hipStream_t)stream>>>(w, Cout, Cin, groups, KH, KW, dilation, dense);
    OTVAE_CHECK_LAUNCH("otvae_weight_expand_fwd");
    return OTVAE_OK;
}

extern "C" int otvae_weight_expand_bwd(const float* gdense, int Cout, int Cin, int groups, int KH, int KW, int dilation, float* gw,
                                       void* stream) {
    if (int rc = weight_expand_check("otvae_weight_expand_bwd", gdense, gw, Cout, Cin, groups, KH, KW, dilation)) return rc;
    const int64_t total = (int64_t)Cout * (Cin / groups) * KH * KW;
    weight_expand_bwd_kernel<<<imin(cdiv(total, 256), 2048), 256, 0, (hipStream_t)stream>>>(gdense, Cout, Cin, groups, KH, KW, dilation, gw);
    OTVAE_CHECK_LAUNCH("otvae_weight_expand_bwd");
    return OTVAE_OK;
}
