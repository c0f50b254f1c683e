// ConvLayer weight (+bias) gradient as an IMAGE-TILE reduction on the fp32 MFMA (v_mfma_f32_16x16x4_f32), gfx950.
//
//   dW[k][n] = sum_pixels A[pixel][k] * gy[pixel][n],   k = (tap, input channel), A = act(x) seen through the tap
//
// conv.hip's conv_wgrad_kernel gathers A per (pixel, k) from global memory (bounds tests + address arithmetic per
// element, every input element re-read once per tap) and is bound by vector-instruction issue.  Here, as in
// conv_tile.hip, a workgroup stages the activated / up-sampled / zero-padded input of IPB whole images ONCE in LDS as a
// virtual grid V, and the output gradient of the same images as G[pixel][n]; then
//   A operand of lane (k-row = lane & 15, pixel = lane >> 4):  V[pixbase(pixel) + rowoff(k-row)]   (one add)
//   B operand of lane (n = lane & 15,     pixel = lane >> 4):  G[pixel][n]
// and D[k-row][n] accumulates in registers over ALL pixels of ALL images the workgroup walks through (grid-stride over
// image groups), so the workspace holds one partial per workgroup instead of one per 128-pixel chunk.  The 4 waves of
// a workgroup take every 4th group of 4 pixels (their accumulators are summed in wave order at the end); the bias gradient is the extra k-row whose A operand is the constant 1.
// Arithmetic: fp32 products and sums as conv_wgrad_kernel (different summation order over pixels); the partials are
// reduced in fixed order by wgrad_reduce(_batched)_kernel.  Reference: autograd backward of networks/cnn.py:183-192.
#include "common.h"
#include "conv_wtile.h"

__device__ __forceinline__ int wdiv(int k, float inv_d) { return (int)(((float)k + 0.5f) * inv_d); }  // exact, see conv.hip

template <int KTM, int NN>
__global__ __launch_bounds__(256) void conv_wtile_kernel(WTilePlan pl, const float* __restrict__ x,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         int relu, const float* __restrict__ gy, float* __restrict__ partial) {
    extern __shared__ __align__(16) float wsm[];
    float* V = wsm;                           // [IPB][Hv][Wv][CKp] then {0.0f, 1.0f}
    float* G = wsm + pl.vfloats + 4;          // [IPB*rowsPI (padded to 4)][CnP]
    int* pixtab = reinterpret_cast<int*>(G + pl.gfloats);  // [rowsPIp]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const int CK = pl.CK, CKp = pl.CKp, Cn = pl.Cn, CnP = pl.CnP;
    const int rowsPI = pl.rowsPI, rowsPIp = pl.rowsPIp;
    const int ZERO = pl.vfloats, ONE = pl.vfloats + 1;
    const int n0 = blockIdx.y * NN * 16;  // this block's first output channel

    // position table: float offset of each output position's origin inside one image's virtual grid
    for (int r = tid; r < rowsPIp; r += 256) {
        int v = -1;
        if (r < rowsPI) {
            const int ry = r / pl.rowsW, rx = r - ry * pl.rowsW;
            v = ((ry * pl.rstride) * pl.Wv + rx * pl.rstride) * CKp;
        }
        pixtab[r] = v;
    }
    if (tid == 0) {
        V[ZERO] = 0.f;
        V[ONE] = 1.f;
    }

    // this lane's k-rows: addr = mul * pixbase + off  (mul = 0 for the bias row / dead rows)
    int roff[KTM], rmul[KTM];
    {
        const float inv_ck = 1.0f / (float)CK;
#pragma unroll
        for (int t = 0; t < KTM; ++t) {
            const int k = t * 16 + r16;
            int off = ZERO, mul = 0;
            if (t < pl.nkt) {
                if (k < pl.K) {
                    const int tap = wdiv(k, inv_ck);
                    const int c = k - tap * CK;
                    if (pl.tapoff[tap] >= 0) {
                        off = pl.tapoff[tap] + c;
                        mul = 1;
                    }
                } else if (k == pl.K && pl.has_bias) {
                    off = ONE;
                }
            }
            roff[t] = off;
            rmul[t] = mul;
        }
    }

    f32x4 acc[KTM][NN];
#pragma unroll
    for (int t = 0; t < KTM; ++t)
#pragma unroll
        for (int j = 0; j < NN; ++j) acc[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int ngroups = (pl.N + pl.IPB - 1) / pl.IPB;
    const int img_floats = pl.Hv * pl.Wv * CKp;
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int img0 = grp * pl.IPB;
        const int nimg = min(pl.IPB, pl.N - img0);
        __syncthreads();  // the previous group's MFMA reads are done
        // ---- V: activated, up-sampled, zero-padded input
        if (pl.vec4) {
            const int ck4 = CK >> 2;
            const float inv_ck4 = 1.0f / (float)ck4, inv_wv = 1.0f / (float)pl.Wv, inv_hv = 1.0f / (float)pl.Hv;
            const int total = pl.IPB * pl.Hv * pl.Wv * ck4;
            for (int e0 = tid; e0 < total; e0 += 256 * 4) {
                float4 v[4];
                int dst[4], cc[4];
                bool ok[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = e0 + 256 * u;
                    const int pix = wdiv(e, inv_ck4);
                    const int c4 = e - pix * ck4;
                    const int t1 = wdiv(pix, inv_wv);
                    const int vx = pix - t1 * pl.Wv;
                    const int img = wdiv(t1, inv_hv);
                    const int vy = t1 - img * pl.Hv;
                    const int uy = vy - pl.voffy, ux = vx - pl.voffx;
                    ok[u] = e < total && img < nimg && (unsigned)uy < (unsigned)pl.limH && (unsigned)ux < (unsigned)pl.limW;
                    dst[u] = e < total ? pix * CKp + c4 * 4 : -1;
                    cc[u] = c4 * 4;
                    const size_t off = ok[u] ? ((size_t)((unsigned)(img0 + img) * pl.srcH + (uy >> pl.ush)) * pl.srcW + (ux >> pl.ush)) * CK + c4 * 4 : 0;
                    v[u] = *reinterpret_cast<const float4*>(x + off);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (dst[u] < 0) continue;
                    float4 a = v[u];
                    if (scale != nullptr) {
                        const float4 sc = *reinterpret_cast<const float4*>(scale + cc[u]);
                        const float4 sh = *reinterpret_cast<const float4*>(shift + cc[u]);
                        a.x = fmaf(a.x, sc.x, sh.x), a.y = fmaf(a.y, sc.y, sh.y);
                        a.z = fmaf(a.z, sc.z, sh.z), a.w = fmaf(a.w, sc.w, sh.w);
                    }
                    if (relu) a.x = fmaxf(a.x, 0.f), a.y = fmaxf(a.y, 0.f), a.z = fmaxf(a.z, 0.f), a.w = fmaxf(a.w, 0.f);
                    if (!ok[u]) a = make_float4(0.f, 0.f, 0.f, 0.f);  // padding is zero AFTER the activation
                    *reinterpret_cast<float4*>(V + dst[u]) = a;
                }
            }
        } else {  // channel count not a multiple of 4 (the 1- and 3-channel image side): one float per slot
            const float inv_ck = 1.0f / (float)CK, inv_wv = 1.0f / (float)pl.Wv, inv_hv = 1.0f / (float)pl.Hv;
            const int total = pl.IPB * pl.Hv * pl.Wv * CK;
            for (int e = tid; e < total; e += 256) {
                const int pix = wdiv(e, inv_ck);
                const int c = e - pix * CK;
                const int t1 = wdiv(pix, inv_wv);
                const int vx = pix - t1 * pl.Wv;
                const int img = wdiv(t1, inv_hv);
                const int vy = t1 - img * pl.Hv;
                const int uy = vy - pl.voffy, ux = vx - pl.voffx;
                float a = 0.f;
                if (img < nimg && (unsigned)uy < (unsigned)pl.limH && (unsigned)ux < (unsigned)pl.limW) {
                    a = x[((size_t)((unsigned)(img0 + img) * pl.srcH + (uy >> pl.ush)) * pl.srcW + (ux >> pl.ush)) * CK + c];
                    if (scale != nullptr) a = fmaf(a, scale[c], shift[c]);
                    if (relu) a = fmaxf(a, 0.f);
                }
                V[pix * CKp + c] = a;
            }
        }
        // ---- G: output gradient [IPB][rowsPIp][CnP], zero in the channel / position / image padding
        if (pl.gvec) {  // rows are contiguous in memory (and 16-byte aligned): a float4 copy
            const int total4 = (pl.IPB * rowsPI * Cn) >> 2, have4 = (nimg * rowsPI * Cn) >> 2;
            const float4* src = reinterpret_cast<const float4*>(gy + (size_t)img0 * rowsPI * Cn);
            float4* dstg = reinterpret_cast<float4*>(G);
            for (int e = tid; e < total4; e += 256) dstg[e] = e < have4 ? src[e] : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            const float inv_cnp = 1.0f / (float)CnP, inv_rp = 1.0f / (float)rowsPIp;
            const int total = pl.IPB * rowsPIp * CnP;
            for (int e = tid; e < total; e += 256) {
                const int p = wdiv(e, inv_cnp);
                const int n = e - p * CnP;
                const int img = wdiv(p, inv_rp);
                const int r = p - img * rowsPIp;
                float g = 0.f;
                if (img < nimg && r < rowsPI && n < Cn) g = gy[((size_t)(img0 + img) * rowsPI + r) * Cn + n];
                G[e] = g;
            }
        }
        __syncthreads();
        // ---- MFMA: wave w takes the pixel quads w, w + 4, ... of every image; QU quads per trip so that the LDS
        // reads of a trip are all in flight before its first MFMA
        const int nquads = rowsPIp >> 2;
        constexpr int QU = (KTM * NN <= 6) ? 4 : 2;
        for (int img = 0; img < pl.IPB; ++img) {
            const int vb = img * img_floats;
            const float* Gi = G + (size_t)img * rowsPIp * CnP + n0 + r16;
            for (int q0 = wave; q0 < nquads; q0 += 4 * QU) {
                int pt[QU], pp[QU];
#pragma unroll
                for (int u = 0; u < QU; ++u) {
                    const int q = q0 + 4 * u;
                    pp[u] = 4 * q + kq;
                    pt[u] = q < nquads ? pixtab[pp[u]] : -1;
                }
                float a[QU][KTM], b[QU][NN];
#pragma unroll
                for (int u = 0; u < QU; ++u) {
                    const bool live = pt[u] >= 0;
                    const int pb = vb + pt[u];
#pragma unroll
                    for (int j = 0; j < NN; ++j)
                        b[u][j] = (live && n0 + j * 16 + r16 < CnP) ? Gi[(size_t)pp[u] * CnP + j * 16] : 0.f;
#pragma unroll
                    for (int t = 0; t < KTM; ++t) a[u][t] = V[live ? rmul[t] * pb + roff[t] : ZERO];
                }
#pragma unroll
                for (int u = 0; u < QU; ++u)
#pragma unroll
                    for (int t = 0; t < KTM; ++t)
#pragma unroll
                        for (int j = 0; j < NN; ++j) acc[t][j] = mfma16(a[u][t], b[u][j], acc[t][j]);
            }
        }
    }

    // ---- the 4 waves' accumulators are summed in wave order through LDS (V/G are dead): one partial per block
    __syncthreads();
    float* red = wsm;  // [KTM*NN tiles][64 lanes][4]
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < KTM; ++t)
#pragma unroll
                for (int j = 0; j < NN; ++j) {
                    float4* slot = reinterpret_cast<float4*>(red) + (t * NN + j) * 64 + lane;
                    float4 v = make_float4(acc[t][j][0], acc[t][j][1], acc[t][j][2], acc[t][j][3]);
                    if (w > 0) {
                        const float4 o = *slot;
                        v.x += o.x, v.y += o.y, v.z += o.z, v.w += o.w;
                    }
                    if (w < 3) *slot = v;
                    else acc[t][j] = (f32x4){v.x, v.y, v.z, v.w};
                }
        }
        __syncthreads();
    }
    if (wave != 3) return;

    // ---- partial[block][k][n]
    float* out = partial + (size_t)blockIdx.x * pl.Kp * Cn;
#pragma unroll
    for (int t = 0; t < KTM; ++t) {
        if (t >= pl.nkt) continue;
#pragma unroll
        for (int j = 0; j < NN; ++j) {
            const int n = n0 + j * 16 + r16;
            if (n >= Cn) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = t * 16 + kq * 4 + r;
                if (k < pl.Kp) out[(size_t)k * Cn + n] = acc[t][j][r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ host
static bool wt_touches(int rows, int rstride, int d, int lim) {
    for (int r = 0; r < rows; ++r) {
        const int v = r * rstride + d;
        if (v >= 0 && v < lim) return true;
    }
    return false;
}

bool conv_wtile_plan(const Geom& g, int has_bias, WTilePlan& pl, int& nblocks, size_t& smem) {
    if (getenv("OTVAE_NO_WTILE")) return false;
    pl = {};
    const int T = g.KH * g.KW;
    if (T > WT_MAXT) return false;
    const int Hu = g.Hs * g.up, Wu = g.Ws * g.up;
    pl.N = g.N, pl.srcH = g.Hs, pl.srcW = g.Ws, pl.CK = g.Cs, pl.Cn = g.Cn;
    pl.ush = g.up - 1, pl.limH = Hu, pl.limW = Wu;
    pl.rowsH = g.Ho, pl.rowsW = g.Wo, pl.rowsPI = g.Ho * g.Wo, pl.rstride = g.stride;
    pl.rowsPIp = (pl.rowsPI + 3) & ~3;
    // deep layers (< 64 positions per image): conv_wgrad_kernel, output-stationary over many blocks; 32x32 maps (1024
    // positions): measured 2.5x slower than the direct / implicit-GEMM kernels (one image per block, staging-bound)
    // maps above 16x16: only where the alternative is the scalar-gather implicit GEMM (channel counts that are no multiple
    // of 4 and not served by the direct kernels, e.g. the 16 -> 3 layer of the RGB decoder at 32x32: 95 us there)
    const bool scalar_gemm_otherwise = ((g.Cs % 4 != 0) || (g.Cn % 4 != 0)) && !conv_small_wgrad_ok(g);
    if (pl.rowsPI < 64 || (pl.rowsPI > 256 && !scalar_gemm_otherwise && !getenv("OTVAE_WTILE_ALL"))) return false;
    if (pl.rowsPI > 1024) return false;
    pl.vec4 = (g.Cs % 4 == 0) ? 1 : 0;
    pl.CKp = pl.vec4 ? g.Cs + 4 : g.Cs + 1;
    pl.K = T * g.Cs;
    pl.has_bias = has_bias ? 1 : 0;
    pl.Kp = pl.K + pl.has_bias;
    pl.nkt = cdiv(pl.Kp, 16);
    const int nnt = cdiv(g.Cn, 16);
    if (pl.nkt > 18) return false;
    // column tiles per block: as many as the accumulator budget (20 tiles) allows, the rest over blockIdx.y
    int nn = imin(3, nnt);
    while (nn > 1 && pl.nkt * nn > 20) --nn;
    if (nn == 3 && pl.nkt > 6) nn = 2;
    if (nn == 2 && pl.nkt > 10) nn = 1;
    pl.nn = nn;
    pl.ny = cdiv(nnt, nn);
    pl.CnP = (g.Cn + 3) & ~3;  // columns beyond it read as zero
    pl.gvec = (pl.CnP == g.Cn && pl.rowsPIp == pl.rowsPI) ? 1 : 0;
    // taps that can touch the image for some output position; crop the virtual grid to their bounding box
    int tdy[WT_MAXT], tdx[WT_MAXT];
    int ylo = 1 << 30, yhi = -1, xlo = 1 << 30, xhi = -1;
    for (int kh = 0; kh < g.KH; ++kh)
        for (int kw = 0; kw < g.KW; ++kw) {
            const int t = kh * g.KW + kw;
            const bool ok = wt_touches(g.Ho, g.stride, kh - g.pad, Hu) && wt_touches(g.Wo, g.stride, kw - g.pad, Wu);
            tdy[t] = ok ? kh : -1;
            tdx[t] = kw;
            if (ok) {
                ylo = imin(ylo, kh), yhi = imax(yhi, kh + (g.Ho - 1) * g.stride);
                xlo = imin(xlo, kw), xhi = imax(xhi, kw + (g.Wo - 1) * g.stride);
            }
        }
    if (yhi < 0) return false;
    pl.Hv = yhi - ylo + 1, pl.Wv = xhi - xlo + 1;
    pl.voffy = g.pad - ylo, pl.voffx = g.pad - xlo;
    for (int t = 0; t < T; ++t) pl.tapoff[t] = tdy[t] < 0 ? -1 : ((tdy[t] - ylo) * pl.Wv + (tdx[t] - xlo)) * pl.CKp;
    // images per block iteration: as many as fit ~56 KiB (so that 2 blocks share a CU), at most 8
    const int per_img = pl.Hv * pl.Wv * pl.CKp + pl.rowsPIp * pl.CnP;
    int ipb = imin(8, imax(1, (14 * 1024) / per_img));
    while (ipb > 1 && (size_t)ipb * per_img * 4 > 56 * 1024) --ipb;
    ipb = imax(1, imin(ipb, g.N / 512));  // at least 512 image groups (2 blocks per CU) when the batch allows
    if ((size_t)per_img * 4 > 120 * 1024) return false;
    ipb = imin(ipb, g.N);
    pl.IPB = ipb;
    pl.vfloats = ipb * pl.Hv * pl.Wv * pl.CKp;
    pl.gfloats = ipb * pl.rowsPIp * pl.CnP;
    const int ngroups = cdiv(g.N, ipb);
    static const int max_blocks = [] {  // one weight-gradient partial per block: fewer blocks = less split-K traffic (A/B switch)
        const char* e = getenv("OTVAE_WTILE_BLOCKS");
        const int v = e ? atoi(e) : 0;
        return v >= 32 && v <= 2048 ? v : 512;
    }();
    nblocks = imin(ngroups, max_blocks);
    smem = ((size_t)pl.vfloats + 4 + pl.gfloats) * sizeof(float) + (size_t)pl.rowsPIp * sizeof(int);
    if (smem < 21 * 1024) smem = 21 * 1024;  // cross-wave reduction area: <= 20 accumulator tiles of 1 KiB
    return true;
}

int conv_wtile_nparts(int nblocks) { return nblocks; }

int conv_wtile(const WTilePlan& plan, int nblocks, size_t smem, hipStream_t st, const float* x, const float* scale,
               const float* shift, int relu, const float* gy, float* partial) {
    WTilePlan pl = plan;
    if (((uintptr_t)gy & 15) != 0) pl.gvec = 0;  // unaligned output gradient: element-wise staging
#define WT(K_, N_)                                                                                                        \
    do {                                                                                                                  \
        static bool attr_done = false;                                                                                    \
        if (!attr_done) {                                                                                                 \
            (void)hipFuncSetAttribute((const void*)conv_wtile_kernel<K_, N_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      128 * 1024);                                                                        \
            attr_done = true;                                                                                             \
        }                                                                                                                 \
        conv_wtile_kernel<K_, N_><<<dim3(nblocks, pl.ny), 256, smem, st>>>(pl, x, scale, shift, relu, gy, partial);                    \
    } while (0)
    const int kt = pl.nkt;
    if (pl.nn == 1) {
        if (kt <= 2) WT(2, 1);
        else if (kt <= 5) WT(5, 1);
        else if (kt <= 10) WT(10, 1);
        else WT(18, 1);
    } else if (pl.nn == 2) {
        if (kt <= 2) WT(2, 2);
        else if (kt <= 5) WT(5, 2);
        else WT(10, 2);
    } else {
        if (kt <= 2) WT(2, 3);
        else WT(6, 3);
    }
#undef WT
    return 0;
}
