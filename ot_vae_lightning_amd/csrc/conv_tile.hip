// ConvLayer forward / data-gradient as an IMAGE-TILE convolution on the fp32 MFMA (v_mfma_f32_16x16x4_f32), gfx950.
//
// The implicit-GEMM kernels of conv.hip gather every (row, k) element of the A operand from global memory with its own
// bounds test and address arithmetic; at this model's layer sizes (8..64 channels, 16x16 .. 1x1 maps, batch ~1000) that
// is ~1000 vector instructions per 64-row tile for 24..144 MFMAs, and the launches are bound by VALU issue, not by
// MFMA, LDS or HBM.  Here a workgroup owns IPB whole images instead:
//   * the layer input of those images (x after BatchNorm-apply + ReLU, nearest-upsampled; or the output gradient for
//     the data-gradient pass) is staged ONCE into LDS as a zero-padded "virtual grid" V[img][vy][vx][c] -- contiguous,
//     coalesced float4 loads, one activation per element, no per-tap re-reads;
//   * every (output position, tap) operand is then V[pixbase(position) + tapoff(tap) + c]: the bounds tests disappear
//     (padding is stored as zeros), the address is one add, and the inner loop is ds_read + MFMA;
//   * the weights of a group of taps sit in LDS next to it ([k][16*NT] slab, conflict-free reads).
// MFMA operands are swapped (A = weights, B = activations) so that D[out-channel][position]: a lane ends up with 4
// consecutive output channels of ONE position = one 16-byte store (and 16-byte loads of bias / residual / x).
//
// Same arithmetic as conv.hip (reference networks/cnn.py:183-192 and its autograd backward): products accumulate in
// fp32 over k = (tap, channel) in the same tap-major order; BatchNorm / BatchNorm-backward partial sums in fp64 with the
// fixed lane -> wave -> block order; workspace layouts ([2][cpad][P]) identical to the implicit-GEMM kernels.
#include <type_traits>

#include "common.h"
#include "conv_tile.h"

#define TILE_WMAX 4096  // floats of one weight chunk in LDS (16 KiB)

__device__ __forceinline__ int tdiv(int k, float inv_d) { return (int)(((float)k + 0.5f) * inv_d); }  // exact for k < 2^22
__device__ __forceinline__ int imax_d(int a, int b) { return a > b ? a : b; }

template <int MODE, int NT, int RBW>
__global__ __launch_bounds__(256) void conv_tile_kernel(TilePlan pl, const float* __restrict__ S,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        int relu, const float* __restrict__ Wg,
                                                        // FWD epilogue
                                                        const float* __restrict__ bias, const float* __restrict__ res,
                                                        float* __restrict__ y,
                                                        // DGRAD epilogue
                                                        const float* __restrict__ xin, const float* __restrict__ mean,
                                                        const float* __restrict__ invstd, float* __restrict__ gv,
                                                        double* __restrict__ partial, BnFold fold) {
    constexpr int BN = 16 * NT;
    extern __shared__ __align__(16) float tsm[];
    // forward: the BatchNorm affine of the input in LDS (CK <= 256 here) -- folded from the statistic slots by every block, or copied
    // from the arrays a finalize launch left (common.h: bn_tab_fill)
    __shared__ __align__(16) float bn_tab[2][256];
    const float* sc_p = nullptr;
    const float* sh_p = nullptr;
    if constexpr (MODE == 0) {
        if (scale != nullptr || fold.slots != nullptr) {
            bn_tab_fill(fold, scale, shift, pl.CK, bn_tab[0], bn_tab[1], (blockIdx.x | blockIdx.y | blockIdx.z) == 0);
            sc_p = bn_tab[0];
            sh_p = bn_tab[1];
        }
    }
    float* V = tsm;                                   // [IPB][Hv][Wv][CKp]
    float* Wl = tsm + pl.vfloats;                     // [tpc*CK][BN]
    double* red = reinterpret_cast<double*>(Wl + TILE_WMAX);  // [4 waves][2][BN]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const int cls = blockIdx.z;
    const TileTaps& tp = pl.taps[cls];
    const int CK = pl.CK, CKp = pl.CKp, NC = pl.NC;
    const int n0 = blockIdx.y * BN;
    const int img0 = blockIdx.x * pl.IPB;
    const int nimg = min(pl.IPB, pl.N - img0);

    // ---- stage the virtual grid of this block's images (zeros in the padding ring and past the last image)
    {
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
                const int pix = tdiv(e, inv_ck4);
                const int c4 = e - pix * ck4;
                const int t1 = tdiv(pix, inv_wv);
                const int vx = pix - t1 * pl.Wv;
                const int img = tdiv(t1, inv_hv);
                const int vy = t1 - img * pl.Hv;
                const int uy = vy - pl.voffy, ux = vx - pl.voffx;
                ok[u] = e < total && img < nimg && (unsigned)uy < (unsigned)pl.limH && (unsigned)ux < (unsigned)pl.limW;
                dst[u] = e < total ? pix * CKp + c4 * 4 : -1;
                cc[u] = c4 * 4;
                const size_t off = ok[u] ? ((size_t)((unsigned)(img0 + img) * pl.srcH + (uy >> pl.ush)) * pl.srcW + (ux >> pl.ush)) * CK + c4 * 4 : 0;
                v[u] = *reinterpret_cast<const float4*>(S + off);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (dst[u] < 0) continue;
                float4 a = v[u];
                if (MODE == 0) {
                    if (sc_p != nullptr) {
                        const float4 sc = *reinterpret_cast<const float4*>(sc_p + cc[u]);
                        const float4 sh = *reinterpret_cast<const float4*>(sh_p + cc[u]);
                        a.x = fmaf(a.x, sc.x, sh.x);
                        a.y = fmaf(a.y, sc.y, sh.y);
                        a.z = fmaf(a.z, sc.z, sh.z);
                        a.w = fmaf(a.w, sc.w, sh.w);
                    }
                    if (relu) {
                        a.x = fmaxf(a.x, 0.f);
                        a.y = fmaxf(a.y, 0.f);
                        a.z = fmaxf(a.z, 0.f);
                        a.w = fmaxf(a.w, 0.f);
                    }
                }
                if (!ok[u]) a = make_float4(0.f, 0.f, 0.f, 0.f);  // padding is zero AFTER the activation
                *reinterpret_cast<float4*>(V + dst[u]) = a;
            }
        }
    }

    // ---- this wave's row blocks: lane (r16) <-> one output position of each
    const int rows_blk = pl.IPB * pl.rowsPI;
    // integer divisions by launch constants go through one float multiply (exact: tdiv), not the ~25-instruction
    // udiv expansion; they sit in per-row-block code that would otherwise out-weigh the MFMAs of a small layer
    const float inv_rpi = 1.0f / (float)pl.rowsPI, inv_rw = 1.0f / (float)pl.rowsW;
    const float inv_ws = 1.0f / (float)imax_d(pl.rowsW >> 1, 1);
    int pixbase[RBW];  // float offset of the position's virtual-grid origin
    int rloc[RBW];     // row within the block, or -1
#pragma unroll
    for (int i = 0; i < RBW; ++i) {
        const int rb = wave + 4 * i;
        const int rl = rb * 16 + r16;
        int pb = 0;
        int ok = -1;
        if (rl < rows_blk) {
            const int img = tdiv(rl, inv_rpi);
            const int r = rl - img * pl.rowsPI;
            int ry, rx;
            if (pl.childmode) {
                const int parent = r >> 2, child = r & 3;
                const int ws = pl.rowsW >> 1;
                const int sy = tdiv(parent, inv_ws), sx = parent - sy * ws;
                ry = 2 * sy + (child >> 1);
                rx = 2 * sx + (child & 1);
            } else {
                ry = tdiv(r, inv_rw);
                rx = r - ry * pl.rowsW;
            }
            pb = ((img * pl.Hv + ry * pl.rstride) * pl.Wv + rx * pl.rstride) * CKp;
            if (img < nimg) ok = rl;
        }
        pixbase[i] = pb + kq;
        rloc[i] = ok;
    }

    f32x4 acc[RBW][NT];
#pragma unroll
    for (int i = 0; i < RBW; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- weight chunks: groups of `tpc` taps x all CK channels, [k][BN] in LDS; next chunk prefetched into registers
    const int tpc = pl.tpc;
    const int nchunk = (tp.n + tpc - 1) / tpc;
    constexpr int NWV = TILE_WMAX / 4 / 256;  // float4 slots per thread per chunk
    float4 wreg[NWV];
    const float inv_ck = 1.0f / (float)CK;
    auto wload = [&](int ch) {
        const int t0 = ch * tpc;
        const int rows = min(tpc, tp.n - t0) * CK;
#pragma unroll
        for (int u = 0; u < NWV; ++u) {
            const int e = tid + 256 * u;
            const int row = e / (BN / 4), c4 = e - row * (BN / 4);
            const bool ok = row < rows && n0 + c4 * 4 < NC;
            const int tl = ok ? tdiv(row, inv_ck) : 0;
            const int c = row - tl * CK;
            const size_t off = ok ? ((size_t)tp.wrow[t0 + tl] * CK + c) * NC + n0 + c4 * 4 : 0;
            const float4 v = *reinterpret_cast<const float4*>(Wg + off);
            wreg[u] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto wstore = [&]() {
#pragma unroll
        for (int u = 0; u < NWV; ++u) {
            const int e = tid + 256 * u;
            if (e < tpc * CK * (BN / 4)) *reinterpret_cast<float4*>(Wl + e * 4) = wreg[u];
        }
    };
    wload(0);
    wstore();
    __syncthreads();
    for (int ch = 0; ch < nchunk; ++ch) {
        if (ch + 1 < nchunk) wload(ch + 1);
        const int t0 = ch * tpc;
        const int tcount = min(tpc, tp.n - t0);
        for (int tl = 0; tl < tcount; ++tl) {
            const int toff = tp.off[t0 + tl];
            const float* wr = Wl + (size_t)(tl * CK + kq) * BN + r16;
            // KS k-steps (4 channels each) per trip: all LDS reads of the trip are issued before its MFMAs, so one
            // lgkmcnt wait covers KS*(NT+RBW) reads instead of one wait per MFMA group (the loop is latency-bound at
            // one wave per SIMD otherwise).  Accumulation order over k is unchanged.
            auto ksteps = [&](int c0, auto ks_tag) {
                constexpr int KS = decltype(ks_tag)::value;
                float w[KS][NT], a[KS][RBW];
#pragma unroll
                for (int s = 0; s < KS; ++s) {
#pragma unroll
                    for (int j = 0; j < NT; ++j) w[s][j] = wr[(c0 + 4 * s) * BN + j * 16];
#pragma unroll
                    for (int i = 0; i < RBW; ++i) a[s][i] = V[pixbase[i] + toff + c0 + 4 * s];
                }
#pragma unroll
                for (int s = 0; s < KS; ++s)
#pragma unroll
                    for (int i = 0; i < RBW; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j) acc[i][j] = mfma16(w[s][j], a[s][i], acc[i][j]);
            };
            int c0 = 0;
            for (; c0 + 16 <= CK; c0 += 16) ksteps(c0, std::integral_constant<int, 4>{});
            for (; c0 + 8 <= CK; c0 += 8) ksteps(c0, std::integral_constant<int, 2>{});
            for (; c0 < CK; c0 += 4) ksteps(c0, std::integral_constant<int, 1>{});
        }
        if (ch + 1 < nchunk) {
            __syncthreads();
            wstore();
            __syncthreads();
        }
    }

    // ---- epilogue: lane = position r16 of the row block, channels n0 + 16 j + 4 kq .. + 3
    double s1[NT][4], s2[NT][4];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) s1[j][r] = s2[j][r] = 0.0;
    const bool want_sums = (MODE == 0) ? (partial != nullptr) : (mean != nullptr);

#pragma unroll
    for (int i = 0; i < RBW; ++i) {
        const int rl = rloc[i];
        // position decode (again; cheap, once per row block)
        int img = 0, r = 0;
        if (rl >= 0) {
            img = tdiv(rl, inv_rpi);
            r = rl - img * pl.rowsPI;
        }
        const unsigned gimg = img0 + img;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int ch = n0 + j * 16 + kq * 4;
            const bool chok = ch < NC;
            float4 val = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            if (MODE == 0) {
                if (rl >= 0 && chok) {
                    const size_t o = ((size_t)gimg * pl.rowsPI + r) * NC + ch;
                    if (bias) {
                        const float4 b = *reinterpret_cast<const float4*>(bias + ch);
                        val.x += b.x, val.y += b.y, val.z += b.z, val.w += b.w;
                    }
                    if (res) {
                        const float4 q = *reinterpret_cast<const float4*>(res + o);
                        val.x += q.x, val.y += q.y, val.z += q.z, val.w += q.w;
                    }
                    *reinterpret_cast<float4*>(y + o) = val;
                    if (partial) {
                        s1[j][0] += (double)val.x, s2[j][0] += (double)val.x * (double)val.x;
                        s1[j][1] += (double)val.y, s2[j][1] += (double)val.y * (double)val.y;
                        s1[j][2] += (double)val.z, s2[j][2] += (double)val.z * (double)val.z;
                        s1[j][3] += (double)val.w, s2[j][3] += (double)val.w * (double)val.w;
                    }
                }
            } else {
                bool writer = rl >= 0 && chok;
                size_t o = 0;
                if (pl.childmode) {
                    // the 4 children of one source pixel are 4 consecutive lanes: fixed-order quad sum
                    float4 t;
                    t.x = val.x + __shfl_xor(val.x, 1, 64), t.y = val.y + __shfl_xor(val.y, 1, 64);
                    t.z = val.z + __shfl_xor(val.z, 1, 64), t.w = val.w + __shfl_xor(val.w, 1, 64);
                    val.x = t.x + __shfl_xor(t.x, 2, 64), val.y = t.y + __shfl_xor(t.y, 2, 64);
                    val.z = t.z + __shfl_xor(t.z, 2, 64), val.w = t.w + __shfl_xor(t.w, 2, 64);
                    writer = writer && ((r16 & 3) == 0);
                    o = ((size_t)gimg * (pl.rowsPI >> 2) + (r >> 2)) * NC + ch;
                } else if (pl.s2) {
                    const int ry = tdiv(r, inv_rw), rx = r - ry * pl.rowsW;
                    const int py = cls >> 1, px = cls & 1;
                    o = (((size_t)gimg * (2 * pl.rowsH) + 2 * ry + py) * (2 * pl.rowsW) + 2 * rx + px) * NC + ch;
                } else {
                    o = ((size_t)gimg * pl.rowsPI + r) * NC + ch;
                }
                if (writer) {
                    float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (relu || mean) xv = *reinterpret_cast<const float4*>(xin + o);
                    if (relu) {
                        float4 a = xv;
                        if (scale) {
                            const float4 sc = *reinterpret_cast<const float4*>(scale + ch);
                            const float4 sh = *reinterpret_cast<const float4*>(shift + ch);
                            a.x = fmaf(a.x, sc.x, sh.x), a.y = fmaf(a.y, sc.y, sh.y);
                            a.z = fmaf(a.z, sc.z, sh.z), a.w = fmaf(a.w, sc.w, sh.w);
                        }
                        val.x = a.x > 0.f ? val.x : 0.f;
                        val.y = a.y > 0.f ? val.y : 0.f;
                        val.z = a.z > 0.f ? val.z : 0.f;
                        val.w = a.w > 0.f ? val.w : 0.f;
                    }
                    *reinterpret_cast<float4*>(gv + o) = val;
                    if (mean) {
                        const float4 mu = *reinterpret_cast<const float4*>(mean + ch);
                        const float4 is = *reinterpret_cast<const float4*>(invstd + ch);
                        s1[j][0] += (double)val.x, s2[j][0] += (double)val.x * (double)((xv.x - mu.x) * is.x);
                        s1[j][1] += (double)val.y, s2[j][1] += (double)val.y * (double)((xv.y - mu.y) * is.y);
                        s1[j][2] += (double)val.z, s2[j][2] += (double)val.z * (double)((xv.z - mu.z) * is.z);
                        s1[j][3] += (double)val.w, s2[j][3] += (double)val.w * (double)((xv.w - mu.w) * is.w);
                    }
                }
            }
        }
    }

    if (want_sums) {
        // fixed order: the 16 position lanes of each kq group (xor tree), then the 4 waves through LDS
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double a = s1[j][r], b = s2[j][r];
                a += __shfl_xor(a, 1, 64), b += __shfl_xor(b, 1, 64);
                a += __shfl_xor(a, 2, 64), b += __shfl_xor(b, 2, 64);
                a += __shfl_xor(a, 4, 64), b += __shfl_xor(b, 4, 64);
                a += __shfl_xor(a, 8, 64), b += __shfl_xor(b, 8, 64);
                if (r16 == 0) {
                    red[(wave * 2 + 0) * BN + j * 16 + kq * 4 + r] = a;
                    red[(wave * 2 + 1) * BN + j * 16 + kq * 4 + r] = b;
                }
            }
        __syncthreads();
        if (tid < 2 * BN) {
            const int which = tid / BN, cc = tid % BN;
            const double t = (red[(0 * 2 + which) * BN + cc] + red[(1 * 2 + which) * BN + cc]) +
                             (red[(2 * 2 + which) * BN + cc] + red[(3 * 2 + which) * BN + cc]);
            const unsigned p = blockIdx.z * gridDim.x + blockIdx.x;
            const unsigned Ptot = gridDim.x * gridDim.z;
            if (n0 + cc < pl.cpad) bn_stat_out(partial, which, pl.cpad, n0 + cc, Ptot, p, t);  // [2][cpad][P], or the statistic slots
        }
    }
}

// ------------------------------------------------------------------------------------------------ host: planning
static bool tap_touches(int rows, int rstride, int d, int lim) {
    // exists r in [0, rows): 0 <= r*rstride + d < lim
    for (int r = 0; r < rows; ++r) {
        const int v = r * rstride + d;
        if (v >= 0 && v < lim) return true;
    }
    return false;
}

bool conv_tile_plan(const Geom& g, int mode, TilePlan& pl, dim3& grid, size_t& smem) {
    if ((g.Cs % 4) || (g.Cn % 4)) return false;
    if (getenv("OTVAE_NO_TILE")) return false;
    pl = {};
    int tdy[4][TILE_MAXT], tdx[4][TILE_MAXT];  // virtual-grid offsets of the kept taps (before cropping)
    const int Hu = g.Hs * g.up, Wu = g.Ws * g.up;
    pl.N = g.N;
    int nz = 1;
    if (mode == 0) {
        pl.srcH = g.Hs, pl.srcW = g.Ws, pl.CK = g.Cs, pl.NC = g.Cn;
        pl.Hv = Hu + 2 * g.pad, pl.Wv = Wu + 2 * g.pad;
        pl.voffy = pl.voffx = g.pad;
        pl.ush = g.up - 1;
        pl.limH = Hu, pl.limW = Wu;
        pl.rowsH = g.Ho, pl.rowsW = g.Wo, pl.rstride = g.stride;
        if ((g.Ho - 1) * g.stride + g.KH > pl.Hv || (g.Wo - 1) * g.stride + g.KW > pl.Wv) return false;
        TileTaps& tp = pl.taps[0];
        for (int kh = 0; kh < g.KH; ++kh)
            for (int kw = 0; kw < g.KW; ++kw) {
                if (!tap_touches(g.Ho, g.stride, kh - g.pad, Hu) || !tap_touches(g.Wo, g.stride, kw - g.pad, Wu)) continue;
                if (tp.n == TILE_MAXT) return false;
                tdy[0][tp.n] = kh, tdx[0][tp.n] = kw;
                tp.wrow[tp.n] = kh * g.KW + kw;
                ++tp.n;
            }
    } else {
        pl.srcH = g.Ho, pl.srcW = g.Wo, pl.CK = g.Cn, pl.NC = g.Cs;
        pl.ush = 0;
        pl.limH = g.Ho, pl.limW = g.Wo;
        pl.rstride = 1;
        if (g.stride == 1) {
            const int vy = g.KH - 1 - g.pad, vx = g.KW - 1 - g.pad;
            if (vy < 0 || vx < 0) return false;
            pl.voffy = vy, pl.voffx = vx;
            pl.Hv = Hu + g.KH - 1, pl.Wv = Wu + g.KW - 1;
            pl.rowsH = Hu, pl.rowsW = Wu;
            pl.childmode = g.up == 2 ? 1 : 0;
            TileTaps& tp = pl.taps[0];
            for (int kh = 0; kh < g.KH; ++kh)
                for (int kw = 0; kw < g.KW; ++kw) {
                    if (!tap_touches(Hu, 1, g.pad - kh, g.Ho) || !tap_touches(Wu, 1, g.pad - kw, g.Wo)) continue;
                    if (tp.n == TILE_MAXT) return false;
                    tdy[0][tp.n] = g.KH - 1 - kh, tdx[0][tp.n] = g.KW - 1 - kw;
                    tp.wrow[tp.n] = kh * g.KW + kw;
                    ++tp.n;
                }
        } else {  // stride 2 (up == 1): one launch slice (blockIdx.z) per input-parity class
            if (g.up != 1 || (g.Hs & 1) || (g.Ws & 1)) return false;
            nz = 4;
            pl.s2 = 1;
            pl.rowsH = g.Hs >> 1, pl.rowsW = g.Ws >> 1;
            // source offset of tap kh for parity py: d = (py + pad - kh) / 2 when that is even
            int dmin = 0, dmax = 0;
            for (int py = 0; py < 2; ++py)
                for (int kh = 0; kh < imax(g.KH, g.KW); ++kh) {
                    const int t = py + g.pad - kh;
                    if (t & 1) continue;
                    const int d = t / 2;  // exact (t even), also for negative t
                    dmin = imin(dmin, d);
                    dmax = imax(dmax, d);
                }
            pl.voffy = pl.voffx = -dmin;
            pl.Hv = pl.rowsH + dmax - dmin, pl.Wv = pl.rowsW + dmax - dmin;
            for (int cls = 0; cls < 4; ++cls) {
                const int py = cls >> 1, px = cls & 1;
                TileTaps& tp = pl.taps[cls];
                for (int kh = 0; kh < g.KH; ++kh) {
                    const int ty = py + g.pad - kh;
                    if (ty & 1) continue;
                    for (int kw = 0; kw < g.KW; ++kw) {
                        const int tx = px + g.pad - kw;
                        if (tx & 1) continue;
                        const int dy = ty / 2, dx = tx / 2;
                        if (!tap_touches(pl.rowsH, 1, dy, g.Ho) || !tap_touches(pl.rowsW, 1, dx, g.Wo)) continue;
                        if (tp.n == TILE_MAXT) return false;
                        tdy[cls][tp.n] = dy + pl.voffy, tdx[cls][tp.n] = dx + pl.voffx;
                        tp.wrow[tp.n] = kh * g.KW + kw;
                        ++tp.n;
                    }
                }
            }
        }
    }
    pl.rowsPI = pl.rowsH * pl.rowsW;
    pl.CKp = pl.CK + 4;
    {
        // crop the virtual grid to the bounding box the kept taps can reach (a 3x3 layer on a 1x1 map keeps only its
        // centre tap: the grid shrinks from 3x3 to 1x1)
        int ylo = 1 << 30, yhi = -1, xlo = 1 << 30, xhi = -1;
        for (int c = 0; c < 4; ++c)
            for (int t = 0; t < pl.taps[c].n; ++t) {
                ylo = imin(ylo, tdy[c][t]), yhi = imax(yhi, tdy[c][t] + (pl.rowsH - 1) * pl.rstride);
                xlo = imin(xlo, tdx[c][t]), xhi = imax(xhi, tdx[c][t] + (pl.rowsW - 1) * pl.rstride);
            }
        if (yhi < 0) return false;  // no tap at all
        pl.Hv = yhi - ylo + 1, pl.Wv = xhi - xlo + 1;
        pl.voffy -= ylo, pl.voffx -= xlo;
        for (int c = 0; c < 4; ++c)
            for (int t = 0; t < pl.taps[c].n; ++t) pl.taps[c].off[t] = ((tdy[c][t] - ylo) * pl.Wv + (tdx[c][t] - xlo)) * pl.CKp;
    }
    if (pl.rowsPI > 256) return false;
    if (pl.rowsPI < 64 && !getenv("OTVAE_TILE_ALL")) return false;  // deep layers: the K-pipelined implicit GEMM is faster
                                                                      // (measured again at the final state: 16 rows 3.17, 4 rows 3.34 vs 3.10 ms)
    if (pl.CK * 16 > TILE_WMAX) return false;  // one tap of weights must fit the chunk (CK <= 256)
    // images per block: 64 .. 256 rows, more rows per block only when that still leaves >= 512 blocks (measured: 384..512
    // is the flat optimum for the whole step, 1024 and 2048 are 1 % slower)
    const int per_img_floats = pl.Hv * pl.Wv * pl.CKp;
    const long total_rows = (long)g.N * pl.rowsPI;
    int target = 64;
    if (total_rows / 128 >= 512) target = 128;
    if (total_rows / 256 >= 512) target = 256;
    int ipb = imax(1, target / pl.rowsPI);
    const int lds_floats = (96 * 1024) / 4 - TILE_WMAX - 4 * 2 * 64 * 2;
    while (ipb > 1 && (long)ipb * per_img_floats > lds_floats) --ipb;
    if ((long)ipb * per_img_floats > lds_floats) return false;
    ipb = imin(ipb, g.N);
    pl.IPB = ipb;
    pl.vfloats = ipb * per_img_floats;
    const int rows_blk = ipb * pl.rowsPI;
    pl.rbw = cdiv(rows_blk, 64);
    if (pl.rbw == 3) pl.rbw = 4;
    if (pl.rbw > 4) return false;
    // column tiles per wave: as conv.hip (>= 512 blocks when possible)
    const int nnt = cdiv(pl.NC, 16);
    int nt = nnt >= 4 ? 4 : nnt;
    const long rb = (long)cdiv(g.N, ipb) * nz;
    while (nt > 1 && rb * cdiv(nnt, nt) < 512) --nt;
    if (pl.rbw * nt > 8) nt = imax(1, 8 / pl.rbw);  // accumulator registers
    while (nt > 1 && pl.CK * 16 * nt > TILE_WMAX) --nt;  // one tap of weights must fit the chunk
    pl.nt = nt;
    const int ny = cdiv(nnt, nt);
    pl.cpad = ny * 16 * nt;
    pl.tpc = imax(1, TILE_WMAX / (pl.CK * 16 * nt));
    grid = dim3(cdiv(g.N, ipb), ny, nz);
    smem = ((size_t)pl.vfloats + TILE_WMAX) * sizeof(float) + sizeof(double) * 4 * 2 * 16 * nt;
    return true;
}

template <int MODE>
static int tile_launch(const TilePlan& pl, dim3 grid, size_t smem, hipStream_t st, const float* S, const float* scale,
                       const float* shift, int relu, const float* Wg, const float* bias, const float* res, float* y,
                       const float* xin, const float* mean, const float* invstd, float* gv, double* partial,
                       const BnFold& fold = BnFold{}) {
#define TL(N_, R_)                                                                                                    \
    do {                                                                                                              \
        static bool attr_done = false;                                                                                \
        if (!attr_done) {                                                                                             \
            (void)hipFuncSetAttribute((const void*)conv_tile_kernel<MODE, N_, R_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                96 * 1024);                                                                           \
            attr_done = true;                                                                                         \
        }                                                                                                             \
        conv_tile_kernel<MODE, N_, R_><<<grid, 256, smem, st>>>(pl, S, scale, shift, relu, Wg, bias, res, y, xin, mean, \
                                                                invstd, gv, partial, fold);                           \
    } while (0)
#define TLR(N_)                                  \
    switch (pl.rbw) {                            \
        case 1: TL(N_, 1); break;                \
        case 2: TL(N_, 2); break;                \
        default: TL(N_, 4); break;               \
    }
    switch (pl.nt) {
        case 1: TLR(1); break;
        case 2: TLR(2); break;
        case 3:
            if (pl.rbw == 4) return -1;
            switch (pl.rbw) {
                case 1: TL(3, 1); break;
                default: TL(3, 2); break;
            }
            break;
        default:
            if (pl.rbw == 4) return -1;
            switch (pl.rbw) {
                case 1: TL(4, 1); break;
                default: TL(4, 2); break;
            }
            break;
    }
#undef TLR
#undef TL
    return 0;
}

int conv_tile_fwd(const TilePlan& pl, dim3 grid, size_t smem, hipStream_t st, const float* x, const float* scale,
                  const float* shift, int relu, const float* wT, const float* bias, const float* res, float* y,
                  double* partial, const BnFold& fold) {
    return tile_launch<0>(pl, grid, smem, st, x, scale, shift, relu, wT, bias, res, y, nullptr, nullptr, nullptr, nullptr,
                          partial, fold);
}

int conv_tile_dgrad(const TilePlan& pl, dim3 grid, size_t smem, hipStream_t st, const float* gy, const float* wD,
                    const float* x, const float* scale, const float* shift, int relu, const float* mean,
                    const float* invstd, float* gv, double* partial) {
    return tile_launch<1>(pl, grid, smem, st, gy, scale, shift, relu, wD, nullptr, nullptr, nullptr, x, mean, invstd, gv,
                          partial);
}
