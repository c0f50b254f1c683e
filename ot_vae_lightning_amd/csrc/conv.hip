// ConvLayer forward / data-gradient / weight-gradient as implicit GEMMs on the fp32 MFMA (v_mfma_f32_16x16x4_f32).
//
// Replaces the ATen sequence  batch_norm -> relu -> upsample_nearest2d -> conv2d (+bias, +residual add)  issued by
// ConvLayer.forward (reference networks/cnn.py:183-192) and its autograd backward, for the layer geometries the
// CNN builder produces (3x3 s1 p1, 4x4 s2 p1, 1x1, optional nearest x2 up-sampling before the conv).
//
// GEMM view (M = pixels, K = taps*channels, N = output channels), all operands gathered straight from NHWC global
// memory, normalisation/activation/up-sampling/zero-padding applied on the operand load:
//   fwd   Y[m][n]  = sum_k act(X)[gather(m,k)] * W[k][n]                  rows = output pixels
//   dgrad dU[q][c] = sum_(t,n) GY[gather(q,t)][n] * Wd[(t,n)][c]          rows = input positions (grouped so that
//                    the 4 nearest-upsample children of a pixel, resp. one stride-parity class, share a wave)
//   wgrad dW[k][n] = sum_m act(X)[gather(m,k)] * GY[m][n]                 rows = k, reduction over pixels
// Tile per wave: 16 rows x (16*NT) columns, K advanced 4 at a time.  A wave's MFMA operands are one fp32 per lane:
// A[row = lane&15][k = lane>>4], B[k = lane>>4][col = lane&15]; D[r] = D[(lane>>4)*4 + r][lane&15].
#include "common.h"

struct Geom {
    int N, Hs, Ws, Cs, up, Ho, Wo, Cn, KH, KW, stride, pad;
};

static inline Geom to_geom(const otvae_conv_geom* g) {
    Geom r = {g->N, g->Hs, g->Ws, g->Cs, g->up, g->Ho, g->Wo, g->Cn, g->KH, g->KW, g->stride, g->pad};
    return r;
}

static int check_geom(const otvae_conv_geom* g, const char* who) {
    OTVAE_REQUIRE(g != nullptr, "%s: geom is NULL", who);
    OTVAE_REQUIRE(g->N > 0 && g->Hs > 0 && g->Ws > 0 && g->Cs > 0 && g->Cn > 0, "%s: non-positive dims", who);
    OTVAE_REQUIRE(g->up == 1 || g->up == 2, "%s: up must be 1 or 2 (got %d)", who, g->up);
    OTVAE_REQUIRE(g->stride == 1 || g->stride == 2, "%s: stride must be 1 or 2 (got %d)", who, g->stride);
    OTVAE_REQUIRE(!(g->up == 2 && g->stride != 1), "%s: up-sampling with stride != 1 unsupported", who);
    OTVAE_REQUIRE(g->KH >= 1 && g->KH <= 7 && g->KW >= 1 && g->KW <= 7, "%s: kernel size out of range", who);
    int Hu = g->Hs * g->up, Wu = g->Ws * g->up;
    int ho = (Hu + 2 * g->pad - g->KH) / g->stride + 1, wo = (Wu + 2 * g->pad - g->KW) / g->stride + 1;
    OTVAE_REQUIRE(ho == g->Ho && wo == g->Wo, "%s: output size mismatch: geom says %dx%d, conv gives %dx%d", who,
                  g->Ho, g->Wo, ho, wo);
    OTVAE_REQUIRE((int64_t)g->N * Hu * Wu * imax(g->Cs, g->Cn) < (int64_t)1 << 31, "%s: tensor too large for 32-bit indexing",
                  who);
    if (g->stride == 2) OTVAE_REQUIRE(g->Hs % 2 == 0 && g->Ws % 2 == 0, "%s: stride 2 needs even input size", who);
    return OTVAE_OK;
}

__device__ __forceinline__ float act_load(const float* __restrict__ x, unsigned idx, int c, const float* __restrict__ scale,
                                          const float* __restrict__ shift, int relu) {
    float a = x[idx];
    if (scale) a = fmaf(a, scale[c], shift[c]);
    if (relu) a = fmaxf(a, 0.f);
    return a;
}

// ------------------------------------------------------------------------------------------------ forward
template <int NT, bool SMALLC>
__global__ __launch_bounds__(256) void conv_fwd_kernel(Geom g, const float* __restrict__ x, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, int relu,
                                                       const float* __restrict__ wT, const float* __restrict__ bias,
                                                       const float* __restrict__ res, float* __restrict__ y) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const unsigned M = (unsigned)g.N * g.Ho * g.Wo;
    const unsigned ntiles = (M + 63) / 64;
    const int n0 = blockIdx.y * (16 * NT);
    const int Hu = g.Hs * g.up, Wu = g.Ws * g.up;
    const int ush = g.up - 1;  // up in {1,2} -> shift 0/1
    const int T = g.KH * g.KW;

    for (unsigned tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const unsigned m = tile * 64 + wave * 16 + r16;
        const bool mv = m < M;
        int ox = 0, oy = 0, n = 0;
        if (mv) {
            ox = m % g.Wo;
            unsigned t = m / g.Wo;
            oy = t % g.Ho;
            n = t / g.Ho;
        }
        f32x4 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};

        if constexpr (!SMALLC) {
            for (int kh = 0; kh < g.KH; ++kh) {
                const int iy = oy * g.stride + kh - g.pad;
                for (int kw = 0; kw < g.KW; ++kw) {
                    const int ix = ox * g.stride + kw - g.pad;
                    const bool inb = mv && iy >= 0 && iy < Hu && ix >= 0 && ix < Wu;
                    const unsigned base = (((unsigned)n * g.Hs + (iy >> ush)) * g.Ws + (ix >> ush)) * g.Cs;
                    const float* wrow = wT + (size_t)(kh * g.KW + kw) * g.Cs * g.Cn;
                    for (int c0 = 0; c0 < g.Cs; c0 += 4) {
                        const int c = c0 + kq;
                        float a = 0.f;
                        if (inb) a = act_load(x, base + c, c, scale, shift, relu);
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
                            const int col = n0 + j * 16 + r16;
                            const float b = col < g.Cn ? wrow[(size_t)c * g.Cn + col] : 0.f;
                            acc[j] = mfma16(a, b, acc[j]);
                        }
                    }
                }
            }
        } else {
            const int K = T * g.Cs;
            for (int k0 = 0; k0 < K; k0 += 4) {
                const int k = k0 + kq;
                const bool kv = k < K;
                const int t = kv ? k / g.Cs : 0;
                const int c = kv ? k - t * g.Cs : 0;
                const int kh = t / g.KW, kw = t - kh * g.KW;
                const int iy = oy * g.stride + kh - g.pad, ix = ox * g.stride + kw - g.pad;
                const bool inb = kv && mv && iy >= 0 && iy < Hu && ix >= 0 && ix < Wu;
                float a = 0.f;
                if (inb) {
                    const unsigned idx = (((unsigned)n * g.Hs + (iy >> ush)) * g.Ws + (ix >> ush)) * g.Cs + c;
                    a = act_load(x, idx, c, scale, shift, relu);
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int col = n0 + j * 16 + r16;
                    const float b = (kv && col < g.Cn) ? wT[(size_t)k * g.Cn + col] : 0.f;
                    acc[j] = mfma16(a, b, acc[j]);
                }
            }
        }
        // epilogue: rows kq*4+r of this wave's 16-pixel tile, column r16 of each n-tile
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = n0 + j * 16 + r16;
            if (col < g.Cn) {
                const float bv = bias ? bias[col] : 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned pm = tile * 64 + wave * 16 + kq * 4 + r;
                    if (pm < M) {
                        const size_t o = (size_t)pm * g.Cn + col;
                        float v = acc[j][r] + bv;
                        if (res) v += res[o];
                        y[o] = v;
                    }
                }
            }
        }
    }
}

template <bool SMALLC>
static void launch_fwd(int NT, dim3 grid, hipStream_t st, Geom g, const float* x, const float* scale, const float* shift,
                       int relu, const float* wT, const float* bias, const float* res, float* y) {
    switch (NT) {
        case 1: conv_fwd_kernel<1, SMALLC><<<grid, 256, 0, st>>>(g, x, scale, shift, relu, wT, bias, res, y); break;
        case 2: conv_fwd_kernel<2, SMALLC><<<grid, 256, 0, st>>>(g, x, scale, shift, relu, wT, bias, res, y); break;
        case 3: conv_fwd_kernel<3, SMALLC><<<grid, 256, 0, st>>>(g, x, scale, shift, relu, wT, bias, res, y); break;
        default: conv_fwd_kernel<4, SMALLC><<<grid, 256, 0, st>>>(g, x, scale, shift, relu, wT, bias, res, y); break;
    }
}

static inline int pick_nt(int ncols) {
    int nnt = cdiv(ncols, 16);
    return nnt >= 4 ? 4 : nnt;
}

extern "C" int otvae_conv_fwd(const otvae_conv_geom* gg, const float* x, const float* scale, const float* shift, int relu,
                              const float* wT, const float* bias, const float* residual, float* y, void* stream) {
    int rc = check_geom(gg, "otvae_conv_fwd");
    if (rc) return rc;
    OTVAE_REQUIRE(x && wT && y, "otvae_conv_fwd: NULL tensor");
    OTVAE_REQUIRE((scale == nullptr) == (shift == nullptr), "otvae_conv_fwd: scale and shift must be given together");
    Geom g = to_geom(gg);
    const int NT = pick_nt(g.Cn);
    const unsigned M = (unsigned)g.N * g.Ho * g.Wo;
    dim3 grid(imin(cdiv(M, 64), 16384), cdiv(cdiv(g.Cn, 16), NT));
    hipStream_t st = (hipStream_t)stream;
    if (g.Cs % 4 == 0)
        launch_fwd<false>(NT, grid, st, g, x, scale, shift, relu, wT, bias, residual, y);
    else
        launch_fwd<true>(NT, grid, st, g, x, scale, shift, relu, wT, bias, residual, y);
    OTVAE_CHECK_LAUNCH("otvae_conv_fwd");
    return OTVAE_OK;
}

// ------------------------------------------------------------------------------------------------ weight transpose
__global__ void weight_transpose_kernel(const float* __restrict__ wT, float* __restrict__ wD, int T, int Cs, int Cn) {
    const size_t total = (size_t)T * Cs * Cn;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        // i indexes wD[t][n][c]
        const int c = i % Cs;
        const size_t r = i / Cs;
        const int n = r % Cn;
        const int t = r / Cn;
        wD[i] = wT[((size_t)t * Cs + c) * Cn + n];
    }
}

extern "C" int otvae_weight_transpose(const float* wT, float* wD, int T, int Cs, int Cn, void* stream) {
    OTVAE_REQUIRE(wT && wD && T > 0 && Cs > 0 && Cn > 0, "otvae_weight_transpose: bad argument");
    const size_t total = (size_t)T * Cs * Cn;
    weight_transpose_kernel<<<imin(cdiv(total, 256), 2048), 256, 0, (hipStream_t)stream>>>(wT, wD, T, Cs, Cn);
    OTVAE_CHECK_LAUNCH("otvae_weight_transpose");
    return OTVAE_OK;
}

// ------------------------------------------------------------------------------------------------ data gradient
// Rows enumerate positions of the conv input (after up-sampling) so that a lane's 4 accumulator registers are
//   up == 2     : the 4 nearest-upsample children of one source pixel (summed in the epilogue)
//   stride == 2 : 4 consecutive positions of ONE parity class (blockIdx.z), whose valid taps are uniform
//   otherwise   : 4 consecutive positions.
__device__ __forceinline__ void dgrad_row_to_pos(const Geom& g, unsigned row, int py, int px, int& n, int& iy, int& ix) {
    if (g.up == 2) {
        const unsigned parent = row >> 2, child = row & 3;
        const int sx = parent % g.Ws;
        const unsigned t = parent / g.Ws;
        const int sy = t % g.Hs;
        n = t / g.Hs;
        iy = 2 * sy + (child >> 1);
        ix = 2 * sx + (child & 1);
    } else if (g.stride == 2) {
        const int W2 = g.Ws >> 1, H2 = g.Hs >> 1;
        const int jx = row % W2;
        const unsigned t = row / W2;
        const int jy = t % H2;
        n = t / H2;
        iy = 2 * jy + py;
        ix = 2 * jx + px;
    } else {
        ix = row % g.Ws;
        const unsigned t = row / g.Ws;
        iy = t % g.Hs;
        n = t / g.Hs;
    }
}

template <int NT, bool SMALLC>
__global__ __launch_bounds__(256) void conv_dgrad_kernel(Geom g, const float* __restrict__ gy, const float* __restrict__ wD,
                                                         const float* __restrict__ x, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, int relu,
                                                         const float* __restrict__ mean, const float* __restrict__ invstd,
                                                         float* __restrict__ gv, double* __restrict__ partial, int CsPad) {
    __shared__ double red[4][2][16 * NT];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const int cls = blockIdx.z;
    const int py = cls >> 1, px = cls & 1;  // parity class (stride 2 only)
    const int Hu = g.Hs * g.up, Wu = g.Ws * g.up;
    const unsigned rows = (g.stride == 2) ? (unsigned)g.N * (g.Hs >> 1) * (g.Ws >> 1) : (unsigned)g.N * Hu * Wu;
    const unsigned ntiles = (rows + 63) / 64;
    const int c0col = blockIdx.y * (16 * NT);
    const int kh0 = (g.stride == 2) ? ((py + g.pad) & 1) : 0;
    const int kw0 = (g.stride == 2) ? ((px + g.pad) & 1) : 0;

    // BatchNorm-backward sums in fp64: sum(gv*xhat) cancels heavily and the reference accumulates in double too
    double s1[NT], s2[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) s1[j] = s2[j] = 0.0;

    for (unsigned tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const unsigned row = tile * 64 + wave * 16 + r16;
        const bool rv = row < rows;
        int n = 0, iy = 0, ix = 0;
        if (rv) dgrad_row_to_pos(g, row, py, px, n, iy, ix);
        f32x4 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};

        for (int kh = kh0; kh < g.KH; kh += g.stride) {
            const int ty = iy + g.pad - kh;
            const int oy = ty / g.stride;  // exact for valid taps
            for (int kw = kw0; kw < g.KW; kw += g.stride) {
                const int tx = ix + g.pad - kw;
                const int ox = tx / g.stride;
                const bool inb = rv && ty >= 0 && tx >= 0 && oy < g.Ho && ox < g.Wo;
                const unsigned base = (((unsigned)n * g.Ho + oy) * g.Wo + ox) * g.Cn;
                const float* wtap = wD + (size_t)(kh * g.KW + kw) * g.Cn * g.Cs;
                if constexpr (!SMALLC) {
                    for (int n0 = 0; n0 < g.Cn; n0 += 4) {
                        const int co = n0 + kq;
                        const float a = inb ? gy[base + co] : 0.f;
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
                            const int col = c0col + j * 16 + r16;
                            const float b = col < g.Cs ? wtap[(size_t)co * g.Cs + col] : 0.f;
                            acc[j] = mfma16(a, b, acc[j]);
                        }
                    }
                } else {
                    for (int n0 = 0; n0 < g.Cn; n0 += 4) {
                        const int co = n0 + kq;
                        const bool cv = co < g.Cn;
                        const float a = (inb && cv) ? gy[base + co] : 0.f;
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
                            const int col = c0col + j * 16 + r16;
                            const float b = (cv && col < g.Cs) ? wtap[(size_t)co * g.Cs + col] : 0.f;
                            acc[j] = mfma16(a, b, acc[j]);
                        }
                    }
                }
            }
        }
        // epilogue
        const unsigned row0 = tile * 64 + wave * 16 + kq * 4;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = c0col + j * 16 + r16;
            if (col >= g.Cs) continue;
            const float sc = scale ? scale[col] : 1.f, sh = scale ? shift[col] : 0.f;
            const float mu = mean ? mean[col] : 0.f, is = mean ? invstd[col] : 0.f;
            if (g.up == 2) {
                const unsigned parent = row0 >> 2;
                if (row0 < rows) {
                    float val = (acc[j][0] + acc[j][1]) + (acc[j][2] + acc[j][3]);
                    const size_t o = (size_t)parent * g.Cs + col;
                    float xv = 0.f;
                    if (relu || mean) xv = x[o];
                    if (relu) {
                        const float v = scale ? fmaf(xv, sc, sh) : xv;
                        val = v > 0.f ? val : 0.f;
                    }
                    gv[o] = val;
                    if (mean) {
                        s1[j] += (double)val;
                        s2[j] += (double)val * (double)((xv - mu) * is);
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned rr = row0 + r;
                    if (rr < rows) {
                        int pn, piy, pix;
                        dgrad_row_to_pos(g, rr, py, px, pn, piy, pix);
                        const size_t o = ((size_t)((unsigned)pn * g.Hs + piy) * g.Ws + pix) * g.Cs + col;
                        float val = acc[j][r];
                        float xv = 0.f;
                        if (relu || mean) xv = x[o];
                        if (relu) {
                            const float v = scale ? fmaf(xv, sc, sh) : xv;
                            val = v > 0.f ? val : 0.f;
                        }
                        gv[o] = val;
                        if (mean) {
                            s1[j] += (double)val;
                            s2[j] += (double)val * (double)((xv - mu) * is);
                        }
                    }
                }
            }
        }
    }
    if (mean) {
        // fixed-order reduction: 4 lane groups of a wave (shuffle), then the 4 waves (LDS), one partial per block
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            double a = s1[j], b = s2[j];
            a += __shfl_xor(a, 16, 64);
            a += __shfl_xor(a, 32, 64);
            b += __shfl_xor(b, 16, 64);
            b += __shfl_xor(b, 32, 64);
            if (kq == 0) {
                red[wave][0][j * 16 + r16] = a;
                red[wave][1][j * 16 + r16] = b;
            }
        }
        __syncthreads();
        if (threadIdx.x < 2 * 16 * NT) {
            const int which = threadIdx.x / (16 * NT), cc = threadIdx.x % (16 * NT);
            const double t = (red[0][which][cc] + red[1][which][cc]) + (red[2][which][cc] + red[3][which][cc]);
            const unsigned p = blockIdx.z * gridDim.x + blockIdx.x;
            partial[((size_t)p * 2 + which) * CsPad + c0col + cc] = t;
        }
    }
}

static void dgrad_grid(const Geom& g, int& NT, dim3& grid, int& CsPad) {
    NT = pick_nt(g.Cs);
    const int Hu = g.Hs * g.up, Wu = g.Ws * g.up;
    const unsigned rows = (g.stride == 2) ? (unsigned)g.N * (g.Hs >> 1) * (g.Ws >> 1) : (unsigned)g.N * Hu * Wu;
    const int ny = cdiv(cdiv(g.Cs, 16), NT);
    const int nz = g.stride == 2 ? 4 : 1;
    // bounded number of blocks (each writes one BatchNorm partial): <= 512 over x*z
    grid = dim3(imax(1, imin(cdiv(rows, 64), 512 / nz)), ny, nz);
    CsPad = ny * 16 * NT;
}

extern "C" int otvae_conv_bwd_data_ws(const otvae_conv_geom* gg, int* P, int* CsPad) {
    int rc = check_geom(gg, "otvae_conv_bwd_data_ws");
    if (rc) return rc;
    Geom g = to_geom(gg);
    int NT;
    dim3 grid;
    int cp;
    dgrad_grid(g, NT, grid, cp);
    if (P) *P = grid.x * grid.z;
    if (CsPad) *CsPad = cp;
    return OTVAE_OK;
}

template <bool SMALLC>
static void launch_dgrad(int NT, dim3 grid, hipStream_t st, Geom g, const float* gy, const float* wD, const float* x,
                         const float* scale, const float* shift, int relu, const float* mean, const float* invstd, float* gv,
                         double* partial, int CsPad) {
#define OTVAE_DG(N_) \
    conv_dgrad_kernel<N_, SMALLC><<<grid, 256, 0, st>>>(g, gy, wD, x, scale, shift, relu, mean, invstd, gv, partial, CsPad)
    switch (NT) {
        case 1: OTVAE_DG(1); break;
        case 2: OTVAE_DG(2); break;
        case 3: OTVAE_DG(3); break;
        default: OTVAE_DG(4); break;
    }
#undef OTVAE_DG
}

extern "C" int otvae_conv_bwd_data(const otvae_conv_geom* gg, const float* gy, const float* wD, const float* x,
                                   const float* scale, const float* shift, int relu, const float* mean, const float* invstd,
                                   float* gv, double* bn_partial, void* stream) {
    int rc = check_geom(gg, "otvae_conv_bwd_data");
    if (rc) return rc;
    OTVAE_REQUIRE(gy && wD && gv, "otvae_conv_bwd_data: NULL tensor");
    OTVAE_REQUIRE((scale == nullptr) == (shift == nullptr), "otvae_conv_bwd_data: scale/shift must come together");
    OTVAE_REQUIRE((mean == nullptr) == (invstd == nullptr), "otvae_conv_bwd_data: mean/invstd must come together");
    OTVAE_REQUIRE(!(relu || mean) || x, "otvae_conv_bwd_data: x needed for the ReLU mask / BatchNorm sums");
    OTVAE_REQUIRE(!mean || bn_partial, "otvae_conv_bwd_data: bn_partial workspace missing");
    Geom g = to_geom(gg);
    int NT, CsPad;
    dim3 grid;
    dgrad_grid(g, NT, grid, CsPad);
    hipStream_t st = (hipStream_t)stream;
    if (g.Cn % 4 == 0)
        launch_dgrad<false>(NT, grid, st, g, gy, wD, x, scale, shift, relu, mean, invstd, gv, bn_partial, CsPad);
    else
        launch_dgrad<true>(NT, grid, st, g, gy, wD, x, scale, shift, relu, mean, invstd, gv, bn_partial, CsPad);
    OTVAE_CHECK_LAUNCH("otvae_conv_bwd_data");
    return OTVAE_OK;
}

// ------------------------------------------------------------------------------------------------ weight gradient
// Job = (pixel chunk pc, k-tile group, n-tile group) per WAVE.  A wave keeps KT x NT accumulator tiles
// (KT = 8/NT) and walks its pixel chunk 4 pixels per MFMA.  Row K (after the last tap*channel row) is the bias row:
// its A operand is 1 for valid pixels, so dBias falls out of the same MFMA stream.
template <int NT>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(Geom g, const float* __restrict__ x, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, int relu,
                                                         const float* __restrict__ gy, float* __restrict__ partial, int P,
                                                         int Kp /* K + has_bias */, int has_bias, unsigned chunk, int nkg,
                                                         int nng) {
    constexpr int KT = 8 / NT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const unsigned job = blockIdx.x * 4 + wave;
    const unsigned njobs = (unsigned)P * nkg * nng;
    if (job >= njobs) return;  // wave-uniform
    const int kgi = job % nkg;
    const int ngi = (job / nkg) % nng;
    const unsigned pc = job / (nkg * nng);
    const unsigned M = (unsigned)g.N * g.Ho * g.Wo;
    const unsigned mbeg = pc * chunk, mend = min(M, mbeg + chunk);
    const int K = g.KH * g.KW * g.Cs;
    const int Hu = g.Hs * g.up, Wu = g.Ws * g.up;
    const int ush = g.up - 1;
    const int n0 = ngi * 16 * NT;
    const int nkt = (Kp + 15) / 16;

    // per-lane description of its row k in each owned k-tile
    int kdy[KT], kdx[KT], kc[KT];
    int kkind[KT];  // 0 = invalid/padding, 1 = tap row, 2 = bias row
    bool tile_on[KT];
#pragma unroll
    for (int i = 0; i < KT; ++i) {
        const int kt = kgi * KT + i;
        const int k = kt * 16 + r16;
        kkind[i] = 0;
        kdy[i] = kdx[i] = kc[i] = 0;
        if (kt < nkt && k < K) {
            const int t = k / g.Cs;
            kc[i] = k - t * g.Cs;
            const int kh = t / g.KW;
            kdy[i] = kh - g.pad;
            kdx[i] = (t - kh * g.KW) - g.pad;
            // a tap that never touches the image for any output pixel contributes nothing: switch the row off
            const bool ytouch = (kdy[i] + (g.Ho - 1) * g.stride >= 0) && (kdy[i] < Hu);
            const bool xtouch = (kdx[i] + (g.Wo - 1) * g.stride >= 0) && (kdx[i] < Wu);
            kkind[i] = (ytouch && xtouch) ? 1 : 0;
        } else if (kt < nkt && has_bias && k == K) {
            kkind[i] = 2;
        }
        tile_on[i] = __ballot(kkind[i] != 0) != 0ull;  // wave-uniform
    }

    f32x4 acc[KT][NT];
#pragma unroll
    for (int i = 0; i < KT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (unsigned m4 = mbeg; m4 < mend; m4 += 4) {
        const unsigned m = m4 + kq;
        const bool mv = m < mend;
        int ox = 0, oy = 0, n = 0;
        if (mv) {
            ox = m % g.Wo;
            unsigned t = m / g.Wo;
            oy = t % g.Ho;
            n = t / g.Ho;
        }
        float b[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = n0 + j * 16 + r16;
            b[j] = (mv && col < g.Cn) ? gy[(size_t)m * g.Cn + col] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < KT; ++i) {
            if (!tile_on[i]) continue;
            float a = 0.f;
            if (kkind[i] == 1) {
                const int iy = oy * g.stride + kdy[i], ix = ox * g.stride + kdx[i];
                if (mv && iy >= 0 && iy < Hu && ix >= 0 && ix < Wu) {
                    const unsigned idx = (((unsigned)n * g.Hs + (iy >> ush)) * g.Ws + (ix >> ush)) * g.Cs + kc[i];
                    a = act_load(x, idx, kc[i], scale, shift, relu);
                }
            } else if (kkind[i] == 2) {
                a = mv ? 1.f : 0.f;
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = mfma16(a, b[j], acc[i][j]);
        }
    }
    // store partial[pc][k][n]
    float* out = partial + (size_t)pc * Kp * g.Cn;
#pragma unroll
    for (int i = 0; i < KT; ++i) {
        const int kt = kgi * KT + i;
        if (kt >= nkt) continue;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = n0 + j * 16 + r16;
            if (col >= g.Cn) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = kt * 16 + kq * 4 + r;
                if (k < Kp) out[(size_t)k * g.Cn + col] = acc[i][j][r];
            }
        }
    }
}

// out[e] = sum_p partial[p][e] in a fixed order: 4 interleaved p-lanes per element, then p-lane 0..3.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int P, int K, int Kp, int Cn,
                                                           float* __restrict__ gw, float* __restrict__ gb) {
    __shared__ float red[4][64];
    const size_t total = (size_t)Kp * Cn;
    const int el = threadIdx.x & 63, pg = threadIdx.x >> 6;
    for (size_t e0 = (size_t)blockIdx.x * 64; e0 < total; e0 += (size_t)gridDim.x * 64) {
        const size_t e = e0 + el;
        float s = 0.f;
        if (e < total)
            for (int p = pg; p < P; p += 4) s += partial[(size_t)p * total + e];
        red[pg][el] = s;
        __syncthreads();
        if (pg == 0 && e < total) {
            const float t = (red[0][el] + red[1][el]) + (red[2][el] + red[3][el]);
            const int k = e / Cn;
            if (k < K)
                gw[e] = t;
            else if (gb)
                gb[e - (size_t)K * Cn] = t;
        }
        __syncthreads();
    }
}

static void wgrad_plan(const Geom& g, int has_bias, int& NT, int& P, unsigned& chunk, int& nkg, int& nng, int& Kp) {
    NT = pick_nt(g.Cn);
    if (NT == 3) NT = 4;  // KT = 8/NT must be integral
    const int KT = 8 / NT;
    Kp = g.KH * g.KW * g.Cs + (has_bias ? 1 : 0);
    const int nkt = cdiv(Kp, 16);
    nkg = cdiv(nkt, KT);
    nng = cdiv(cdiv(g.Cn, 16), NT);
    const unsigned M = (unsigned)g.N * g.Ho * g.Wo;
    // enough wave-jobs to fill the chip (256 CUs x 4 SIMDs x ~2), pixel chunks of >= 64 pixels, workspace <= 16 MiB
    int want = cdiv(2048, nkg * nng);
    int maxp_pix = imax(1, (int)(M / 64));
    int maxp_ws = imax(1, (int)((4u << 20) / ((unsigned)Kp * g.Cn)));
    P = imax(1, imin(imin(want, maxp_pix), imin(maxp_ws, 512)));
    chunk = ((M + P - 1) / P + 3) & ~3u;
    P = cdiv(M, chunk);
}

extern "C" int otvae_conv_bwd_weight_ws(const otvae_conv_geom* gg, int has_bias, int* P) {
    int rc = check_geom(gg, "otvae_conv_bwd_weight_ws");
    if (rc) return rc;
    Geom g = to_geom(gg);
    int NT, p, nkg, nng, Kp;
    unsigned chunk;
    wgrad_plan(g, has_bias, NT, p, chunk, nkg, nng, Kp);
    if (P) *P = p;
    return OTVAE_OK;
}

extern "C" int otvae_conv_bwd_weight(const otvae_conv_geom* gg, const float* x, const float* scale, const float* shift,
                                     int relu, const float* gy, int has_bias, float* partial, float* gw, float* gb,
                                     void* stream) {
    int rc = check_geom(gg, "otvae_conv_bwd_weight");
    if (rc) return rc;
    OTVAE_REQUIRE(x && gy && partial && gw, "otvae_conv_bwd_weight: NULL tensor");
    OTVAE_REQUIRE(!has_bias || gb, "otvae_conv_bwd_weight: gb missing");
    OTVAE_REQUIRE((scale == nullptr) == (shift == nullptr), "otvae_conv_bwd_weight: scale/shift must come together");
    Geom g = to_geom(gg);
    int NT, P, nkg, nng, Kp;
    unsigned chunk;
    wgrad_plan(g, has_bias, NT, P, chunk, nkg, nng, Kp);
    const unsigned njobs = (unsigned)P * nkg * nng;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(cdiv(njobs, 4));
    switch (NT) {
        case 1: conv_wgrad_kernel<1><<<grid, 256, 0, st>>>(g, x, scale, shift, relu, gy, partial, P, Kp, has_bias, chunk, nkg, nng); break;
        case 2: conv_wgrad_kernel<2><<<grid, 256, 0, st>>>(g, x, scale, shift, relu, gy, partial, P, Kp, has_bias, chunk, nkg, nng); break;
        default: conv_wgrad_kernel<4><<<grid, 256, 0, st>>>(g, x, scale, shift, relu, gy, partial, P, Kp, has_bias, chunk, nkg, nng); break;
    }
    OTVAE_CHECK_LAUNCH("otvae_conv_bwd_weight");
    const size_t total = (size_t)Kp * g.Cn;
    wgrad_reduce_kernel<<<imin(cdiv(total, 64), 2048), 256, 0, st>>>(partial, P, Kp - (has_bias ? 1 : 0), Kp, g.Cn, gw, gb);
    OTVAE_CHECK_LAUNCH("otvae_conv_bwd_weight(reduce)");
    return OTVAE_OK;
}
