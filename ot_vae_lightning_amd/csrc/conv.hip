// ConvLayer forward / data-gradient / weight-gradient as LDS-staged implicit GEMMs on the fp32 MFMA
// (v_mfma_f32_16x16x4_f32), CDNA4 / gfx950.
//
// Replaces the ATen sequence  batch_norm -> relu -> upsample_nearest2d -> conv2d (+bias, +residual add)  issued by
// ConvLayer.forward (reference networks/cnn.py:183-192) and its autograd backward, for the layer geometries the
// CNN builder produces (3x3 s1 p1, 4x4 s2 p1, 1x1, optional nearest x2 up-sampling before the conv).
//
// GEMM view, NHWC activations, HWIO weights:
//   fwd   Y[m][n]  = sum_k A[m][k] W[k][n]        A[m][k] = act(x)[pixel(m) + tap(k)][chan(k)]   rows = output pixels
//   dgrad dU[q][c] = sum_k G[q][k] Wd[k][c]       G[q][k] = gy[pixel(q) - tap(k)][chan(k)]       rows = input positions
//   wgrad dW[k][n] = sum_m A[m][k] gy[m][n]       rows = k (taps*channels, + one bias row), reduction over pixels
// A workgroup (4 waves) owns a 64-row x (16*NT)-column tile.  Per K-chunk of 32 the 256 threads gather the A tile
// (normalisation, ReLU, nearest up-sampling and zero padding applied on the way) and the weight tile into LDS with
// all loads of the chunk in flight at once, double-buffered against the MFMAs of the previous chunk; each wave then
// feeds v_mfma_f32_16x16x4_f32 from LDS (A[row = lane&15][k = lane>>4], B[k = lane>>4][col = lane&15],
// D[r] = D[(lane>>4)*4 + r][lane&15]).  LDS leading dimensions (KC+2 / BN+16) make both operand reads conflict-free.
// Taps that cannot touch the image for any row of the layer (e.g. 8 of the 9 taps of a 3x3 conv on a 1x1 map) are
// dropped from K at kernel start.
#include "common.h"
#include "conv_small.h"  // struct Geom + the direct VALU kernels used when both channel counts are tiny
#include "conv_tile.h"   // image-tile MFMA convolution (whole images in LDS) for maps up to 16x16
#include "conv_wtile.h"  // image-tile MFMA weight gradient

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
    OTVAE_REQUIRE(g->Cs <= 2048 && g->Cn <= 2048, "%s: more than 2048 channels unsupported", who);
    if (g->stride == 2) OTVAE_REQUIRE(g->Hs % 2 == 0 && g->Ws % 2 == 0, "%s: stride 2 needs even input size", who);
    return OTVAE_OK;
}

// Column tiles (16 wide) per wave.  A wide tile re-uses the staged A rows for more MFMAs, but a layer with few row tiles
// (deep layers: 1x1 .. 4x4 maps) then runs on a fraction of the 256 CUs with ONE wave per SIMD, where the loop is bound
// by instruction issue, not by MFMA throughput: narrow the tile until the launch has >= 256 workgroups (one per CU;
// measured: thresholds of 192..384 are within 1 % of each other for the whole step, 128 and 1024 are 2 % slower).
static inline int pick_nt(int ncols, long row_blocks) {
    const int nnt = cdiv(ncols, 16);
    int nt = nnt >= 4 ? 4 : nnt;
    while (nt > 1 && row_blocks * cdiv(nnt, nt) < 256) --nt;
    return nt;
}

#define MAX_TAPS 49
#define KC 32  // K-chunk staged per pipeline step
#define TM 64  // rows per workgroup tile
#define PAD_MARK 0x7fc00001  // NaN payload marking "padding / out of range": becomes an exact 0 AFTER the activation
#define ONE_MARK 0x7fc00002  // bias row of the weight gradient: exact 1

// exact floor(k / d) for 0 <= k < 2^22, d >= 1 with one multiply: (k+0.5)/d is at least 0.5/d away from an integer
// while the float product (rounded reciprocal, rounded product) is off by < 2^-23 * (k+0.5)/d, which is < 0.5/d.
__device__ __forceinline__ int fast_div(int k, float inv_d) { return (int)(((float)k + 0.5f) * inv_d); }
__device__ __forceinline__ int imax_dev(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int imin_dev(int a, int b) { return a < b ? a : b; }

// ------------------------------------------------------------------------------------------------ fwd + dgrad
// MODE 0 (FWD):   S = x  [N][Hs][Ws][Cs], CK = Cs, NC = Cn, Bmat = HWIO weight [T][Cs][Cn]
// MODE 1 (DGRAD): S = gy [N][Ho][Wo][Cn], CK = Cn, NC = Cs, Bmat = wD [T][Cn][Cs]
// Rows of DGRAD are grouped so that a lane's 4 accumulator registers are the 4 nearest-upsample children of one source
// pixel (up == 2), or 4 positions of ONE stride-parity class (stride == 2, class = blockIdx.z).
// Row index -> (image, y, x) decodes divide by launch constants.  A 32-bit udiv expands to ~25 vector instructions and
// these decodes sit in per-tile / per-output-element code of kernels that are bound by vector-instruction issue, so
// they go through fast_div (one multiply) whenever the row count is below 2^22 (exactness bound of fast_div).
struct RowDiv {
    float iWo, iHo, iWs, iHs, iW2, iH2;
    bool small;
};
__device__ __forceinline__ RowDiv make_rowdiv(const Geom& g, unsigned rows_max) {
    RowDiv r;
    r.iWo = 1.0f / (float)g.Wo, r.iHo = 1.0f / (float)g.Ho;
    r.iWs = 1.0f / (float)g.Ws, r.iHs = 1.0f / (float)g.Hs;
    r.iW2 = 1.0f / (float)imax_dev(g.Ws >> 1, 1), r.iH2 = 1.0f / (float)imax_dev(g.Hs >> 1, 1);
    r.small = rows_max < (1u << 22);
    return r;
}
__device__ __forceinline__ unsigned rdiv(unsigned k, unsigned d, float inv_d, bool small) {
    return small ? (unsigned)fast_div((int)k, inv_d) : k / d;
}

__device__ __forceinline__ void dgrad_row_to_pos(const Geom& g, const RowDiv& rd, unsigned row, int py, int px, int& n, int& iy,
                                                 int& ix) {
    if (g.up == 2) {
        const unsigned parent = row >> 2, child = row & 3;
        const unsigned t = rdiv(parent, g.Ws, rd.iWs, rd.small);
        const int sx = parent - t * g.Ws;
        n = rdiv(t, g.Hs, rd.iHs, rd.small);
        const int sy = t - (unsigned)n * g.Hs;
        iy = 2 * sy + (child >> 1);
        ix = 2 * sx + (child & 1);
    } else if (g.stride == 2) {
        const int W2 = g.Ws >> 1, H2 = g.Hs >> 1;
        const unsigned t = rdiv(row, W2, rd.iW2, rd.small);
        const int jx = row - t * W2;
        n = rdiv(t, H2, rd.iH2, rd.small);
        const int jy = t - (unsigned)n * H2;
        iy = 2 * jy + py;
        ix = 2 * jx + px;
    } else {
        const unsigned t = rdiv(row, g.Ws, rd.iWs, rd.small);
        ix = row - t * g.Ws;
        n = rdiv(t, g.Hs, rd.iHs, rd.small);
        iy = t - (unsigned)n * g.Hs;
    }
}

// LDS layout of one workgroup (carved out of a raw buffer so that the same body can run inside the multi-job kernel)
template <int NT>
struct GemmSmem {
    static constexpr int BN = 16 * NT;
    static constexpr int LDA = KC + 2;
    static constexpr int LDB = BN + ((BN % 32 == 0) ? 16 : 0);
    static constexpr size_t as_off = 0;
    static constexpr size_t bs_off = as_off + sizeof(float) * 2 * TM * LDA;
    static constexpr size_t red_off = bs_off + sizeof(float) * 2 * KC * LDB;
    static constexpr size_t tap_off = red_off + sizeof(double) * 4 * 2 * BN;
    static constexpr size_t row_off = tap_off + sizeof(int) * (3 * MAX_TAPS + 1);
    static constexpr size_t bytes = (row_off + sizeof(int) * 3 * TM + 15) / 16 * 16;
};

// (bx, by, bz) / (gx, gz): the block's coordinates and the extent of the (virtual) grid of THIS layer
template <int MODE, int NT, bool VEC, bool UT = false>
__device__ __forceinline__ void conv_gemm_body(char* __restrict__ smem, const Geom& g, const float* __restrict__ S,
                                               const float* __restrict__ scale, const float* __restrict__ shift, int relu,
                                               const float* __restrict__ Bmat,
                                               // FWD epilogue
                                               const float* __restrict__ bias, const float* __restrict__ res,
                                               float* __restrict__ y,
                                               // DGRAD epilogue
                                               const float* __restrict__ xin, const float* __restrict__ mean,
                                               const float* __restrict__ invstd, float* __restrict__ gv,
                                               double* __restrict__ partial, int CsPad, int bx, int by, int bz, int gx,
                                               int gz) {
    using SM = GemmSmem<NT>;
    constexpr int BN = SM::BN;
    constexpr int LDA = SM::LDA;
    constexpr int LDB = SM::LDB;
    constexpr int NB_ELEMS = KC * BN / 256;  // weight-tile elements staged per thread per chunk
    float(*As)[TM * LDA] = reinterpret_cast<float(*)[TM * LDA]>(smem + SM::as_off);
    float(*Bs)[KC * LDB] = reinterpret_cast<float(*)[KC * LDB]>(smem + SM::bs_off);
    double* red = reinterpret_cast<double*>(smem + SM::red_off);
    int* tap_dy = reinterpret_cast<int*>(smem + SM::tap_off);
    int* tap_dx = tap_dy + MAX_TAPS;
    int* tap_w = tap_dx + MAX_TAPS;
    int& s_ntaps = tap_w[MAX_TAPS];
    int* row_n = reinterpret_cast<int*>(smem + SM::row_off);
    int* row_y = row_n + TM;
    int* row_x = row_y + TM;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const int cls = bz, py = cls >> 1, px = cls & 1;
    const int Hu = g.Hs * g.up, Wu = g.Ws * g.up;
    const int CK = MODE == 0 ? g.Cs : g.Cn;
    const int NC = MODE == 0 ? g.Cn : g.Cs;
    const int n0 = by * BN;
    // source geometry seen by the gather: a tap is in range iff 0 <= t < lim, source coordinate = t >> sh
    const int lim_y = MODE == 0 ? Hu : g.Ho * g.stride, lim_x = MODE == 0 ? Wu : g.Wo * g.stride;
    const int sh = MODE == 0 ? g.up - 1 : g.stride - 1;
    const int srcH = MODE == 0 ? g.Hs : g.Ho, srcW = MODE == 0 ? g.Ws : g.Wo;
    const unsigned rows = MODE == 0 ? (unsigned)g.N * g.Ho * g.Wo
                                    : (g.stride == 2 ? (unsigned)g.N * (g.Hs >> 1) * (g.Ws >> 1) : (unsigned)g.N * Hu * Wu);
    const unsigned ntiles = (rows + TM - 1) / TM;
    const RowDiv rd = make_rowdiv(g, rows);

    // ---- valid tap list (uniform over the launch, resp. over the parity class)
    if (tid == 0) {
        int nt = 0;
        for (int kh = 0; kh < g.KH; ++kh)
            for (int kw = 0; kw < g.KW; ++kw) {
                bool ok;
                int dy, dx;
                if (MODE == 0) {
                    // row anchor = (oy*stride, ox*stride); tap offset = kh - pad
                    dy = kh - g.pad;
                    dx = kw - g.pad;
                    ok = (dy + (g.Ho - 1) * g.stride >= 0) && (dy < Hu) && (dx + (g.Wo - 1) * g.stride >= 0) && (dx < Wu);
                } else {
                    // row anchor = (iy, ix); t = iy + pad - kh must be a multiple of stride inside [0, Ho*stride)
                    dy = g.pad - kh;
                    dx = g.pad - kw;
                    ok = true;
                    if (g.stride == 2) ok = (((py + dy) & 1) == 0) && (((px + dx) & 1) == 0);
                    ok = ok && (Hu - 1 + dy >= 0) && (dy < lim_y) && (Wu - 1 + dx >= 0) && (dx < lim_x);
                }
                if (ok) {
                    tap_dy[nt] = dy;
                    tap_dx[nt] = dx;
                    tap_w[nt] = kh * g.KW + kw;
                    ++nt;
                }
            }
        s_ntaps = nt;
    }
    __syncthreads();
    const int ntaps = s_ntaps;
    const int K = ntaps * CK;
    const int nch = (K + KC - 1) / KC;
    const float inv_ck = 1.0f / (float)CK;
    // UT (host-selected when VEC and CK % KC == 0): the uniform-tap pipeline below replaces the general staging
    const int cpt = CK / KC;  // chunks per tap (ut)
    const float inv_cpt = 1.0f / (float)(cpt > 0 ? cpt : 1);

    // staging role for the A tile: one k per thread, rows a_r0 + 8 i
    const int a_kl = tid & 31, a_r0 = tid >> 5;

    double s1[NT], s2[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) s1[j] = s2[j] = 0.0;

    for (unsigned tile = bx; tile < ntiles; tile += gx) {
        __syncthreads();  // readers of row_* / LDS buffers of the previous tile are done
        if (tid < TM) {
            const unsigned row = tile * TM + tid;
            int n = -1, ry = 0, rx = 0;
            if (row < rows) {
                if (MODE == 0) {
                    const unsigned t = rdiv(row, g.Wo, rd.iWo, rd.small);
                    const int ox = row - t * g.Wo;
                    n = rdiv(t, g.Ho, rd.iHo, rd.small);
                    const int oy = t - (unsigned)n * g.Ho;
                    ry = oy * g.stride;
                    rx = ox * g.stride;
                } else {
                    int iy, ix;
                    dgrad_row_to_pos(g, rd, row, py, px, n, iy, ix);
                    ry = iy;
                    rx = ix;
                }
            }
            row_n[tid] = n;
            row_y[tid] = ry;
            row_x[tid] = rx;
        }
        __syncthreads();

        f32x4 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};

        // Staged operands travel through registers between the global load (issued one chunk ahead) and the LDS store.
        // VEC: channel counts are multiples of 4 -> float4 gathers along the channel (k) dimension, 4x fewer address
        // computations and memory instructions; otherwise one scalar per slot.
        constexpr int NBV = (8 * BN + 255) / 256;  // float4 weight slots per thread (VEC)
        float areg[8], breg[NB_ELEMS];
        float4 areg4[2], breg4[NBV];
        float4 a_sc4 = make_float4(1.f, 1.f, 1.f, 1.f), a_sh4 = make_float4(0.f, 0.f, 0.f, 0.f);
        int a_ok = 0;
        float a_sc = 1.f, a_sh = 0.f;
        // VEC role: k4 = tid & 7 (4 consecutive k), rows (tid >> 3) + 32 i
        const int v_k4 = tid & 7, v_r0 = tid >> 3;
        int vr_n[2], vr_y[2], vr_x[2];
        if constexpr (VEC) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                vr_n[i] = row_n[v_r0 + 32 * i];
                vr_y[i] = row_y[v_r0 + 32 * i];
                vr_x[i] = row_x[v_r0 + 32 * i];
            }
        }

        // Uniform-tap fast path (CK % KC == 0, i.e. 32, 64, 96 ... channels: every deep layer): a K-chunk lies inside
        // ONE tap, so the tap decode is per chunk instead of per thread, the rows' source pixels and bounds tests are
        // recomputed only when the tap changes (every CK/32 chunks), and the weight rows of a chunk are contiguous.
        // These kernels are bound by vector-instruction issue; this removes ~60 % of the staging instructions.
        int u_tl = -1, u_ok = 0, u_tw = 0;
        unsigned u_off[2] = {0u, 0u};
        auto stage_load = [&](int ch) {
            if constexpr (VEC) {
                const int k = ch * KC + v_k4 * 4;
                const bool kv = k < K;
                const int tl = kv ? fast_div(k, inv_ck) : 0;
                const int c = k - tl * CK;
                const int dy = tap_dy[tl], dx = tap_dx[tl];
                if (MODE == 0 && scale != nullptr && kv) {
                    a_sc4 = *reinterpret_cast<const float4*>(scale + c);
                    a_sh4 = *reinterpret_cast<const float4*>(shift + c);
                }
                a_ok = 0;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int ty = vr_y[i] + dy, tx = vr_x[i] + dx;
                    areg4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (kv && vr_n[i] >= 0 && (unsigned)ty < (unsigned)lim_y && (unsigned)tx < (unsigned)lim_x) {
                        const unsigned pix = ((unsigned)vr_n[i] * srcH + (ty >> sh)) * srcW + (tx >> sh);
                        areg4[i] = *reinterpret_cast<const float4*>(S + (size_t)pix * CK + c);
                        a_ok |= 1 << i;
                    }
                }
#pragma unroll
                for (int j = 0; j < NBV; ++j) {
                    const int e = tid + 256 * j;
                    const int kb = e / (BN / 4), c4 = e - kb * (BN / 4);
                    const int kk = ch * KC + kb;
                    breg4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (e < 8 * BN && kk < K && n0 + c4 * 4 < NC) {
                        const int t2 = fast_div(kk, inv_ck);
                        const int c2 = kk - t2 * CK;
                        breg4[j] = *reinterpret_cast<const float4*>(Bmat + ((size_t)tap_w[t2] * CK + c2) * NC + n0 + c4 * 4);
                    }
                }
                return;
            }
            // ---- A
            const int k = ch * KC + a_kl;
            const bool kv = k < K;
            const int tl = kv ? fast_div(k, inv_ck) : 0;
            const int c = k - tl * CK;
            const int dy = tap_dy[tl], dx = tap_dx[tl];
            if (MODE == 0 && scale != nullptr && kv) {
                a_sc = scale[c];
                a_sh = shift[c];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = a_r0 + 8 * i;
                const int n = row_n[r];
                const int ty = row_y[r] + dy, tx = row_x[r] + dx;
                float v = __int_as_float(PAD_MARK);
                if (kv && n >= 0 && ty >= 0 && ty < lim_y && tx >= 0 && tx < lim_x) {
                    const unsigned pix = ((unsigned)n * srcH + (ty >> sh)) * srcW + (tx >> sh);
                    v = S[(size_t)pix * CK + c];
                }
                areg[i] = v;
            }
            // ---- weights
#pragma unroll
            for (int j = 0; j < NB_ELEMS; ++j) {
                const int e = tid + 256 * j;
                const int kb = e / BN, col = e - kb * BN;
                const int kk = ch * KC + kb;
                float v = 0.f;
                if (kk < K && n0 + col < NC) {
                    const int t2 = fast_div(kk, inv_ck);
                    const int c2 = kk - t2 * CK;
                    v = Bmat[((size_t)tap_w[t2] * CK + c2) * NC + n0 + col];
                }
                breg[j] = v;
            }
        };
        auto stage_store = [&](int buf) {
            if constexpr (VEC) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    float4 v = areg4[i];
                    if ((a_ok >> i) & 1) {
                        if (MODE == 0) {
                            if (scale != nullptr) {
                                v.x = fmaf(v.x, a_sc4.x, a_sh4.x);
                                v.y = fmaf(v.y, a_sc4.y, a_sh4.y);
                                v.z = fmaf(v.z, a_sc4.z, a_sh4.z);
                                v.w = fmaf(v.w, a_sc4.w, a_sh4.w);
                            }
                            if (relu) {
                                v.x = fmaxf(v.x, 0.f);
                                v.y = fmaxf(v.y, 0.f);
                                v.z = fmaxf(v.z, 0.f);
                                v.w = fmaxf(v.w, 0.f);
                            }
                        }
                    }
                    float* dst = &As[buf][(v_r0 + 32 * i) * LDA + v_k4 * 4];  // 8-byte aligned (LDA even)
                    *reinterpret_cast<float2*>(dst) = make_float2(v.x, v.y);
                    *reinterpret_cast<float2*>(dst + 2) = make_float2(v.z, v.w);
                }
#pragma unroll
                for (int j = 0; j < NBV; ++j) {
                    const int e = tid + 256 * j;
                    if (e < 8 * BN) {
                        const int kb = e / (BN / 4), c4 = e - kb * (BN / 4);
                        *reinterpret_cast<float4*>(&Bs[buf][kb * LDB + c4 * 4]) = breg4[j];  // LDB % 4 == 0
                    }
                }
                return;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float v = areg[i];
                if (__float_as_int(v) == PAD_MARK) {
                    v = 0.f;
                } else if (MODE == 0) {
                    if (scale != nullptr) v = fmaf(v, a_sc, a_sh);
                    if (relu) v = fmaxf(v, 0.f);
                }
                As[buf][(a_r0 + 8 * i) * LDA + a_kl] = v;
            }
#pragma unroll
            for (int j = 0; j < NB_ELEMS; ++j) {
                const int e = tid + 256 * j;
                const int kb = e / BN, col = e - kb * BN;
                Bs[buf][kb * LDB + col] = breg[j];
            }
        };

        if constexpr (VEC && UT) {
            {
                // Uniform-tap pipeline, TWO chunks of operands in flight in registers (set = chunk & 1) on top of the
                // double-buffered LDS tile.  Measured on a 64->64 3x3 layer at 2x2 (18 chunks, one wave per SIMD):
                // of 18.3 us, 6 us were the exposed part of the load latency with one chunk in flight.  Every global
                // load here is unconditional (out-of-range slots read offset 0 and are zeroed at the LDS store), so the
                // vmcnt wait before a set is consumed counts exactly the one younger chunk.
                float4 ra[2][2], rb[2][NBV], rsc[2], rsh[2];
                int rok[2];
                auto ut_load = [&](int ch, float4(&a)[2], float4(&b)[NBV], float4& sc4, float4& sh4, int& ok) {
                    const int tl = fast_div(ch, inv_cpt);
                    const int cb = (ch - tl * cpt) * KC;
                    if (tl != u_tl) {  // uniform over the block; no global loads inside
                        u_tl = tl;
                        const int dy = tap_dy[tl], dx = tap_dx[tl];
                        u_tw = tap_w[tl] * CK;
                        u_ok = 0;
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            const int ty = vr_y[i] + dy, tx = vr_x[i] + dx;
                            const bool valid = vr_n[i] >= 0 && (unsigned)ty < (unsigned)lim_y && (unsigned)tx < (unsigned)lim_x;
                            u_off[i] = valid ? (((unsigned)vr_n[i] * srcH + (ty >> sh)) * srcW + (tx >> sh)) * (unsigned)CK : 0u;
                            u_ok |= (valid ? 1 : 0) << i;
                        }
                    }
                    const int c = cb + v_k4 * 4;
                    if (MODE == 0 && scale != nullptr) {
                        sc4 = *reinterpret_cast<const float4*>(scale + c);
                        sh4 = *reinterpret_cast<const float4*>(shift + c);
                    }
                    ok = u_ok;
#pragma unroll
                    for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const float4*>(S + (size_t)u_off[i] + c);
                    const float* brow = Bmat + (size_t)(u_tw + cb) * NC + n0;
#pragma unroll
                    for (int j = 0; j < NBV; ++j) {
                        const int e = tid + 256 * j;
                        const int kb = e / (BN / 4), c4 = e - kb * (BN / 4);
                        const bool valid = e < 8 * BN && n0 + c4 * 4 < NC;
                        b[j] = *reinterpret_cast<const float4*>(valid ? brow + (size_t)kb * NC + c4 * 4 : Bmat);
                        ok |= (valid ? 1 : 0) << (2 + j);
                    }
                };
                auto ut_store = [&](const float4(&a)[2], const float4(&b)[NBV], const float4& sc4, const float4& sh4, int ok,
                                    int buf) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        float4 v = a[i];
                        if (MODE == 0) {
                            if (scale != nullptr) {
                                v.x = fmaf(v.x, sc4.x, sh4.x);
                                v.y = fmaf(v.y, sc4.y, sh4.y);
                                v.z = fmaf(v.z, sc4.z, sh4.z);
                                v.w = fmaf(v.w, sc4.w, sh4.w);
                            }
                            if (relu) {
                                v.x = fmaxf(v.x, 0.f);
                                v.y = fmaxf(v.y, 0.f);
                                v.z = fmaxf(v.z, 0.f);
                                v.w = fmaxf(v.w, 0.f);
                            }
                        }
                        if (!((ok >> i) & 1)) v = make_float4(0.f, 0.f, 0.f, 0.f);  // padding is zero AFTER the activation
                        float* dst = &As[buf][(v_r0 + 32 * i) * LDA + v_k4 * 4];  // 8-byte aligned (LDA even)
                        *reinterpret_cast<float2*>(dst) = make_float2(v.x, v.y);
                        *reinterpret_cast<float2*>(dst + 2) = make_float2(v.z, v.w);
                    }
#pragma unroll
                    for (int j = 0; j < NBV; ++j) {
                        const int e = tid + 256 * j;
                        if (e < 8 * BN) {
                            const int kb = e / (BN / 4), c4 = e - kb * (BN / 4);
                            const float4 v = ((ok >> (2 + j)) & 1) ? b[j] : make_float4(0.f, 0.f, 0.f, 0.f);
                            *reinterpret_cast<float4*>(&Bs[buf][kb * LDB + c4 * 4]) = v;  // LDB % 4 == 0
                        }
                    }
                };
                const int last = nch - 1;
                ut_load(0, ra[0], rb[0], rsc[0], rsh[0], rok[0]);
                ut_load(imin_dev(1, last), ra[1], rb[1], rsc[1], rsh[1], rok[1]);
                ut_store(ra[0], rb[0], rsc[0], rsh[0], rok[0], 0);
                __syncthreads();
                for (int ch0 = 0; ch0 < nch; ch0 += 2) {
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        const int ch = ch0 + d;
                        if (ch < nch) {
                            const int buf = d;  // == ch & 1
                            ut_load(imin_dev(ch + 2, last), ra[d], rb[d], rsc[d], rsh[d], rok[d]);
                            {
                                const float* Ab = &As[buf][(wave * 16 + r16) * LDA + kq];
                                const float* Bb = &Bs[buf][kq * LDB + r16];
                                float av[KC / 4], bv[KC / 4][NT];
#pragma unroll
                                for (int ks = 0; ks < KC / 4; ++ks) {
                                    av[ks] = Ab[ks * 4];
#pragma unroll
                                    for (int j = 0; j < NT; ++j) bv[ks][j] = Bb[ks * 4 * LDB + j * 16];
                                }
#pragma unroll
                                for (int ks = 0; ks < KC / 4; ++ks)
#pragma unroll
                                    for (int j = 0; j < NT; ++j) acc[j] = mfma16(av[ks], bv[ks][j], acc[j]);
                            }
                            if (ch + 1 < nch) ut_store(ra[d ^ 1], rb[d ^ 1], rsc[d ^ 1], rsh[d ^ 1], rok[d ^ 1], buf ^ 1);
                            __syncthreads();
                        }
                    }
                }
            }
        } else {
            if (nch > 0) {
                stage_load(0);
                stage_store(0);
            }
            __syncthreads();
            for (int ch = 0; ch < nch; ++ch) {
                const int buf = ch & 1;
                if (ch + 1 < nch) stage_load(ch + 1);  // global loads of the next chunk fly under this chunk's MFMAs
                // all LDS operand reads of the chunk first, then the MFMAs back to back: with one wave per SIMD a
                // read -> wait -> MFMA chain per k-step leaves the matrix pipe idle for the LDS latency 8 times per chunk
                {
                    const float* Ab = &As[buf][(wave * 16 + r16) * LDA + kq];
                    const float* Bb = &Bs[buf][kq * LDB + r16];
                    float av[KC / 4], bv[KC / 4][NT];
#pragma unroll
                    for (int ks = 0; ks < KC / 4; ++ks) {
                        av[ks] = Ab[ks * 4];
#pragma unroll
                        for (int j = 0; j < NT; ++j) bv[ks][j] = Bb[ks * 4 * LDB + j * 16];
                    }
#pragma unroll
                    for (int ks = 0; ks < KC / 4; ++ks)
#pragma unroll
                        for (int j = 0; j < NT; ++j) acc[j] = mfma16(av[ks], bv[ks][j], acc[j]);
                }
                if (ch + 1 < nch) stage_store(buf ^ 1);
                __syncthreads();
            }
        }

        // ---- epilogue
        const unsigned row0 = tile * TM + wave * 16 + kq * 4;
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int col = n0 + j * 16 + r16;
                if (col < g.Cn) {
                    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const unsigned pm = row0 + r;
                        if (pm < rows) {
                            const size_t o = (size_t)pm * g.Cn + col;
                            float v = acc[j][r] + bv;
                            if (res) v += res[o];
                            y[o] = v;
                            if (partial) {  // per-channel sums of the OUTPUT: the next layer's BatchNorm statistics
                                s1[j] += (double)v;
                                s2[j] += (double)v * (double)v;
                            }
                        }
                    }
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int col = n0 + j * 16 + r16;
                if (col >= g.Cs) continue;
                const float sc = scale ? scale[col] : 1.f, shf = scale ? shift[col] : 0.f;
                const float mu = mean ? mean[col] : 0.f, is = mean ? invstd[col] : 0.f;
                if (g.up == 2) {
                    if (row0 < rows) {
                        const unsigned parent = row0 >> 2;
                        float val = (acc[j][0] + acc[j][1]) + (acc[j][2] + acc[j][3]);
                        const size_t o = (size_t)parent * g.Cs + col;
                        float xv = 0.f;
                        if (relu || mean) xv = xin[o];
                        if (relu) {
                            const float v = scale ? fmaf(xv, sc, shf) : xv;
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
                            size_t o;
                            if (g.stride == 1) {  // rows enumerate the input positions in memory order
                                o = (size_t)rr * g.Cs + col;
                            } else {
                                int pn, piy, pix;
                                dgrad_row_to_pos(g, rd, rr, py, px, pn, piy, pix);
                                o = ((size_t)((unsigned)pn * g.Hs + piy) * g.Ws + pix) * g.Cs + col;
                            }
                            float val = acc[j][r];
                            float xv = 0.f;
                            if (relu || mean) xv = xin[o];
                            if (relu) {
                                const float v = scale ? fmaf(xv, sc, shf) : xv;
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
    }

    if ((MODE == 1 && mean) || (MODE == 0 && partial)) {
        // fixed-order reduction of the per-channel sums (BatchNorm-backward sums, resp. output statistics): lane groups (shuffle), waves (LDS), one partial per block
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            double a = s1[j], b = s2[j];
            a += __shfl_xor(a, 16, 64);
            a += __shfl_xor(a, 32, 64);
            b += __shfl_xor(b, 16, 64);
            b += __shfl_xor(b, 32, 64);
            if (kq == 0) {
                red[(wave * 2 + 0) * BN + j * 16 + r16] = a;
                red[(wave * 2 + 1) * BN + j * 16 + r16] = b;
            }
        }
        __syncthreads();
        if (tid < 2 * BN) {
            const int which = tid / BN, cc = tid % BN;
            const double t = (red[(0 * 2 + which) * BN + cc] + red[(1 * 2 + which) * BN + cc]) +
                             (red[(2 * 2 + which) * BN + cc] + red[(3 * 2 + which) * BN + cc]);
            const unsigned p = bz * gx + bx;
            const unsigned Ptot = gx * gz;
            bn_stat_out(partial, which, CsPad, n0 + cc, Ptot, p, t);  // [2][CsPad][P], or the statistic slots
        }
    }
}

// TAB (forward only): the BatchNorm affine of the input is kept in LDS -- computed by every block from the statistic slots (fold) or
// copied from the global arrays -- and the body reads it there; TAB = false is the plain form (global arrays; any channel count).
template <int MODE, int NT, bool VEC, bool UT, bool TAB = false>
__global__ __launch_bounds__(256) void conv_gemm_kernel(Geom g, const float* __restrict__ S, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, int relu,
                                                        const float* __restrict__ Bmat, const float* __restrict__ bias,
                                                        const float* __restrict__ res, float* __restrict__ y,
                                                        const float* __restrict__ xin, const float* __restrict__ mean,
                                                        const float* __restrict__ invstd, float* __restrict__ gv,
                                                        double* __restrict__ partial, int CsPad, BnFold fold) {
    __shared__ __align__(16) char smem[GemmSmem<NT>::bytes];
    if constexpr (TAB) {
        __shared__ __align__(16) float bn_tab[2][BN_TAB];
        bn_tab_fill(fold, scale, shift, g.Cs, bn_tab[0], bn_tab[1], (blockIdx.x | blockIdx.y | blockIdx.z) == 0);
        conv_gemm_body<MODE, NT, VEC, UT>(smem, g, S, bn_tab[0], bn_tab[1], relu, Bmat, bias, res, y, xin, mean, invstd, gv, partial, CsPad,
                                      blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.z);
    } else {
        conv_gemm_body<MODE, NT, VEC, UT>(smem, g, S, scale, shift, relu, Bmat, bias, res, y, xin, mean, invstd, gv, partial, CsPad,
                                      blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.z);
    }
}

template <int MODE, bool VEC, bool UT>
static void launch_gemm_v(int NT, dim3 grid, hipStream_t st, Geom g, const float* S, const float* scale, const float* shift,
                          int relu, const float* Bmat, const float* bias, const float* res, float* y, const float* xin,
                          const float* mean, const float* invstd, float* gv, double* partial, int CsPad, const BnFold& fold) {
    // forward launches with a BatchNorm in front (and not more channels than the table holds) take the LDS-table form
    const bool tab = MODE == 0 && (scale != nullptr || fold.slots != nullptr) && g.Cs <= BN_TAB;
#define OTVAE_CG(N_)                                                                                                        \
    do {                                                                                                                    \
        if constexpr (MODE == 0) {                                                                                          \
            if (tab) {                                                                                                      \
                conv_gemm_kernel<MODE, N_, VEC, UT, true><<<grid, 256, 0, st>>>(g, S, scale, shift, relu, Bmat, bias, res, y, xin, mean, \
                                                                            invstd, gv, partial, CsPad, fold);              \
                break;                                                                                                      \
            }                                                                                                               \
        }                                                                                                                   \
        conv_gemm_kernel<MODE, N_, VEC, UT, false><<<grid, 256, 0, st>>>(g, S, scale, shift, relu, Bmat, bias, res, y, xin, mean, invstd, \
                                                                     gv, partial, CsPad, fold);                             \
    } while (0)
    switch (NT) {
        case 1: OTVAE_CG(1); break;
        case 2: OTVAE_CG(2); break;
        case 3: OTVAE_CG(3); break;
        default: OTVAE_CG(4); break;
    }
#undef OTVAE_CG
}

static inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

template <int MODE>
static void launch_gemm(int NT, dim3 grid, hipStream_t st, Geom g, const float* S, const float* scale, const float* shift,
                        int relu, const float* Bmat, const float* bias, const float* res, float* y, const float* xin,
                        const float* mean, const float* invstd, float* gv, double* partial, int CsPad, const BnFold& fold = BnFold{}) {
    const bool vec = (g.Cs % 4 == 0) && (g.Cn % 4 == 0) && aligned16(S) && aligned16(Bmat) &&
                     (scale == nullptr || (aligned16(scale) && aligned16(shift)));
    const int CK = MODE == 0 ? g.Cs : g.Cn;
    if (vec && CK % KC == 0)  // a K-chunk lies inside one tap: uniform-tap pipeline
        launch_gemm_v<MODE, true, true>(NT, grid, st, g, S, scale, shift, relu, Bmat, bias, res, y, xin, mean, invstd, gv, partial, CsPad, fold);
    else if (vec)
        launch_gemm_v<MODE, true, false>(NT, grid, st, g, S, scale, shift, relu, Bmat, bias, res, y, xin, mean, invstd, gv, partial, CsPad, fold);
    else
        launch_gemm_v<MODE, false, false>(NT, grid, st, g, S, scale, shift, relu, Bmat, bias, res, y, xin, mean, invstd, gv, partial, CsPad, fold);
}

static void fwd_grid(const Geom& g, int& NT, dim3& grid, int& CnPad) {
    const unsigned M = (unsigned)g.N * g.Ho * g.Wo;
    NT = pick_nt(g.Cn, cdiv(M, TM));
    const int ny = cdiv(cdiv(g.Cn, 16), NT);
    grid = dim3(imin(cdiv(M, TM), 2048), ny, 1);  // <= 2048 blocks: each may write one statistics partial
    CnPad = ny * 16 * NT;
}

extern "C" int otvae_conv_fwd_stats_ws(const otvae_conv_geom* gg, int* P, int* CnPad) {
    int rc = check_geom(gg, "otvae_conv_fwd_stats_ws");
    if (rc) return rc;
    Geom g = to_geom(gg);
    int NT, cp;
    dim3 grid;
    fwd_grid(g, NT, grid, cp);
    if (conv_small_ok(g)) grid.x = imin(grid.x, 1024);
    TilePlan pl;
    dim3 tg;
    size_t sm;
    if (!conv_small_ok(g) && conv_tile_plan(g, 0, pl, tg, sm)) {
        grid.x = tg.x * tg.z;
        cp = pl.cpad;
    }
    if (P) *P = grid.x;
    if (CnPad) *CnPad = cp;
    return OTVAE_OK;
}

static inline bool g_cs_le_tab(const otvae_conv_geom* g) { return g->Cs <= BN_TAB; }

// the forward launch with its two round-4 extras: the BatchNorm of the input folded in (fold.slots != NULL), and the output's statistics
// into slots instead of partials (stat_slots != NULL)
static int conv_fwd_ex(const otvae_conv_geom* gg, const float* x, const float* scale, const float* shift, int relu, const float* wT,
                       const float* bias, const float* residual, float* y, double* stat_dst, const BnFold& fold, void* stream) {
    int rc = check_geom(gg, "otvae_conv_fwd");
    if (rc) return rc;
    OTVAE_REQUIRE(x && wT && y, "otvae_conv_fwd: NULL tensor");
    OTVAE_REQUIRE((scale == nullptr) == (shift == nullptr), "otvae_conv_fwd: scale and shift must be given together");
    OTVAE_REQUIRE(fold.slots == nullptr || g_cs_le_tab(gg), "otvae_conv_fwd: a folded BatchNorm takes at most %d channels", BN_TAB);
    Geom g = to_geom(gg);
    int NT, CnPad;
    dim3 grid;
    fwd_grid(g, NT, grid, CnPad);
    if (conv_small_ok(g)) {
        conv_small_fwd(g, imin(grid.x, 1024), x, scale, shift, relu, wT, bias, residual, y, stat_dst, CnPad, fold,
                       (hipStream_t)stream);
        OTVAE_CHECK_LAUNCH("otvae_conv_fwd(small)");
        return OTVAE_OK;
    }
    {
        TilePlan pl;
        dim3 tg;
        size_t sm;
        if (conv_tile_plan(g, 0, pl, tg, sm)) {
            OTVAE_REQUIRE(aligned16(x) && aligned16(wT) && aligned16(y) && (!bias || aligned16(bias)) &&
                              (!residual || aligned16(residual)) && (!scale || (aligned16(scale) && aligned16(shift))),
                          "otvae_conv_fwd: tensors of a layer with channel counts %% 4 == 0 must be 16-byte aligned");
            rc = conv_tile_fwd(pl, tg, sm, (hipStream_t)stream, x, scale, shift, relu, wT, bias, residual, y, stat_dst, fold);
            OTVAE_REQUIRE(rc == 0, "otvae_conv_fwd: no image-tile kernel for nt=%d rbw=%d", pl.nt, pl.rbw);
            OTVAE_CHECK_LAUNCH("otvae_conv_fwd(tile)");
            return OTVAE_OK;
        }
    }
    launch_gemm<0>(NT, grid, (hipStream_t)stream, g, x, scale, shift, relu, wT, bias, residual, y, nullptr, nullptr, nullptr,
                   nullptr, stat_dst, CnPad, fold);
    OTVAE_CHECK_LAUNCH("otvae_conv_fwd");
    return OTVAE_OK;
}

extern "C" int otvae_conv_fwd(const otvae_conv_geom* gg, const float* x, const float* scale, const float* shift, int relu,
                              const float* wT, const float* bias, const float* residual, float* y, double* stat_partial,
                              void* stream) {
    return conv_fwd_ex(gg, x, scale, shift, relu, wT, bias, residual, y, stat_partial, BnFold{}, stream);
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

// all conv weights of a model in ONE launch: table[l] = {src offset, dst offset, T, Cs, Cn} (element offsets into the
// flat parameter buffer / the flat dgrad-layout buffer), blockIdx.y = layer
__global__ void weight_transpose_batched_kernel(const float* __restrict__ src_base, float* __restrict__ dst_base,
                                                const int64_t* __restrict__ table) {
    const int64_t* d = table + (size_t)blockIdx.y * 5;
    const float* wT = src_base + d[0];
    float* wD = dst_base + d[1];
    const int T = (int)d[2], Cs = (int)d[3], Cn = (int)d[4];
    const size_t total = (size_t)T * Cs * Cn;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = i % Cs;
        const size_t r = i / Cs;
        const int n = r % Cn;
        const int t = r / Cn;
        wD[i] = wT[((size_t)t * Cs + c) * Cn + n];
    }
}

extern "C" int otvae_weight_transpose_batched(const float* src_base, float* dst_base, const int64_t* table, int n_layers,
                                              int64_t max_elems, void* stream) {
    OTVAE_REQUIRE(src_base && dst_base && table && n_layers > 0 && max_elems > 0, "otvae_weight_transpose_batched: bad argument");
    dim3 grid(imin(cdiv(max_elems, 256), 256), n_layers);
    weight_transpose_batched_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(src_base, dst_base, table);
    OTVAE_CHECK_LAUNCH("otvae_weight_transpose_batched");
    return OTVAE_OK;
}

// ------------------------------------------------------------------------------------------------ data gradient
static void dgrad_grid(const Geom& g, int& NT, dim3& grid, int& CsPad) {
    const int Hu = g.Hs * g.up, Wu = g.Ws * g.up;
    const unsigned rows = (g.stride == 2) ? (unsigned)g.N * (g.Hs >> 1) * (g.Ws >> 1) : (unsigned)g.N * Hu * Wu;
    const int nz = g.stride == 2 ? 4 : 1;
    NT = pick_nt(g.Cs, (long)cdiv(rows, TM) * nz);
    const int ny = cdiv(cdiv(g.Cs, 16), NT);
    // bounded number of blocks (each writes one BatchNorm partial): <= 1024 over x*z
    grid = dim3(imax(1, imin(cdiv(rows, TM), 1024 / nz)), ny, nz);
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
    TilePlan pl;
    dim3 tg;
    size_t sm;
    if (!conv_small_ok(g) && conv_tile_plan(g, 1, pl, tg, sm)) {
        grid = tg;
        cp = pl.cpad;
    }
    if (P) *P = grid.x * grid.z;
    if (CsPad) *CsPad = cp;
    return OTVAE_OK;
}

// bn_partial: the [2][CsPad][P] partials, or tagged statistic slots (bn_tag_slots)
static int conv_bwd_data_ex(const otvae_conv_geom* gg, const float* gy, const float* wD, const float* x, const float* scale,
                            const float* shift, int relu, const float* mean, const float* invstd, float* gv, double* bn_partial,
                            void* stream);

extern "C" int otvae_conv_bwd_data(const otvae_conv_geom* gg, const float* gy, const float* wD, const float* x,
                                   const float* scale, const float* shift, int relu, const float* mean, const float* invstd,
                                   float* gv, double* bn_partial, void* stream) {
    return conv_bwd_data_ex(gg, gy, wD, x, scale, shift, relu, mean, invstd, gv, bn_partial, stream);
}

static int conv_bwd_data_ex(const otvae_conv_geom* gg, const float* gy, const float* wD, const float* x, const float* scale,
                            const float* shift, int relu, const float* mean, const float* invstd, float* gv, double* bn_partial,
                            void* stream) {
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
    if (conv_small_ok(g)) {  // same number of BatchNorm partials (grid.x * grid.z) as the workspace query promised
        conv_small_dgrad(g, grid.x * grid.z, gy, wD, x, scale, shift, relu, mean, invstd, gv, bn_partial, CsPad,
                         (hipStream_t)stream);
        OTVAE_CHECK_LAUNCH("otvae_conv_bwd_data(small)");
        return OTVAE_OK;
    }
    {
        TilePlan pl;
        dim3 tg;
        size_t sm;
        if (conv_tile_plan(g, 1, pl, tg, sm)) {
            OTVAE_REQUIRE(aligned16(gy) && aligned16(wD) && aligned16(gv) && (!x || aligned16(x)) &&
                              (!scale || (aligned16(scale) && aligned16(shift))) && (!mean || (aligned16(mean) && aligned16(invstd))),
                          "otvae_conv_bwd_data: tensors of a layer with channel counts %% 4 == 0 must be 16-byte aligned");
            rc = conv_tile_dgrad(pl, tg, sm, (hipStream_t)stream, gy, wD, x, scale, shift, relu, mean, invstd, gv, bn_partial);
            OTVAE_REQUIRE(rc == 0, "otvae_conv_bwd_data: no image-tile kernel for nt=%d rbw=%d", pl.nt, pl.rbw);
            OTVAE_CHECK_LAUNCH("otvae_conv_bwd_data(tile)");
            return OTVAE_OK;
        }
    }
    launch_gemm<1>(NT, grid, (hipStream_t)stream, g, gy, scale, shift, relu, wD, nullptr, nullptr, nullptr, x, mean, invstd, gv,
                   bn_partial, CsPad);
    OTVAE_CHECK_LAUNCH("otvae_conv_bwd_data");
    return OTVAE_OK;
}

// ------------------------------------------------------------------------------------------------ weight gradient
// Block = 64 k-rows x (16*NT) n-cols, reduction over one pixel chunk in sub-chunks of 32 pixels staged in LDS
// (At[pixel][k], Gt[pixel][n], double-buffered).  Row K (after the last tap*channel row) is the bias row: A = 1.
#define PC 32
template <int NT>
struct WgradSmem {
    static constexpr int BN = 16 * NT;
    static constexpr int LDA = 64 + 16;
    static constexpr int LDB = BN + ((BN % 32 == 0) ? 16 : 0);
    static constexpr size_t at_off = 0;
    static constexpr size_t gt_off = at_off + sizeof(float) * 2 * PC * LDA;
    static constexpr size_t live_off = gt_off + sizeof(float) * 2 * PC * LDB;
    static constexpr size_t bytes = live_off + 16;
};

template <int NT, bool VEC>
__device__ __forceinline__ void conv_wgrad_body(char* __restrict__ smem, const Geom& g, const float* __restrict__ x,
                                                const float* __restrict__ scale, const float* __restrict__ shift, int relu,
                                                const float* __restrict__ gy, float* __restrict__ partial, int Kp,
                                                int has_bias, int skip_dead, unsigned chunk, int bx, int by, int bz) {
    using SM = WgradSmem<NT>;
    constexpr int BN = SM::BN;
    constexpr int LDA = SM::LDA;
    constexpr int LDB = SM::LDB;
    constexpr int NB_ELEMS = PC * BN / 256;
    float(*At)[PC * LDA] = reinterpret_cast<float(*)[PC * LDA]>(smem + SM::at_off);
    float(*Gt)[PC * LDB] = reinterpret_cast<float(*)[PC * LDB]>(smem + SM::gt_off);
    int* s_live = reinterpret_cast<int*>(smem + SM::live_off);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const int kb = bx, n0 = by * BN;
    const unsigned pc = bz;
    const unsigned M = (unsigned)g.N * g.Ho * g.Wo;
    const unsigned mbeg = pc * chunk, mend = min(M, mbeg + chunk);
    const RowDiv wrd = make_rowdiv(g, M + 64);
    const int K = g.KH * g.KW * g.Cs;
    const int Hu = g.Hs * g.up, Wu = g.Ws * g.up;
    const int ush = g.up - 1;

    // ---- this thread's k row, fixed for the whole block: staging role = (k-local = tid & 63, pixels (tid >> 6) + 4 i)
    const int a_kl = tid & 63, a_p0 = tid >> 6;
    const int k = kb * 64 + a_kl;
    int kind = 0, kdy = 0, kdx = 0, kc = 0;  // kind: 0 dead, 1 tap row, 2 bias row
    float a_sc = 1.f, a_sh = 0.f;
    if (k < K) {
        const int t = k / g.Cs;
        kc = k - t * g.Cs;
        const int kh = t / g.KW;
        kdy = kh - g.pad;
        kdx = (t - kh * g.KW) - g.pad;
        const bool ytouch = (kdy + (g.Ho - 1) * g.stride >= 0) && (kdy < Hu);
        const bool xtouch = (kdx + (g.Wo - 1) * g.stride >= 0) && (kdx < Wu);
        kind = (ytouch && xtouch) ? 1 : 0;
        if (kind && scale) {
            a_sc = scale[kc];
            a_sh = shift[kc];
        }
    } else if (has_bias && k == K) {
        kind = 2;
    }
    // VEC role (channel counts multiples of 4): k4 = tid & 15 (4 consecutive k of ONE tap), pixels (tid >> 4) + 16 i
    const int v_k4 = tid & 15, v_p0 = tid >> 4;
    int vkind = 0, vdy = 0, vdx = 0, vc = 0;
    float4 v_sc4 = make_float4(1.f, 1.f, 1.f, 1.f), v_sh4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (VEC) {
        const int k0 = kb * 64 + v_k4 * 4;
        if (k0 < K) {
            const int t = k0 / g.Cs;
            vc = k0 - t * g.Cs;
            const int kh = t / g.KW;
            vdy = kh - g.pad;
            vdx = (t - kh * g.KW) - g.pad;
            const bool ytouch = (vdy + (g.Ho - 1) * g.stride >= 0) && (vdy < Hu);
            const bool xtouch = (vdx + (g.Wo - 1) * g.stride >= 0) && (vdx < Wu);
            vkind = (ytouch && xtouch) ? 1 : 0;
            if (vkind && scale) {
                v_sc4 = *reinterpret_cast<const float4*>(scale + vc);
                v_sh4 = *reinterpret_cast<const float4*>(shift + vc);
            }
        } else if (has_bias && k0 == K) {
            vkind = 2;  // K % 4 == 0: the bias row opens a group {1, 0, 0, 0}
        }
    }
    // which 16-row MFMA tiles hold a live row?  (every staging wave sees all 64 k-rows in its lanes)
    {
        if constexpr (VEC) {
            const unsigned long long bal = __ballot(vkind != 0);  // lane l <-> k4 = l & 15
            if (tid < 4) s_live[tid] = ((bal >> (4 * tid)) & 0xfull) ? 1 : 0;
        } else {
            const unsigned long long bal = __ballot(kind != 0);
            if (tid < 4) s_live[tid] = ((bal >> (16 * tid)) & 0xffffull) ? 1 : 0;
        }
        __syncthreads();
    }
    const bool live = s_live[wave] != 0;
    const bool any_live = (s_live[0] | s_live[1] | s_live[2] | s_live[3]) != 0;

    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const unsigned span = mend > mbeg ? mend - mbeg : 0;
    const int nsub = any_live ? (int)((span + PC - 1) / PC) : 0;
    constexpr int NGV = (8 * BN + 255) / 256;
    float areg[8], greg[NB_ELEMS];
    float4 areg4[2], greg4[NGV];
    int a_ok = 0;

    auto stage_load = [&](int sub) {
        const unsigned mb = mbeg + (unsigned)sub * PC;
        if constexpr (VEC) {
            unsigned m = mb + v_p0;
            unsigned t = rdiv(m, g.Wo, wrd.iWo, wrd.small);
            int ox = m - t * g.Wo;
            int n = rdiv(t, g.Ho, wrd.iHo, wrd.small);
            int oy = t - (unsigned)n * g.Ho;
            a_ok = 0;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                areg4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (m < mend) {
                    if (vkind == 1) {
                        const int iy = oy * g.stride + vdy, ix = ox * g.stride + vdx;
                        if ((unsigned)iy < (unsigned)Hu && (unsigned)ix < (unsigned)Wu) {
                            areg4[i] = *reinterpret_cast<const float4*>(
                                x + ((size_t)((unsigned)n * g.Hs + (iy >> ush)) * g.Ws + (ix >> ush)) * g.Cs + vc);
                            a_ok |= 1 << i;
                        }
                    } else if (vkind == 2) {
                        areg4[i].x = 1.f;
                    }
                }
                m += 16;
                ox += 16;
                while (ox >= g.Wo) {
                    ox -= g.Wo;
                    if (++oy >= g.Ho) {
                        oy = 0;
                        ++n;
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < NGV; ++j) {
                const int e = tid + 256 * j;
                const int p = e / (BN / 4), c4 = e - p * (BN / 4);
                const unsigned mm = mb + p;
                greg4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < 8 * BN && mm < mend && n0 + c4 * 4 < g.Cn)
                    greg4[j] = *reinterpret_cast<const float4*>(gy + (size_t)mm * g.Cn + n0 + c4 * 4);
            }
            return;
        }
        // A: pixels mb + a_p0 + 4 i: decode the first, then step by 4 pixels
        unsigned m = mb + a_p0;
        unsigned t = rdiv(m, g.Wo, wrd.iWo, wrd.small);
        int ox = m - t * g.Wo;
        int n = rdiv(t, g.Ho, wrd.iHo, wrd.small);
        int oy = t - (unsigned)n * g.Ho;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float v = __int_as_float(PAD_MARK);
            if (m < mend) {
                if (kind == 1) {
                    const int iy = oy * g.stride + kdy, ix = ox * g.stride + kdx;
                    if (iy >= 0 && iy < Hu && ix >= 0 && ix < Wu)
                        v = x[((size_t)((unsigned)n * g.Hs + (iy >> ush)) * g.Ws + (ix >> ush)) * g.Cs + kc];
                } else if (kind == 2) {
                    v = __int_as_float(ONE_MARK);
                }
            }
            areg[i] = v;
            m += 4;
            ox += 4;
            while (ox >= g.Wo) {
                ox -= g.Wo;
                if (++oy >= g.Ho) {
                    oy = 0;
                    ++n;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NB_ELEMS; ++j) {
            const int e = tid + 256 * j;
            const int p = e / BN, col = e - p * BN;
            const unsigned mm = mb + p;
            greg[j] = (mm < mend && n0 + col < g.Cn) ? gy[(size_t)mm * g.Cn + n0 + col] : 0.f;
        }
    };
    auto stage_store = [&](int buf) {
        if constexpr (VEC) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float4 v = areg4[i];
                if ((a_ok >> i) & 1) {
                    if (scale) {
                        v.x = fmaf(v.x, v_sc4.x, v_sh4.x);
                        v.y = fmaf(v.y, v_sc4.y, v_sh4.y);
                        v.z = fmaf(v.z, v_sc4.z, v_sh4.z);
                        v.w = fmaf(v.w, v_sc4.w, v_sh4.w);
                    }
                    if (relu) {
                        v.x = fmaxf(v.x, 0.f);
                        v.y = fmaxf(v.y, 0.f);
                        v.z = fmaxf(v.z, 0.f);
                        v.w = fmaxf(v.w, 0.f);
                    }
                }
                *reinterpret_cast<float4*>(&At[buf][(v_p0 + 16 * i) * LDA + v_k4 * 4]) = v;
            }
#pragma unroll
            for (int j = 0; j < NGV; ++j) {
                const int e = tid + 256 * j;
                if (e < 8 * BN) {
                    const int p = e / (BN / 4), c4 = e - p * (BN / 4);
                    *reinterpret_cast<float4*>(&Gt[buf][p * LDB + c4 * 4]) = greg4[j];
                }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float v = areg[i];
            const int bits = __float_as_int(v);
            if (bits == PAD_MARK) {
                v = 0.f;
            } else if (bits == ONE_MARK) {
                v = 1.f;
            } else {
                if (scale) v = fmaf(v, a_sc, a_sh);
                if (relu) v = fmaxf(v, 0.f);
            }
            At[buf][(a_p0 + 4 * i) * LDA + a_kl] = v;
        }
#pragma unroll
        for (int j = 0; j < NB_ELEMS; ++j) {
            const int e = tid + 256 * j;
            const int p = e / BN, col = e - p * BN;
            Gt[buf][p * LDB + col] = greg[j];
        }
    };

    if (nsub > 0) {
        stage_load(0);
        stage_store(0);
    }
    __syncthreads();
    for (int sub = 0; sub < nsub; ++sub) {
        const int buf = sub & 1;
        if (sub + 1 < nsub) stage_load(sub + 1);
        if (live) {
            const float* Ab = &At[buf][kq * LDA + wave * 16 + r16];
            const float* Gb = &Gt[buf][kq * LDB + r16];
            float av[PC / 4], gv4[PC / 4][NT];
#pragma unroll
            for (int ps = 0; ps < PC / 4; ++ps) {
                av[ps] = Ab[ps * 4 * LDA];
#pragma unroll
                for (int j = 0; j < NT; ++j) gv4[ps][j] = Gb[ps * 4 * LDB + j * 16];
            }
#pragma unroll
            for (int ps = 0; ps < PC / 4; ++ps)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[j] = mfma16(av[ps], gv4[ps][j], acc[j]);
        }
        if (sub + 1 < nsub) stage_store(buf ^ 1);
        __syncthreads();
    }
    // partial[pc][k][n].  A 16-row tile without a live row holds only rows of taps that never touch the image (or rows
    // past Kp): with skip_dead the reduction knows them (otvae_conv_dead_taps) and does not read them
    if (skip_dead && !live) return;
    float* out = partial + (size_t)pc * Kp * g.Cn;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int col = n0 + j * 16 + r16;
        if (col >= g.Cn) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kk = kb * 64 + wave * 16 + kq * 4 + r;
            if (kk < Kp) out[(size_t)kk * g.Cn + col] = acc[j][r];
        }
    }
}

template <int NT, bool VEC>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(Geom g, const float* __restrict__ x, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, int relu,
                                                         const float* __restrict__ gy, float* __restrict__ partial, int Kp,
                                                         int has_bias, int skip_dead, unsigned chunk) {
    __shared__ __align__(16) char smem[WgradSmem<NT>::bytes];
    conv_wgrad_body<NT, VEC>(smem, g, x, scale, shift, relu, gy, partial, Kp, has_bias, skip_dead, chunk, blockIdx.x,
                             blockIdx.y, blockIdx.z);
}

// out[e] = sum_p partial[p][e] in a fixed order.
//   few partials : 4 interleaved p-lanes per element, then p-lane 0..3 through LDS
//   many partials: one wave per element, lanes stride over p, shuffle tree
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int P, int K, int Kp, int Cn,
                                                           float* __restrict__ gw, float* __restrict__ gb) {
    __shared__ double red[4][64];  // fp32 partials, fp64 sum (see wgrad_reduce_batched_kernel)
    const size_t total = (size_t)Kp * Cn;
    if (P >= 32) {
        const int lane = threadIdx.x & 63;
        for (size_t e = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); e < total; e += (size_t)gridDim.x * 4) {
            double s = 0.0;
            for (int p = lane; p < P; p += 64) s += (double)partial[(size_t)p * total + e];
            s = wave_sum(s);
            if (lane == 0) {
                const int k = e / Cn;
                if (k < K)
                    gw[e] = (float)s;
                else if (gb)
                    gb[e - (size_t)K * Cn] = (float)s;
            }
        }
        return;
    }
    const int el = threadIdx.x & 63, pg = threadIdx.x >> 6;
    for (size_t e0 = (size_t)blockIdx.x * 64; e0 < total; e0 += (size_t)gridDim.x * 64) {
        const size_t e = e0 + el;
        double s = 0.0;
        if (e < total)
            for (int p = pg; p < P; p += 4) s += (double)partial[(size_t)p * total + e];
        red[pg][el] = s;
        __syncthreads();
        if (pg == 0 && e < total) {
            const float t = (float)((red[0][el] + red[1][el]) + (red[2][el] + red[3][el]));
            const int k = e / Cn;
            if (k < K)
                gw[e] = t;
            else if (gb)
                gb[e - (size_t)K * Cn] = t;
        }
        __syncthreads();
    }
}

static void wgrad_plan(const Geom& g, int has_bias, int& NT, int& P, unsigned& chunk, int& nkb, int& nnb, int& Kp) {
    const unsigned M = (unsigned)g.N * g.Ho * g.Wo;
    Kp = g.KH * g.KW * g.Cs + (has_bias ? 1 : 0);
    nkb = cdiv(Kp, 64);
    {
        const int pix = imax(1, (int)(M / 128)), ws = imax(1, (int)((4u << 20) / ((unsigned)Kp * g.Cn)));
        NT = pick_nt(g.Cn, (long)nkb * imin(imin(pix, ws), 2048));
    }
    nnb = cdiv(cdiv(g.Cn, 16), NT);
    if (conv_small_wgrad_ok(g)) {
        conv_small_wgrad_plan(g, P, chunk);
        return;
    }
    // ~256 workgroups (1 per CU).  Measured on MI355X: 192..2048 are within 0.5 % of each other for the whole step -- the jobs run
    // beside the data-gradient chain on their own stream -- and the split-K partial traffic (written here, read by the batched
    // reduction) is proportional to the number of pixel chunks; pixel chunks of >= 128 pixels, workspace <= 16 MiB, P <= 2048
    const unsigned ws_cap = 4u << 20;
    static const int target_wgs = [] {
        const char* e = getenv("OTVAE_WGRAD_WGS");
        const int v = e ? atoi(e) : 0;
        return v >= 64 && v <= 8192 ? v : 256;
    }();
    int want = cdiv(target_wgs, nkb * nnb);
    int maxp_pix = imax(1, (int)(M / 128));
    int maxp_ws = imax(1, (int)(ws_cap / ((unsigned)Kp * g.Cn)));
    P = imax(1, imin(imin(want, maxp_pix), imin(maxp_ws, 2048)));
    chunk = ((M + P - 1) / P + PC - 1) / PC * PC;
    P = cdiv(M, chunk);
}

extern "C" int otvae_conv_bwd_weight_ws(const otvae_conv_geom* gg, int has_bias, int* P) {
    int rc = check_geom(gg, "otvae_conv_bwd_weight_ws");
    if (rc) return rc;
    Geom g = to_geom(gg);
    int NT, p, nkb, nnb, Kp;
    unsigned chunk;
    wgrad_plan(g, has_bias, NT, p, chunk, nkb, nnb, Kp);
    {
        WTilePlan wp;
        int nbk;
        size_t sm;
        if (conv_wtile_plan(g, has_bias, wp, nbk, sm)) p = conv_wtile_nparts(nbk);
    }
    if (P) *P = p;
    return OTVAE_OK;
}

extern "C" int otvae_conv_bwd_weight(const otvae_conv_geom* gg, const float* x, const float* scale, const float* shift,
                                     int relu, const float* gy, int has_bias, float* partial, float* gw, float* gb,
                                     int defer_reduce, void* stream) {
    int rc = check_geom(gg, "otvae_conv_bwd_weight");
    if (rc) return rc;
    OTVAE_REQUIRE(x && gy && partial && gw, "otvae_conv_bwd_weight: NULL tensor");
    OTVAE_REQUIRE(!has_bias || gb, "otvae_conv_bwd_weight: gb missing");
    OTVAE_REQUIRE((scale == nullptr) == (shift == nullptr), "otvae_conv_bwd_weight: scale/shift must come together");
    Geom g = to_geom(gg);
    int NT, P, nkb, nnb, Kp;
    unsigned chunk;
    wgrad_plan(g, has_bias, NT, P, chunk, nkb, nnb, Kp);
    hipStream_t st = (hipStream_t)stream;
    const size_t total = (size_t)Kp * g.Cn;
    {
        WTilePlan wp;
        int nbk;
        size_t sm;
        if (conv_wtile_plan(g, has_bias, wp, nbk, sm)) {
            OTVAE_REQUIRE(!wp.vec4 || (aligned16(x) && (!scale || (aligned16(scale) && aligned16(shift)))),
                          "otvae_conv_bwd_weight: x / scale / shift of a layer with Cs %% 4 == 0 must be 16-byte aligned");
            conv_wtile(wp, nbk, sm, st, x, scale, shift, relu, gy, partial);
            OTVAE_CHECK_LAUNCH("otvae_conv_bwd_weight(tile)");
            if (defer_reduce) return OTVAE_OK;
            P = conv_wtile_nparts(nbk);
            wgrad_reduce_kernel<<<imin(cdiv(total, P >= 32 ? 4 : 64), 2048), 256, 0, st>>>(partial, P, Kp - (has_bias ? 1 : 0),
                                                                                         Kp, g.Cn, gw, gb);
            OTVAE_CHECK_LAUNCH("otvae_conv_bwd_weight(reduce)");
            return OTVAE_OK;
        }
    }
    if (conv_small_wgrad_ok(g)) {
        conv_small_wgrad(g, x, scale, shift, relu, gy, has_bias, partial, st);
        OTVAE_CHECK_LAUNCH("otvae_conv_bwd_weight(small)");
        if (defer_reduce) return OTVAE_OK;
        wgrad_reduce_kernel<<<imin(cdiv(total, P >= 32 ? 4 : 64), 2048), 256, 0, st>>>(partial, P, Kp - (has_bias ? 1 : 0), Kp,
                                                                                     g.Cn, gw, gb);
        OTVAE_CHECK_LAUNCH("otvae_conv_bwd_weight(reduce)");
        return OTVAE_OK;
    }
    dim3 grid(nkb, nnb, P);
    const bool vec = (g.Cs % 4 == 0) && (g.Cn % 4 == 0) && aligned16(x) && aligned16(gy) &&
                     (scale == nullptr || (aligned16(scale) && aligned16(shift)));
#define OTVAE_WG(N_, V_) \
    conv_wgrad_kernel<N_, V_><<<grid, 256, 0, st>>>(g, x, scale, shift, relu, gy, partial, Kp, has_bias, \
                                                    defer_reduce == OTVAE_DEFER_SPARSE, chunk)
    if (vec) {
        switch (NT) {
            case 1: OTVAE_WG(1, true); break;
            case 2: OTVAE_WG(2, true); break;
            case 3: OTVAE_WG(3, true); break;
            default: OTVAE_WG(4, true); break;
        }
    } else {
        switch (NT) {
            case 1: OTVAE_WG(1, false); break;
            case 2: OTVAE_WG(2, false); break;
            case 3: OTVAE_WG(3, false); break;
            default: OTVAE_WG(4, false); break;
        }
    }
#undef OTVAE_WG
    OTVAE_CHECK_LAUNCH("otvae_conv_bwd_weight");
    if (defer_reduce) return OTVAE_OK;
    wgrad_reduce_kernel<<<imin(cdiv(total, P >= 32 ? 4 : 64), 2048), 256, 0, st>>>(partial, P, Kp - (has_bias ? 1 : 0), Kp, g.Cn, gw, gb);
    OTVAE_CHECK_LAUNCH("otvae_conv_bwd_weight(reduce)");
    return OTVAE_OK;
}

// ------------------------------------------------------------------------------------------------ batched reduce
// The partial -> gradient reductions of up to 32 layers in ONE launch (blockIdx.y = layer): the ~50 per-layer
// reductions of a backward pass are each a few microseconds of dependent-load latency; batching them removes their
// launch gaps and lets them share the chip.  Same fixed summation order as wgrad_reduce_kernel.
#define WRB_MAX 32
struct WrbDesc {
    const float* partial[WRB_MAX];
    float* gw[WRB_MAX];
    float* gb[WRB_MAX];
    int P[WRB_MAX], K[WRB_MAX], Kp[WRB_MAX], Cn[WRB_MAX];
    int Cs[WRB_MAX];         // rows per tap (0: no dead-tap information)
    unsigned dead[WRB_MAX];  // bit t: tap t never touches the image, its Cs rows are zero and may be unwritten
};

__global__ __launch_bounds__(256) void wgrad_reduce_batched_kernel(WrbDesc d) {
    // the partials are fp32 (the matrix cores' accumulators); their SUM is formed in fp64 -- up to 512 terms that partly cancel -- and
    // rounded once (round 4: held against the float64 truth, the fp32 sum of the partials alone cost up to 2.6x the reference's own error)
    __shared__ double red[4][64];
    const int l = blockIdx.y;
    const float* __restrict__ partial = d.partial[l];
    float* __restrict__ gw = d.gw[l];
    float* __restrict__ gb = d.gb[l];
    const int P = d.P[l], K = d.K[l], Cn = d.Cn[l];
    const size_t total = (size_t)d.Kp[l] * Cn;
    const int lane = threadIdx.x & 63;
    const unsigned dead = d.dead[l];
    const int rows_per_tap = d.Cs[l];
    // element e lies in a row of a dead tap: exactly zero, and the partials may hold garbage there
    auto is_dead = [&](size_t e) -> bool {
        if (dead == 0u) return false;
        const int k = (int)(e / (size_t)Cn);
        return k < K && ((dead >> (k / rows_per_tap)) & 1u);
    };
    if (total >= 4096 && P >= 16) {
        // large gradient: partial rows are far apart (total*4 bytes), so lanes run along e (coalesced 256-byte reads)
        // and the 4 waves split the partials; fixed order: p ascending within a wave, then waves 0..3 through LDS
        const int pg = threadIdx.x >> 6;
        for (size_t e0 = (size_t)blockIdx.x * 64; e0 < total; e0 += (size_t)gridDim.x * 64) {
            const size_t e = e0 + lane;
            double s = 0.0;
            if (e < total && !is_dead(e)) {
                int p = pg;
                for (; p + 12 < P; p += 16) {
                    const float a0 = partial[(size_t)p * total + e], a1 = partial[(size_t)(p + 4) * total + e];
                    const float a2 = partial[(size_t)(p + 8) * total + e], a3 = partial[(size_t)(p + 12) * total + e];
                    s += (double)a0;
                    s += (double)a1;
                    s += (double)a2;
                    s += (double)a3;
                }
                for (; p < P; p += 4) s += (double)partial[(size_t)p * total + e];
            }
            red[pg][lane] = s;
            __syncthreads();
            if (pg == 0 && e < total) {
                const float t = (float)((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]));
                const int k = e / Cn;
                if (k < K) gw[e] = t;
                else if (gb) gb[e - (size_t)K * Cn] = t;
            }
            __syncthreads();
        }
    } else if (P >= 16) {  // small gradient, many partials: one wave per element (neighbouring waves share the lines)
        for (size_t e = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); e < total; e += (size_t)gridDim.x * 4) {
            double s = 0.0;
            if (!is_dead(e))
                for (int p = lane; p < P; p += 64) s += (double)partial[(size_t)p * total + e];
            s = wave_sum(s);
            if (lane == 0) {
                const int k = e / Cn;
                if (k < K) gw[e] = (float)s;
                else if (gb) gb[e - (size_t)K * Cn] = (float)s;
            }
        }
    } else {  // one lane per element, serial over the few partials
        for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
            double s = 0.0;
            if (!is_dead(e))
                for (int p = 0; p < P; ++p) s += (double)partial[(size_t)p * total + e];
            const int k = e / Cn;
            if (k < K) gw[e] = (float)s;
            else if (gb) gb[e - (size_t)K * Cn] = (float)s;
        }
    }
}

// The taps of a layer that touch the image for no output position (the predicate of conv_wgrad_body): bit kh*KW + kw.
static unsigned dead_taps(const Geom& g) {
    const int Hu = g.Hs * g.up, Wu = g.Ws * g.up;
    unsigned m = 0;
    if (g.KH * g.KW > 32) return 0;
    for (int kh = 0; kh < g.KH; ++kh)
        for (int kw = 0; kw < g.KW; ++kw) {
            const int dy = kh - g.pad, dx = kw - g.pad;
            const bool ytouch = (dy + (g.Ho - 1) * g.stride >= 0) && (dy < Hu);
            const bool xtouch = (dx + (g.Wo - 1) * g.stride >= 0) && (dx < Wu);
            if (!(ytouch && xtouch)) m |= 1u << (kh * g.KW + kw);
        }
    return m;
}

extern "C" int otvae_conv_dead_taps(const otvae_conv_geom* gg, uint32_t* mask) {
    int rc = check_geom(gg, "otvae_conv_dead_taps");
    if (rc) return rc;
    OTVAE_REQUIRE(mask, "otvae_conv_dead_taps: NULL mask");
    *mask = dead_taps(to_geom(gg));
    return OTVAE_OK;
}

extern "C" int otvae_wgrad_reduce_batched(int n, const float* const* partial, const int* P, const int* K, const int* Kp,
                                          const int* Cn, float* const* gw, float* const* gb, const int* Cs,
                                          const uint32_t* dead, void* stream) {
    OTVAE_REQUIRE(n > 0 && partial && P && K && Kp && Cn && gw && gb, "otvae_wgrad_reduce_batched: bad argument");
    OTVAE_REQUIRE((Cs == nullptr) == (dead == nullptr), "otvae_wgrad_reduce_batched: Cs and dead come together");
    hipStream_t st = (hipStream_t)stream;
    for (int base = 0; base < n; base += WRB_MAX) {
        WrbDesc d = {};
        const int m = imin(WRB_MAX, n - base);
        size_t maxwork = 1;
        for (int i = 0; i < m; ++i) {
            const int j = base + i;
            OTVAE_REQUIRE(partial[j] && gw[j] && P[j] > 0 && Kp[j] >= K[j] && Cn[j] > 0, "otvae_wgrad_reduce_batched: entry %d", j);
            d.partial[i] = partial[j];
            d.gw[i] = gw[j];
            d.gb[i] = gb[j];
            d.P[i] = P[j];
            d.K[i] = K[j];
            d.Kp[i] = Kp[j];
            d.Cn[i] = Cn[j];
            if (dead && dead[j]) {
                OTVAE_REQUIRE(Cs[j] > 0 && K[j] % Cs[j] == 0 && K[j] / Cs[j] <= 32, "otvae_wgrad_reduce_batched: entry %d: Cs", j);
                d.Cs[i] = Cs[j];
                d.dead[i] = dead[j];
            }
            const size_t total = (size_t)Kp[j] * Cn[j];
            const size_t work = P[j] < 16 ? cdiv(total, 256) : (total >= 4096 ? cdiv(total, 64) : cdiv(total, 4));
            if (work > maxwork) maxwork = work;
        }
        dim3 grid(imin((int)maxwork, 512), m);
        wgrad_reduce_batched_kernel<<<grid, 256, 0, st>>>(d);
        OTVAE_CHECK_LAUNCH("otvae_wgrad_reduce_batched");
    }
    return OTVAE_OK;
}

// ------------------------------------------------------------------------------------------------ multi-job launch
// Up to CJ_MAX independent ConvLayer kernels (forward / data-gradient / weight-gradient bodies above, vector path) in
// ONE launch: block b belongs to the job j with block0[j] <= b < block0[j+1] and runs that job's body with its own
// virtual (bx, by, bz).  At this model's layer sizes a single job leaves most of the 256 CUs idle (10-500 workgroups),
// and every launch costs ~4-5 us of fixed dispatch/flush time; the two branches of a ConvBlock and the wgrad/dgrad
// pairs of the backward pass are independent, so they share a launch.
#define CJ_MAX 4
struct DevJob {
    Geom g;
    const float* a0;      // fwd: x            dgrad: gy           wgrad: x
    const float* scale;
    const float* shift;
    const float* b0;      // fwd: HWIO weight  dgrad: wD           wgrad: gy
    const float* bias;    // fwd
    const float* res;     // fwd
    const float* xin;     // dgrad: x
    const float* mean;    // dgrad
    const float* invstd;  // dgrad
    float* out;           // fwd: y            dgrad: gv           wgrad: partial workspace
    double* partial;      // fwd: stat partial dgrad: bn partial
    int kind, NT, relu, cpad, Kp, has_bias, skip_dead;
    unsigned chunk;
    int gx, gy, gz, block0;
    BnFold fold;          // fwd: the BatchNorm of x folded into this launch (fold.slots != NULL)
};
struct DevJobs {
    int n;
    DevJob j[CJ_MAX];
};

template <bool UT>
__global__ __launch_bounds__(256) void conv_jobs_kernel(DevJobs t) {
    extern __shared__ __align__(16) char jobs_smem[];
    int ji = 0;
#pragma unroll
    for (int i = 1; i < CJ_MAX; ++i)
        if (i < t.n && (int)blockIdx.x >= t.j[i].block0) ji = i;
    const DevJob& J = t.j[ji];
    const int lb = (int)blockIdx.x - J.block0;
    const int bx = lb % J.gx;
    const int r = lb / J.gx;
    const int by = r % J.gy, bz = r / J.gy;
    // forward jobs with a BatchNorm in front read its affine from LDS: folded from the statistic slots, or copied from the arrays a
    // finalize launch left (the host packs such a job only with Cs <= BN_TAB)
    __shared__ __align__(16) float bn_tab[2][BN_TAB];
    const bool fwd_norm = J.kind == OTVAE_JOB_FWD && (J.scale != nullptr || J.fold.slots != nullptr);
    if (fwd_norm) bn_tab_fill(J.fold, J.scale, J.shift, J.g.Cs, bn_tab[0], bn_tab[1], lb == 0);
    const float* const fsc = fwd_norm ? bn_tab[0] : nullptr;
    const float* const fsh = fwd_norm ? bn_tab[1] : nullptr;
#define CJ_FWD(N_)                                                                                                        \
    conv_gemm_body<0, N_, true, UT>(jobs_smem, J.g, J.a0, fsc, fsh, J.relu, J.b0, J.bias, J.res, J.out, nullptr, nullptr, \
                                nullptr, nullptr, J.partial, J.cpad, bx, by, bz, J.gx, J.gz)
#define CJ_DGRAD(N_)                                                                                                      \
    conv_gemm_body<1, N_, true, UT>(jobs_smem, J.g, J.a0, J.scale, J.shift, J.relu, J.b0, nullptr, nullptr, nullptr, J.xin,     \
                                J.mean, J.invstd, J.out, J.partial, J.cpad, bx, by, bz, J.gx, J.gz)
#define CJ_WGRAD(N_) \
    conv_wgrad_body<N_, true>(jobs_smem, J.g, J.a0, J.scale, J.shift, J.relu, J.b0, J.out, J.Kp, J.has_bias, J.skip_dead, \
                              J.chunk, bx, by, bz)
    switch (J.kind * 4 + J.NT - 1) {
        case 0: CJ_FWD(1); break;
        case 1: CJ_FWD(2); break;
        case 2: CJ_FWD(3); break;
        case 3: CJ_FWD(4); break;
        case 4: CJ_DGRAD(1); break;
        case 5: CJ_DGRAD(2); break;
        case 6: CJ_DGRAD(3); break;
        case 7: CJ_DGRAD(4); break;
        case 8: CJ_WGRAD(1); break;
        case 9: CJ_WGRAD(2); break;
        case 10: CJ_WGRAD(3); break;
        default: CJ_WGRAD(4); break;
    }
#undef CJ_FWD
#undef CJ_DGRAD
#undef CJ_WGRAD
}

static size_t job_smem_bytes(int kind, int NT) {
    if (kind == OTVAE_JOB_BWD_WEIGHT) {
        switch (NT) {
            case 1: return WgradSmem<1>::bytes;
            case 2: return WgradSmem<2>::bytes;
            case 3: return WgradSmem<3>::bytes;
            default: return WgradSmem<4>::bytes;
        }
    }
    switch (NT) {
        case 1: return GemmSmem<1>::bytes;
        case 2: return GemmSmem<2>::bytes;
        case 3: return GemmSmem<3>::bytes;
        default: return GemmSmem<4>::bytes;
    }
}

static int run_single_job(const otvae_conv_job& jb, void* stream) {
    switch (jb.kind) {
        case OTVAE_JOB_FWD: {
            BnFold fold;
            if (int rc = bn_fold_from_abi("otvae_conv_multi", jb.fold, jb.geom.Cs, &fold)) return rc;
            return conv_fwd_ex(&jb.geom, jb.x, jb.scale, jb.shift, jb.relu, jb.w, jb.bias, jb.residual, jb.y,
                               jb.stat_slots ? bn_tag_slots(jb.stat_slots, jb.stat_nslots) : jb.stat_partial, fold, stream);
        }
        case OTVAE_JOB_BWD_DATA:
            return conv_bwd_data_ex(&jb.geom, jb.gy, jb.w, jb.x, jb.scale, jb.shift, jb.relu, jb.mean, jb.invstd, jb.gv,
                                    jb.bn_slots ? bn_tag_slots(jb.bn_slots, jb.bn_nslots) : jb.bn_partial, stream);
        default:
            return otvae_conv_bwd_weight(&jb.geom, jb.x, jb.scale, jb.shift, jb.relu, jb.gy, jb.has_bias, jb.wpartial, jb.gw,
                                         jb.gb, jb.defer_reduce, stream);
    }
}

// what the calling thread's last otvae_conv_multi did with its jobs (bit i: job i ran inside a packed conv_jobs_kernel launch)
static thread_local unsigned g_multi_packed_mask = 0;
static thread_local int g_multi_packed_ut = -1;

extern "C" int otvae_conv_multi_last(unsigned* packed_mask, int* uniform_tap) {
    OTVAE_REQUIRE(packed_mask && uniform_tap, "otvae_conv_multi_last: NULL argument");
    *packed_mask = g_multi_packed_mask;
    *uniform_tap = g_multi_packed_ut;
    return OTVAE_OK;
}

extern "C" int otvae_conv_multi(int n, const otvae_conv_job* jobs, void* stream) {
    OTVAE_REQUIRE(n > 0 && jobs, "otvae_conv_multi: no jobs");
    hipStream_t st = (hipStream_t)stream;
    g_multi_packed_mask = 0;
    g_multi_packed_ut = -1;
    DevJobs pack = {};
    size_t smem = 0;
    int nblocks = 0;
    int packed_idx[CJ_MAX];
    int pack_ut = -1;  // uniform-tap flavour of the packed forward / data-gradient jobs (-1: none yet)
    auto flush = [&]() -> int {
        if (pack.n == 0) return OTVAE_OK;
        if (pack.n == 1) {  // nothing to share a launch with: the dedicated kernel (static LDS, same body)
            int rc = run_single_job(jobs[packed_idx[0]], stream);
            pack = {};
            smem = 0;
            nblocks = 0;
            pack_ut = -1;
            return rc;
        }
        if (pack_ut == 1)
            conv_jobs_kernel<true><<<nblocks, 256, smem, st>>>(pack);
        else
            conv_jobs_kernel<false><<<nblocks, 256, smem, st>>>(pack);
        OTVAE_CHECK_LAUNCH("otvae_conv_multi");
        for (int i = 0; i < pack.n; ++i) g_multi_packed_mask |= 1u << packed_idx[i];
        g_multi_packed_ut = pack_ut == 1 ? 1 : 0;
        for (int i = 0; i < pack.n; ++i) {  // immediate reductions of the packed weight-gradient jobs
            const otvae_conv_job& jb = jobs[packed_idx[i]];
            if (jb.kind != OTVAE_JOB_BWD_WEIGHT || jb.defer_reduce) continue;
            const DevJob& d = pack.j[i];
            const size_t total = (size_t)d.Kp * d.g.Cn;
            wgrad_reduce_kernel<<<imin(cdiv(total, d.gz >= 32 ? 4 : 64), 2048), 256, 0, st>>>(
                jb.wpartial, d.gz, d.Kp - (jb.has_bias ? 1 : 0), d.Kp, d.g.Cn, jb.gw, jb.gb);
            OTVAE_CHECK_LAUNCH("otvae_conv_multi(reduce)");
        }
        pack = {};
        smem = 0;
        nblocks = 0;
        pack_ut = -1;
        return OTVAE_OK;
    };
    for (int i = 0; i < n; ++i) {
        const otvae_conv_job& jb = jobs[i];
        OTVAE_REQUIRE(jb.kind >= OTVAE_JOB_FWD && jb.kind <= OTVAE_JOB_BWD_WEIGHT, "otvae_conv_multi: job %d: bad kind %d", i,
                      jb.kind);
        int rc = check_geom(&jb.geom, "otvae_conv_multi");
        if (rc) return rc;
        OTVAE_REQUIRE((jb.scale == nullptr) == (jb.shift == nullptr), "otvae_conv_multi: job %d: scale/shift must come together", i);
        OTVAE_REQUIRE(jb.kind == OTVAE_JOB_FWD || (jb.fold.slots == nullptr && jb.stat_slots == nullptr),
                      "otvae_conv_multi: job %d: fold / stat_slots belong to forward jobs", i);
        Geom g = to_geom(&jb.geom);
        DevJob d = {};
        d.g = g;
        d.kind = jb.kind;
        d.relu = jb.relu;
        d.scale = jb.scale;
        d.shift = jb.shift;
        bool packable = false;
        const bool ch4 = (g.Cs % 4 == 0) && (g.Cn % 4 == 0);
        if (jb.kind == OTVAE_JOB_FWD) {
            OTVAE_REQUIRE(jb.x && jb.w && jb.y, "otvae_conv_multi: job %d (forward): NULL tensor", i);
            dim3 grid;
            fwd_grid(g, d.NT, grid, d.cpad);
            rc = bn_fold_from_abi("otvae_conv_multi", jb.fold, g.Cs, &d.fold);
            if (rc) return rc;
            OTVAE_REQUIRE(!(jb.stat_partial && jb.stat_slots), "otvae_conv_multi: job %d: statistics go to partials OR to slots", i);
            OTVAE_REQUIRE_SLOTS("otvae_conv_multi (stat_slots)", jb.stat_slots, jb.stat_nslots);
            OTVAE_REQUIRE(jb.fold.slots == nullptr || g.Cs <= BN_TAB, "otvae_conv_multi: job %d: a folded BatchNorm takes at most %d channels", i, BN_TAB);
            packable = !conv_small_ok(g) && ch4 && aligned16(jb.x) && aligned16(jb.w) &&
                       (jb.scale == nullptr || (aligned16(jb.scale) && aligned16(jb.shift))) &&
                       ((jb.scale == nullptr && jb.fold.slots == nullptr) || g.Cs <= BN_TAB);
            {
                TilePlan pl;
                dim3 tg;
                size_t sm;
                if (packable && conv_tile_plan(g, 0, pl, tg, sm)) packable = false;  // image-tile kernel: own launch
            }
            d.a0 = jb.x;
            d.b0 = jb.w;
            d.bias = jb.bias;
            d.res = jb.residual;
            d.out = jb.y;
            d.partial = jb.stat_slots ? bn_tag_slots(jb.stat_slots, jb.stat_nslots) : jb.stat_partial;
            d.gx = grid.x, d.gy = grid.y, d.gz = grid.z;
        } else if (jb.kind == OTVAE_JOB_BWD_DATA) {
            OTVAE_REQUIRE(jb.gy && jb.w && jb.gv, "otvae_conv_multi: job %d (data gradient): NULL tensor", i);
            OTVAE_REQUIRE((jb.mean == nullptr) == (jb.invstd == nullptr), "otvae_conv_multi: job %d: mean/invstd must come together", i);
            OTVAE_REQUIRE(!(jb.relu || jb.mean) || jb.x, "otvae_conv_multi: job %d: x needed for the ReLU mask / BatchNorm sums", i);
            OTVAE_REQUIRE(!jb.mean || jb.bn_partial || jb.bn_slots, "otvae_conv_multi: job %d: bn_partial workspace / bn_slots missing", i);
            OTVAE_REQUIRE(!(jb.bn_partial && jb.bn_slots), "otvae_conv_multi: job %d: BatchNorm-backward sums go to partials OR to slots", i);
            OTVAE_REQUIRE_SLOTS("otvae_conv_multi (bn_slots)", jb.bn_slots, jb.bn_nslots);
            dim3 grid;
            dgrad_grid(g, d.NT, grid, d.cpad);
            packable = !conv_small_ok(g) && ch4 && aligned16(jb.gy) && aligned16(jb.w);
            {
                TilePlan pl;
                dim3 tg;
                size_t sm;
                if (packable && conv_tile_plan(g, 1, pl, tg, sm)) packable = false;  // image-tile kernel: own launch
            }
            d.a0 = jb.gy;
            d.b0 = jb.w;
            d.xin = jb.x;
            d.mean = jb.mean;
            d.invstd = jb.invstd;
            d.out = jb.gv;
            d.partial = jb.bn_slots ? bn_tag_slots(jb.bn_slots, jb.bn_nslots) : jb.bn_partial;
            d.gx = grid.x, d.gy = grid.y, d.gz = grid.z;
        } else {
            OTVAE_REQUIRE(jb.x && jb.gy && jb.wpartial && jb.gw, "otvae_conv_multi: job %d (weight gradient): NULL tensor", i);
            OTVAE_REQUIRE(!jb.has_bias || jb.gb, "otvae_conv_multi: job %d: gb missing", i);
            int P, nkb, nnb;
            wgrad_plan(g, jb.has_bias, d.NT, P, d.chunk, nkb, nnb, d.Kp);
            packable = !conv_small_wgrad_ok(g) && ch4 && aligned16(jb.x) && aligned16(jb.gy) &&
                       (jb.scale == nullptr || (aligned16(jb.scale) && aligned16(jb.shift)));
            {
                WTilePlan wp;
                int nbk;
                size_t sm;
                if (conv_wtile_plan(g, jb.has_bias, wp, nbk, sm)) packable = false;  // image-tile kernel: own launch
            }
            d.a0 = jb.x;
            d.b0 = jb.gy;
            d.out = jb.wpartial;
            d.has_bias = jb.has_bias;
            d.skip_dead = jb.defer_reduce == OTVAE_DEFER_SPARSE;
            d.gx = nkb, d.gy = nnb, d.gz = P;
        }
        if (!packable) {
            rc = run_single_job(jb, stream);
            if (rc) return rc;
            continue;
        }
        int job_ut = -1;
        if (jb.kind != OTVAE_JOB_BWD_WEIGHT) job_ut = ((jb.kind == OTVAE_JOB_FWD ? g.Cs : g.Cn) % KC == 0) ? 1 : 0;
        if (pack.n == CJ_MAX || (job_ut >= 0 && pack_ut >= 0 && job_ut != pack_ut)) {
            rc = flush();
            if (rc) return rc;
        }
        if (job_ut >= 0) pack_ut = job_ut;
        d.block0 = nblocks;
        nblocks += d.gx * d.gy * d.gz;
        const size_t need = job_smem_bytes(d.kind, d.NT);
        if (need > smem) smem = need;
        packed_idx[pack.n] = i;
        pack.j[pack.n++] = d;
    }
    return flush();
}
