/*
 * otvae.h -- C ABI of libotvae_hip.so, the MI355X (gfx950) implementation of the training-step hot path of
 * theoad/ot-vae-lightning.  Plain pointers and sizes only (no torch / HIP types): `stream` is a hipStream_t
 * passed as void* (NULL = default stream).  All pointers are DEVICE pointers unless stated otherwise.
 *
 * Conventions
 *   - Activations are fp32 NHWC ([N][H][W][C] contiguous).  Conv weights are HWIO ([KH][KW][Cin][Cout]); the
 *     Python side keeps the reference's logical OIHW shape on a tensor whose strides give this memory order.
 *   - Every function only enqueues work on `stream`: no allocation, no host synchronisation, no global state, so
 *     all of them can be captured into a hipGraph.  Workspaces are supplied by the caller.
 *   - Return value: OTVAE_OK or a negative OTVAE_E* code (bad shape/argument -> nothing was launched).
 *   - There is no CPU implementation behind this ABI.
 *
 * Each entry point cites the reference code (paths under ot_vae_lightning/) whose arithmetic it replaces.
 */
#ifndef OTVAE_H
#define OTVAE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OTVAE_OK 0
#define OTVAE_EINVAL (-1)       /* bad argument / shape */
#define OTVAE_EUNSUPPORTED (-2) /* valid but not implemented for this size */
#define OTVAE_ELAUNCH (-3)      /* hipLaunch / runtime error */

int otvae_abi_version(void);           /* bumps when a signature changes */
const char* otvae_last_error(void);    /* host string describing the last non-OK return of this thread */
int otvae_device_info(int* n_cu, int* wave_size, char* arch, int arch_len); /* host query helper */
/* A HIP stream of the library's own on the calling thread's current device (hipStreamNonBlocking), for hosts whose framework hands out
 * streams from a small round-robin pool (torch: 32 per device -- the 33rd "new" stream is the first one again, and two lanes of one
 * captured step end up on one queue).  The Python mirror recycles the streams it creates and never destroys them. */
int otvae_stream_create(void** stream);
int otvae_stream_destroy(void* stream);

/* Geometry of one ConvLayer (networks/cnn.py:48-154,183-192): input [N][Hs][Ws][Cs] --(nearest x`up`)-->
 * conv KHxKW / stride / pad --> [N][Ho][Wo][Cn].  `up` is 1 or 2; up==2 requires stride==1. */
typedef struct {
    int32_t N, Hs, Ws, Cs;
    int32_t up;
    int32_t Ho, Wo, Cn;
    int32_t KH, KW, stride, pad;
} otvae_conv_geom;

/* ---- BatchNorm2d, training mode (networks/cnn.py:122,184) ------------------------------------------------- */
/* per-channel partial sums of x[M][C]: partial[2][C][P] (double, P fastest), P = otvae_bn_stats_nparts(M, C). */
int otvae_bn_stats_nparts(int64_t M, int C);
int otvae_bn_stats(const float* x, int64_t M, int C, double* partial, void* stream);
/* mean/invstd from the partials; for each of `n_bn` BatchNorm modules sharing this input (a ConvBlock's
 * block[0] and skip normalise the same tensor) writes scale = gamma*invstd, shift = beta - mean*scale and updates
 * running_mean/var (momentum, unbiased var) and num_batches_tracked.  Arrays of n_bn device pointers are HOST
 * arrays. running pointers may be NULL (no update).  `ld` = row stride of the partials (C for otvae_bn_stats, CnPad when
 * they come from the producing conv's epilogue, see otvae_conv_fwd). */
int otvae_bn_finalize(const double* partial, int P, int ld, int64_t M, int C, float eps, float momentum,
                      float* mean, float* invstd, int n_bn,
                      const float* const* gamma, const float* const* beta,
                      float* const* running_mean, float* const* running_var, int64_t* const* num_batches_tracked,
                      float* const* scale, float* const* shift, void* stream);

/* ---- ConvLayer.forward: y = conv(up(relu?(x*scale+shift))) + bias (+ residual) ---------------------------- */
/* scale/shift NULL -> no normalisation; relu applies after the affine; bias/residual NULL -> absent.
 * residual has y's shape (ConvBlock `out + skip(x)`, networks/cnn.py:334). wT is the HWIO weight. */
/* stat_partial (nullable): fp64 [2][CnPad][P] per-block partial sums (sum y, sum y^2) per output channel, written by
 * the epilogue so that the NEXT layer's BatchNorm needs no separate pass over y; P, CnPad from _stats_ws. */
int otvae_conv_fwd_stats_ws(const otvae_conv_geom* g, int* P, int* CnPad);
int otvae_conv_fwd(const otvae_conv_geom* g, const float* x, const float* scale, const float* shift, int relu,
                   const float* wT, const float* bias, const float* residual, float* y, double* stat_partial,
                   void* stream);

/* HWIO [T][Cs][Cn] -> [T][Cn][Cs] (the dgrad operand layout) */
int otvae_weight_transpose(const float* wT, float* wD, int T, int Cs, int Cn, void* stream);
/* the same for every conv weight of a model in one launch: device table[n_layers][5] of int64
 * {src element offset (from src_base), dst element offset (from dst_base), T, Cs, Cn}; max_elems = largest T*Cs*Cn */
int otvae_weight_transpose_batched(const float* src_base, float* dst_base, const int64_t* table, int n_layers,
                                   int64_t max_elems, void* stream);

/* ---- ConvLayer backward ------------------------------------------------------------------------------------ */
/* Data gradient.  gv[N][Hs][Ws][Cs] = d loss / d (x*scale+shift) i.e. the gradient entering BatchNorm's output
 * (after the nearest-upsample sum and the ReLU mask recomputed from x).  With mean/invstd != NULL also writes
 * the per-block fp64 partial sums bn_partial[2][CsPad][P] of (gv, gv*xhat) that BatchNorm backward needs;
 * P and CsPad from otvae_conv_bwd_data_ws. */
int otvae_conv_bwd_data_ws(const otvae_conv_geom* g, int* P, int* CsPad);
int otvae_conv_bwd_data(const otvae_conv_geom* g, const float* gy, const float* wD,
                        const float* x, const float* scale, const float* shift, int relu,
                        const float* mean, const float* invstd,
                        float* gv, double* bn_partial, void* stream);
/* BatchNorm backward for up to two branches that normalise the same x (block[0] and skip):
 * finalize: reduces the partials, writes dgamma/dbeta of each branch and the coefficient vectors coef[(2+nb)][C]
 * apply:    dx = sum_b coef[2+b][c]*gv_b - coef[0][c]*x - coef[1][c]   (elementwise, M*C elements) */
int otvae_bn_bwd_finalize(int nb, const double* const* bn_partial, const int* P, int CsPad, int64_t M, int C,
                          const float* mean, const float* invstd, const float* const* gamma,
                          float* const* dgamma, float* const* dbeta, float* coef, void* stream);
int otvae_bn_bwd_apply(int nb, const float* const* gv, const float* x, const float* coef, int64_t M, int C,
                       float* dx, void* stream);

/* Weight (+bias) gradient.  Workspace partial[P][K+hasb][Cn] floats with K = KH*KW*Cs; P from _ws.
 * gw is written in HWIO order, gb[Cn] if has_bias. */
int otvae_conv_bwd_weight_ws(const otvae_conv_geom* g, int has_bias, int* P);
int otvae_conv_bwd_weight(const otvae_conv_geom* g, const float* x, const float* scale, const float* shift, int relu,
                          const float* gy, int has_bias, float* partial, float* gw, float* gb, int defer_reduce,
                          void* stream);
/* With defer_reduce != 0 only the partials are written; the caller later reduces any number of layers in one launch
 * (host arrays of n entries; K = KH*KW*Cs, Kp = K + has_bias, P from otvae_conv_bwd_weight_ws).
 * Taps that touch the image for no output position (8 of the 9 taps of a 3x3 layer on a 1x1 map, 12 of the 16 of a
 * 4x4 stride-2 layer from 2x2 to 1x1) have an exactly zero gradient: otvae_conv_dead_taps returns them (bit kh*KW+kw).
 * With defer_reduce == OTVAE_DEFER_SPARSE the weight-gradient kernels may leave the partial rows of those taps
 * unwritten; the caller must then pass the layer's Cs and mask to otvae_wgrad_reduce_batched, which writes zeros there
 * without reading the partials (Cs and dead may both be NULL: every row is read). */
#define OTVAE_DEFER_DENSE 1
#define OTVAE_DEFER_SPARSE 2
int otvae_conv_dead_taps(const otvae_conv_geom* g, uint32_t* mask);
int otvae_wgrad_reduce_batched(int n, const float* const* partial, const int* P, const int* K, const int* Kp,
                               const int* Cn, float* const* gw, float* const* gb, const int* Cs, const uint32_t* dead,
                               void* stream);

/* ---- several independent ConvLayer kernels in ONE launch ----------------------------------------------------
 * The two branches of a ConvBlock (block[0] and skip read the same x, networks/cnn.py:311-335) in the forward pass,
 * and the weight- and data-gradient of every branch in the backward pass, are independent of each other; at the
 * layer sizes of this model each one alone fills a fraction of the 256 CUs.  otvae_conv_multi gives exactly the
 * results of otvae_conv_fwd / otvae_conv_bwd_data / otvae_conv_bwd_weight called once per job (same kernels bodies,
 * same fixed-order reductions, same workspace layouts), but packs the MFMA-path jobs into one launch whose workgroups
 * are divided among the jobs; jobs on another path (tiny channel counts) are launched one by one.  No job may read
 * what another job of the same call writes.  `jobs` is a HOST array. */
#define OTVAE_JOB_FWD 0
#define OTVAE_JOB_BWD_DATA 1
#define OTVAE_JOB_BWD_WEIGHT 2
/* ---- BatchNorm statistic SLOTS (round 4): cross-block sums without a finalize launch, bit-reproducible --------------------------------
 * Instead of P per-block partials a producer may add its per-channel sums into `nslots` (a power of two <= 64: the caller's choice, the
 * same number for the producer and the consumer of a buffer; atomics on one address are performed one after the other, ~0.1 us each,
 * so nslots should grow with the producer's block count) accumulators of int64 fixed-point limbs with integer atomics (associative:
 * the totals do not depend on arrival order); layout and arithmetic in csrc/common.h.  `slots` buffers hold
 * otvae_bn_slots_words(ld, nslots) int64 words, are 16-byte
 * aligned and must be ZERO before the producer runs.  A consumer kernel that is handed an otvae_bn_fold
 * turns the sums into (scale, shift) in its own prologue -- every block for itself -- and its first block leaves mean / invstd / scale /
 * shift in global memory for the backward pass and advances the running buffers: what otvae_bn_finalize did in a launch of its own
 * (reference: nn.BatchNorm2d in training mode, networks/cnn.py:122,184).  otvae_bn_finalize_slots is the stand-alone form. */
typedef struct otvae_bn_fold {
    const void* slots;            /* NULL: no fold */
    int32_t ld;                   /* channel stride of the slots (>= channels) */
    int32_t nslots;               /* slots in use (power of two <= 64): what the producer of `slots` was given */
    int64_t count;                /* elements per channel: N * H * W of the normalised tensor */
    float eps, momentum;
    const float* gamma;
    const float* beta;
    float* running_mean;          /* nullable (with running_var, num_batches_tracked) */
    float* running_var;
    int64_t* num_batches_tracked;
    float* mean_out;              /* nullable pair: the branch that publishes the statistics the branches share */
    float* invstd_out;
    float* scale_out;             /* this branch's affine, for the backward pass */
    float* shift_out;
} otvae_bn_fold;
int64_t otvae_bn_slots_words(int ld, int nslots);
int otvae_bn_stats_slots(const float* x, int64_t M, int C, void* slots, int ld, int nslots, void* stream);
int otvae_bn_finalize_slots(int n_bn, const otvae_bn_fold* folds, int C, void* stream);
/* The BatchNorm backward pair (otvae_bn_bwd_finalize + otvae_bn_bwd_apply below) as ONE launch, for sums (sum gv, sum gv * xhat) that a
 * data-gradient job left in slots (otvae_conv_job.bn_slots, otvae_attn_stage_bwd_slots): the finalize arithmetic runs in the launch's
 * prologue, its first block writes dgamma / dbeta.  dx == NULL: parameter gradients only (gv / x may then be NULL).  training == 0:
 * eval-mode BatchNorm (a fixed affine: no batch-statistics terms in dx).  C <= 1024. */
int otvae_bn_bwd_apply_slots(int nb, const float* const* gv, const float* x, const void* const* slots, const int* nslots, int ld,
                             int64_t M, int C, const float* mean, const float* invstd, const float* const* gamma,
                             float* const* dgamma, float* const* dbeta, int training, float* dx, void* stream);

typedef struct otvae_conv_job {
    int32_t kind;               /* OTVAE_JOB_* */
    int32_t relu;               /* ReLU after the (optional) affine of the layer INPUT x */
    int32_t has_bias;           /* BWD_WEIGHT */
    int32_t defer_reduce;       /* BWD_WEIGHT: 0, OTVAE_DEFER_DENSE or OTVAE_DEFER_SPARSE */
    otvae_conv_geom geom;
    const float* x;             /* layer input [N][Hs][Ws][Cs] (BWD_DATA: only for the ReLU mask / BatchNorm sums) */
    const float* scale;         /* BatchNorm scale/shift of x, or NULL */
    const float* shift;
    const float* w;             /* FWD: HWIO weight; BWD_DATA: dgrad-layout weight (otvae_weight_transpose) */
    const float* bias;          /* FWD */
    const float* residual;      /* FWD */
    float* y;                   /* FWD */
    double* stat_partial;       /* FWD (nullable) */
    const float* gy;            /* BWD_DATA / BWD_WEIGHT: gradient of the layer output */
    const float* mean;          /* BWD_DATA (nullable, with invstd and bn_partial) */
    const float* invstd;
    float* gv;                  /* BWD_DATA */
    double* bn_partial;
    float* wpartial;            /* BWD_WEIGHT workspace */
    float* gw;                  /* BWD_WEIGHT */
    float* gb;
    void* stat_slots;           /* FWD (nullable, instead of stat_partial): statistic slots of the OUTPUT, channel stride = the ld of otvae_conv_fwd_stats_ws */
    void* bn_slots;             /* BWD_DATA (nullable, instead of bn_partial): statistic slots of the BatchNorm-backward sums, channel stride = the CsPad of otvae_conv_bwd_data_ws */
    int32_t stat_nslots;        /* slots in use in stat_slots / bn_slots (power of two <= 64) */
    int32_t bn_nslots;
    otvae_bn_fold fold;         /* FWD (fold.slots nullable): the BatchNorm of the INPUT x folded into this launch; scale / shift are then ignored */
} otvae_conv_job;
int otvae_conv_multi(int n, const otvae_conv_job* jobs, void* stream);
/* Introspection for measurement (bench.py's per-kernel roofline): what the calling thread's last otvae_conv_multi did --
 * bit i of packed_mask: job i ran inside one packed conv_jobs_kernel launch; uniform_tap: that launch's template flavour
 * (1 = conv_jobs_kernel<true>), -1 if nothing was packed. */
int otvae_conv_multi_last(unsigned* packed_mask, int* uniform_tap);

/* ---- QKVAttention (networks/nets_utils.py:63-82) ---------------------------------------------------------- */
/* qkv [N][T][3*H*C] (channel = which*H*C + h*C + c) -> out [N][T][H*C]; lse [N][H][T] saved for backward.
 * aux (nullable; honoured for head widths C <= 2, ignored otherwise): [N][H][T][C*C] per-query covariance of values and
 * keys under the attention weights, D[c'][c] = sum_s p_s (v_s[c'] - out[c']) k_s[c].  Passing the same buffer to
 * otvae_attn_bwd lets it form dq[c] = sum_c' gout[c'] D[c'][c] without a pass over the keys; with aux == NULL it
 * recomputes the pairs. */
int otvae_attn_fwd(const float* qkv, int N, int T, int H, int C, float* out, float* lse, float* aux, void* stream);
int otvae_attn_bwd(const float* qkv, const float* out, const float* lse, const float* gout, const float* aux,
                   int N, int T, int H, int C, float* gqkv, void* stream);
/* The same kernels with an explicit score scale: scores = scale * q.k (otvae_attn_fwd / _bwd use 1/C, the product of the
 * two C^-1/2 factors of QKVAttention; torch.nn.MultiheadAttention inside the reference's ViT, networks/vit.py:169-172,
 * uses 1/sqrt(C)). */
int otvae_attn_fwd_scaled(const float* qkv, int N, int T, int H, int C, float scale, float* out, float* lse, float* aux,
                          void* stream);
int otvae_attn_bwd_scaled(const float* qkv, const float* out, const float* lse, const float* gout, const float* aux, int N,
                          int T, int H, int C, float scale, float* gqkv, void* stream);

/* The whole AttentionBlock forward (networks/cnn.py:212-240: proj_out(attention(qkv(BN(x)))) [+ the ConvBlock's skip, cnn.py:331-335])
 * as ONE launch: a workgroup owns whole images, forms q / k / v of its tokens from the normalised input (scale / shift [H*C] = the
 * BatchNorm affine in front of the bias-free 1x1 qkv convolution, both NULL without one; wqkv [H*C][3*H*C], wproj [H*C][H*C] in the
 * HWIO order of otvae_conv_fwd), runs the attention of otvae_attn_fwd_scaled and applies the bias-free 1x1 output projection,
 * the residual sum (nullable) and the per-channel partial sums of y for the next BatchNorm (stat_partial [2][H*C][rows], nullable;
 * rows from otvae_attn_stage_plan).  qkv [N][T][3*H*C] (nullable: only a three-launch backward pass reads it), out [N][T][H*C] (the attention output), lse and aux are written as by the
 * three separate launches: the backward pass is theirs.  otvae_attn_stage_plan returns OTVAE_EUNSUPPORTED (no error text) for
 * shapes the fused kernel does not take (T == 1, T % 4 != 0, a width that is not a power of two <= 32, heads that do not fit one
 * workgroup): the caller then issues the three launches. */
int otvae_attn_stage_plan(int N, int T, int H, int C, int need_aux, int* stat_rows);
int otvae_attn_stage_fwd(const float* x, const float* scale, const float* shift, const float* wqkv, const float* wproj,
                         const float* residual, int N, int T, int H, int C, float qk_scale, float* qkv, float* out, float* lse,
                         float* aux, float* y, double* stat_partial, void* stream);
/* The same launch with the BatchNorm in front of the qkv convolution FOLDED IN (fold->slots != NULL: scale / shift are ignored, see
 * otvae_bn_fold) and / or the output's statistics into statistic slots (stat_slots, channel stride H * C) instead of stat_partial. */
int otvae_attn_stage_fwd_fold(const float* x, const otvae_bn_fold* fold, const float* scale, const float* shift, const float* wqkv,
                              const float* wproj, const float* residual, int N, int T, int H, int C, float qk_scale, float* qkv,
                              float* out, float* lse, float* aux, float* y, double* stat_partial, void* stat_slots, int stat_nslots,
                              void* stream);

/* The AttentionBlock's backward pass as ONE launch on what otvae_attn_stage_fwd wrote (qkv, out, lse, aux): the attention output's
 * gradient is formed from gy [N][T][H*C] (gout = gy . wproj^T), the attention backward of otvae_attn_bwd_scaled writes gqkv
 * [N][T][3*H*C] (read afterwards by the qkv weight-gradient job), and gv [N][T][H*C] = gqkv . wqkv^T, the gradient of the qkv
 * convolution's normalised input, leaves with the BatchNorm-backward partial sums bn_partial [2][H*C][rows] (sum gv, sum gv * xhat per
 * channel, xhat = (x - mean) * invstd: what otvae_conv_bwd_data emits for otvae_bn_bwd_finalize; mean / invstd / x / bn_partial all
 * NULL without a BatchNorm).  qkv == NULL (otvae_attn_stage_fwd was given qkv == NULL and wrote none: 12 of the 28 bytes per value it
 * moves at the small maps): q / k / v are formed again from x through scale / shift, the forward kernel's arithmetic.  rows from
 * otvae_attn_stage_bwd_plan, which returns OTVAE_EUNSUPPORTED for shapes the kernel does not
 * take (the caller then issues the three launches). */
int otvae_attn_stage_bwd_plan(int N, int T, int H, int C, int* bn_rows);
int otvae_attn_stage_bwd(const float* gy, const float* wproj, const float* wqkv, const float* x, const float* mean,
                         const float* invstd, const float* scale, const float* shift, const float* qkv, const float* out,
                         const float* lse, const float* aux, int N, int T, int H, int C, float qk_scale, float* gqkv, float* gv,
                         double* bn_partial, void* stream);
/* The same launch with the BatchNorm-backward sums into statistic slots (channel stride H * C; consumed by otvae_bn_bwd_apply_slots). */
int otvae_attn_stage_bwd_slots(const float* gy, const float* wproj, const float* wqkv, const float* x, const float* mean,
                               const float* invstd, const float* scale, const float* shift, const float* qkv, const float* out,
                               const float* lse, const float* aux, int N, int T, int H, int C, float qk_scale, float* gqkv, float* gv,
                               void* bn_slots, int bn_nslots, void* stream);

/* Element-wise dropout with the same counter-based masks (keep(row, col) of a [rows][D] tensor, D % 4 == 0), optionally
 * fused with the ReLU in front of it: y = keep ? act(x)/(1-p) : 0.  relu != 0: the dropout(relu(linear1(x))) of a training-
 * mode nn.TransformerEncoderLayer; relu == 0: PositionalEmbedding's embedding dropout (networks/vit.py:54-58).  The backward
 * recomputes the mask from used[0] and reads x for the ReLU gate; otvae_layernorm_dropout_mask returns the same mask. */
int otvae_dropout_fwd(const float* x, int64_t rows, int D, int relu, float p, const int64_t* key, int stream_id, float* y,
                      int64_t* used, void* stream);
int otvae_dropout_bwd(const float* x, const float* gy, int64_t rows, int D, int relu, float p, const int64_t* used, float* gx,
                      void* stream);

/* Self-attention with dropout on the attention probabilities: nn.MultiheadAttention(dropout=p) in training mode, which is
 * how the reference's ViT builds every layer (networks/vit.py:157-172 hand `dropout` to nn.TransformerEncoderLayer;
 * configs/vae/vit.yaml trains with 0.1).  P = softmax(scale * q k^T), out = (P o keep / (1-p)) v, keep ~ Bernoulli(1-p).
 * The T x T mask is never stored: keep[t][s] is a counter-based hash of (call key, slice, t, s) that the backward pass
 * recomputes.  key: device int64[2] {seed, call counter} -- device memory, so that a captured hipGraph draws a fresh mask
 * on every replay once the host bumps the counter with a captured add; stream_id (0..4094) tells call sites apart.  The
 * forward writes the call key it derived to used[0] (device int64[1]); the backward and _mask read it from there.
 * causal != 0 restricts the softmax of token t to tokens s <= t (the ViT's `causal_mask`, networks/vit.py:215-217); p may
 * then be 0.  lse is the natural log of the UN-dropped row sums.  T <= 256 and T*(2C+3) <= 16384; C in {1,2,4,8,16,32}.
 * otvae_attn_dropout_mask writes keep as uint8 [N][H][T][T] (test / debugging aid). */
int otvae_attn_dropout_fwd(const float* qkv, int N, int T, int H, int C, float scale, float p, int causal, const int64_t* key,
                           int stream_id, float* out, float* lse, int64_t* used, void* stream);
int otvae_attn_dropout_bwd(const float* qkv, const float* out, const float* lse, const float* gout, int N, int T, int H,
                           int C, float scale, float p, int causal, const int64_t* used, float* gqkv, void* stream);
int otvae_attn_dropout_mask(int N, int T, int H, float p, const int64_t* used, uint8_t* keep, void* stream);
/* Cross-attention: nn.MultiheadAttention(query, memory, memory) inside the nn.TransformerDecoderLayer of the ViT's
 * `preprocess_depth` variant (networks/vit.py:171-181, 240-244): Tq queries of one token set against Tk keys / values of another,
 * scores = scale * q.k, the same dropout on the probabilities as above (p may be 0: key / used may then be NULL).  q, k and v
 * are addressed as ptr[n * img_stride + t * row_stride + h * C + c] (strides in floats), so thirds of one in-projected
 * tensor or three separate tensors both fit; out [N][Tq][H*C], lse [N][H][Tq]; the backward writes gq / gk / gv through their
 * own strides and touches nothing else.  max(Tq, Tk) <= 256 and max(Tq, Tk)*(2C+3) <= 16384; C in {1,2,4,8,16,32}. */
int otvae_attn_cross_fwd(const float* q, int64_t q_img_stride, int q_row_stride, const float* k, const float* v,
                         int64_t kv_img_stride, int kv_row_stride, int N, int Tq, int Tk, int H, int C, float scale, float p,
                         const int64_t* key, int stream_id, float* out, float* lse, int64_t* used, void* stream);
int otvae_attn_cross_bwd(const float* q, int64_t q_img_stride, int q_row_stride, const float* k, const float* v,
                         int64_t kv_img_stride, int kv_row_stride, const float* out, const float* lse, const float* gout, int N,
                         int Tq, int Tk, int H, int C, float scale, float p, const int64_t* used, float* gq,
                         int64_t gq_img_stride, int gq_row_stride, float* gk, float* gv, int64_t gkv_img_stride,
                         int gkv_row_stride, void* stream);
/* keep as uint8 [N][H][Tq][Tk] of the cross-attention call whose forward left `used` (test / debugging aid) */
int otvae_attn_cross_mask(int N, int Tq, int Tk, int H, float p, const int64_t* used, uint8_t* keep, void* stream);

/* ---- LayerNorm over the last dimension (the token streams of the ViT: networks/vit.py:38,54 and the two norms of each
 * nn.TransformerEncoderLayer, :169-172; torch.nn.functional.layer_norm arithmetic) ---------------------------------------
 * y[m][:] = (s - mean(s)) * rstd(s) * gamma + beta with s = x[m][:] + res[m][:] (res nullable: the "x + sublayer(x)" of the
 * post-norm block summed in; s is written to sum_out [M][D], which otvae_layernorm_bwd takes as xs; without a residual
 * sum_out may be NULL and xs = x).  mean / rstd [M] are saved for backward.  D <= 2048.
 * Backward: gx [M][D] (the gradient of both x and res), dgamma / dbeta [D]; ws: otvae_layernorm_bwd_ws(M, D) floats. */
int otvae_layernorm_fwd(const float* x, const float* res, const float* gamma, const float* beta, int M, int D, float eps,
                        float* sum_out, float* y, float* mean, float* rstd, void* stream);
int otvae_layernorm_bwd_ws(int M, int D);
int otvae_layernorm_bwd(const float* xs, const float* gy, const float* gamma, const float* mean, const float* rstd, int M, int D,
                        float* gx, float* dgamma, float* dbeta, float* ws, void* stream);
/* The "x + dropout(sublayer(x))" of a training-mode nn.TransformerEncoderLayer (networks/vit.py:157-172 with dropout > 0)
 * folded into the same kernels: s = res + x o keep / (1-p), y = LayerNorm(s).  keep(row, col) is the counter-based hash of
 * otvae_attn_dropout_* (key = device int64[2] {seed, call counter}, stream_id per call site, the forward leaves its call key
 * in used[0]); nothing but s is stored.  The backward returns gx = dL/ds (the residual's gradient) and
 * gx_dropped = gx o keep / (1-p) (the sublayer output's).  otvae_layernorm_dropout_mask: keep as uint8 [M][D] (test aid). */
int otvae_layernorm_dropout_fwd(const float* x, const float* res, const float* gamma, const float* beta, int M, int D,
                                float eps, float p, const int64_t* key, int stream_id, float* sum_out, float* y, float* mean,
                                float* rstd, int64_t* used, void* stream);
int otvae_layernorm_dropout_bwd(const float* xs, const float* gy, const float* gamma, const float* mean, const float* rstd,
                                int M, int D, float p, const int64_t* used, float* gx, float* gx_dropped, float* dgamma,
                                float* dbeta, float* ws, void* stream);
int otvae_layernorm_dropout_mask(int M, int D, float p, const int64_t* used, uint8_t* keep, void* stream);

/* ---- GaussianPrior (prior/gaussian.py:63-96) + Prior.forward scaling (prior/base.py:74-78) ---------------- */
/* h [B][S][2D] (S = H*W positions, channels-last) ; eps,z [B][S][D]; loss[B] = coeff * KL(q||N(0,I)) */
int otvae_gaussian_prior_fwd(const float* h, const float* eps, int B, int S, int D, float coeff,
                             float* z, float* loss, void* stream);
int otvae_gaussian_prior_bwd(const float* h, const float* eps, const float* gz, const float* gloss,
                             int B, int S, int D, float coeff, float* gh, void* stream);
/* the options of GaussianPrior (prior/gaussian.py:38-41,63-96; prior/base.py:65-68).  mode bit 0: empirical_kl -- the
 * Monte-Carlo estimate sum log q(z) - log p(z) at the drawn z instead of the closed form; bit 1: fixed_var -- q = N(h, s) with
 * s = 1, or temp[b] + 1e-8 when a per-sample temperature temp[B] is given (`time` of encode); h is then [B][S][D] (no
 * log-variance half).  mode 0 = otvae_gaussian_prior_fwd / _bwd. */
int otvae_gaussian_prior_ex_fwd(const float* h, const float* eps, const float* temp, int B, int S, int D, float coeff, int mode,
                                float* z, float* loss, void* stream);
int otvae_gaussian_prior_ex_bwd(const float* h, const float* eps, const float* temp, const float* gz, const float* gloss, int B,
                                int S, int D, float coeff, int mode, float* gh, void* stream);

/* ---- ConditionalGaussianPrior (prior/conditional_gaussian.py:84-93): the same re-parametrisation against a per-sample
 * diagonal prior N(prior_mean, exp(prior_log_std)^2) (rows of the class embeddings gathered by label).  h [B][2n]
 * (mu | log_var), eps, z, prior_mean, prior_log_std [B][n]; loss[B] = coeff * KL(q || p).  Backward also returns the
 * gradients of the gathered prior rows (nullable). */
int otvae_gaussian_prior_cond_fwd(const float* h, const float* eps, const float* prior_mean, const float* prior_log_std, int B,
                                  int n, float coeff, float* z, float* loss, void* stream);
int otvae_gaussian_prior_cond_bwd(const float* h, const float* eps, const float* prior_mean, const float* prior_log_std,
                                  const float* gz, const float* gloss, int B, int n, float coeff, float* gh,
                                  float* g_prior_mean, float* g_prior_log_std, void* stream);

/* The same with the options ConditionalGaussianPrior inherits from GaussianPrior (prior/conditional_gaussian.py:44-50 over
 * prior/gaussian.py:58-96, prior/base.py:65-68): mode bit 0 = empirical_kl (log q(z) - log p_y(z)), bit 1 = fixed_var (unit variance, h
 * holds the means only), and a re-parametrisation dimension other than 1: h [B][S][2 D] ([B][S][D] with fixed_var) = torch.chunk on a
 * dimension with S entries in front of it; eps, z and the gathered prior rows [B][S * D]. */
int otvae_gaussian_prior_cond_ex_fwd(const float* h, const float* eps, const float* prior_mean, const float* prior_log_std, int B, int S,
                                     int D, float coeff, int mode, float* z, float* loss, void* stream);
int otvae_gaussian_prior_cond_ex_bwd(const float* h, const float* eps, const float* prior_mean, const float* prior_log_std,
                                     const float* gz, const float* gloss, int B, int S, int D, float coeff, int mode, float* gh,
                                     float* g_prior_mean, float* g_prior_log_std, void* stream);

/* ---- VAE.nelbo reduction (model/vae.py:158-176) ----------------------------------------------------------- */
/* out[3] = {total, recon, prior}: recon = mean((pred-target)^2) over numel entries, prior = mean(prior_loss[0..B))/chw.
 * B = the number of entries of prior_loss: one per latent the prior saw = expansion * batch (model/vae.py:165-169,
 * utils/__init__.py:154-175), independent of numel.  ws: double[otvae_nelbo_ws()] scratch. */
int otvae_nelbo_ws(void);
int otvae_nelbo_fwd(const float* pred, const float* target, int64_t numel, const float* prior_loss, int B,
                    float chw, double* ws, float* out, void* stream);
/* gout[3] = upstream gradient of out: gpred = (gout[0]+gout[1]) * 2*(pred-target)/numel ;
 * gprior[B] = (gout[0]+gout[2])/(B*chw).  gout NULL = {1,0,0}. */
int otvae_nelbo_bwd(const float* pred, const float* target, int64_t numel, int B, float chw,
                    const float* gout, float* gpred, float* gprior, void* stream);

/* ---- ConvLayer activations other than ReLU, and equalized_lr (networks/cnn.py:114-118,128-147,186-188) -----------
 * kind: 0 identity, 1 ReLU, 2 LeakyReLU(0.2), 3 SELU, 4 GELU (erf form), 5 SiLU.  x, out, ga, gv: [M][C] channels-last.
 * The fused convolution kernels carry ReLU only; these run unfused around them (csrc/activation.hip). */
/* out = act(x * scale[c] + shift[c]); scale == shift == NULL: out = act(x) */
int otvae_bn_act_fwd(const float* x, const float* scale, const float* shift, int kind, int64_t M, int C, float* out,
                     void* stream);
/* blocks (= partial sums per channel) otvae_bn_act_bwd uses for M rows */
int otvae_bn_act_bwd_parts(int64_t M);
/* gv = ga * act'(x * scale + shift).  With mean / invstd (training-mode BatchNorm in front of the activation) also the
 * BatchNorm-backward sums per block, partial[2][C][P] (fp64) = {sum gv, sum gv * (x - mean) * invstd}: the layout
 * otvae_bn_bwd_finalize reads (its CsPad = C). */
int otvae_bn_act_bwd(const float* ga, const float* x, const float* scale, const float* shift, const float* mean,
                     const float* invstd, int kind, int64_t M, int C, float* gv, double* partial, void* stream);
/* dst[i] = alpha * src[i], n contiguous floats (weight * conv_scale * lr_mult, bias * lr_mult and their gradients) */
int otvae_scale_f32(const float* src, float alpha, int64_t n, float* dst, void* stream);

/* ---- grouped / dilated ConvLayer (networks/cnn.py:66-67,103-104: nn.Conv2d(in, out, k, stride, padding, dilation, groups)).
 * The layer's weight w [Cout][Cin / groups][KH][KW] (contiguous, as nn.Conv2d stores it) is expanded into the dense weight of the
 * convolution entry points, dense [(KH-1) dil + 1][(KW-1) dil + 1][Cin][Cout] (HWIO memory): zeros between groups and in the holes
 * of the dilation.  _bwd gathers the dense weight's gradient back onto w's layout.  The expanded kernel may span at most 7 x 7. */
int otvae_weight_expand_fwd(const float* w, int Cout, int Cin, int groups, int KH, int KW, int dilation, float* dense, void* stream);
int otvae_weight_expand_bwd(const float* gdense, int Cout, int Cin, int groups, int KH, int KW, int dilation, float* gw, void* stream);

/* ---- GroupNorm / InstanceNorm2d in front of a ConvLayer's activation (networks/cnn.py:121-125: nn.GroupNorm(div_sqrt(C // groups), C),
 * nn.InstanceNorm2d(C) = G == C without gamma / beta).  x, out, ga, dx [N][HW][C] channels-last; statistics per (sample, group) over
 * the group's C / G channels and all HW positions (biased variance, eps inside the root); kind: the activation codes above. */
/* out = act(xhat * gamma + beta) (gamma == beta == NULL: act(xhat)); mean, rstd [N][G] are kept for the backward */
int otvae_group_norm_act_fwd(const float* x, const float* gamma, const float* beta, int N, int HW, int C, int G, float eps, int kind,
                             float* out, float* mean, float* rstd, void* stream);
/* dx of the same chain from ga = dL/d out; with gamma also the per-sample parameter partials pgamma, pbeta [N][C]
 * (d gamma = otvae_colsum_f32(pgamma), d beta likewise) */
int otvae_group_norm_act_bwd(const float* ga, const float* x, const float* gamma, const float* beta, const float* mean,
                             const float* rstd, int N, int HW, int C, int G, int kind, float* dx, float* pgamma, float* pbeta,
                             void* stream);
/* dst[c] = sum_r src[r][c] in fixed order, src [R][C] */
int otvae_colsum_f32(const float* src, int R, int C, float* dst, void* stream);
/* Backward pass of an embedding lookup (nn.Embedding: the ViT's class token, networks/vit.py:167,203; the class rows of
 * ConditionalGaussianPrior, prior/conditional_gaussian.py:76-77): gw[k][:] = sum of g[b][:] over the b with idx[b] == k, in
 * increasing b.  g [B][d], idx [B] int64, gw [K][d] (every row written). */
int otvae_embedding_bwd(const float* g, const int64_t* idx, int B, int K, int d, float* gw, void* stream);

/* ---- FiLM conditioning (`additional_embed`) and Dropout2d of ConvLayer (networks/cnn.py:112-118,160-192); x, out, g [N][HW][C] ------- */
/* out = x * scale[n][c] + bias[n][c] (scale / bias [N][C]: the two Linear projections of the activated embedding) */
int otvae_film_fwd(const float* x, const float* scale, const float* bias, int N, int HW, int C, float* out, void* stream);
/* gx = g * scale; gscale[n][c] = sum_hw g x; gbias[n][c] = sum_hw g */
int otvae_film_bwd(const float* g, const float* x, const float* scale, int N, int HW, int C, float* gx, float* gscale, float* gbias,
                   void* stream);
/* nn.Dropout2d(p): y = keep(n, c) ? x / (1 - p) : 0 with keep = hash(call key, n, c); key = device int64[2] {seed, call counter},
 * used[0] <- the call key (the backward and otvae_dropout2d_mask recompute the mask from it; no mask tensor) */
int otvae_dropout2d_fwd(const float* x, int N, int HW, int C, float p, const int64_t* key, int stream_id, float* y, int64_t* used,
                        void* stream);
int otvae_dropout2d_bwd(const float* gy, int N, int HW, int C, float p, const int64_t* used, float* gx, void* stream);
int otvae_dropout2d_mask(int N, int C, float p, const int64_t* used, uint8_t* keep, void* stream);

/* ---- GaussianModel(update_with_autograd=True): log-density under N(mean, L L^T) / N(mean, diag(sigma^2))
 * (ot/distribution_models/gaussian_model.py:52-55,76-93,125-128; torch.distributions.MultivariateNormal(scale_tril=) /
 * Independent(Normal) in the reference).  fp64, x / y / qg [nb][B][D], mean [nb][D], L [nb][D][D] lower triangular with a
 * positive diagonal (diag != 0: sigma [nb][D]), D <= 128. */
/* y = L^-1 (x - mean); lp[nb][B] = -|y|^2 / 2 - sum_i log L_ii - D/2 log(2 pi) */
int otvae_mvn_logprob_fwd(const double* x, const double* mean, const double* L, int nb, int B, int D, int diag, double* y,
                          double* lp, void* stream);
/* qg_b = g_b L^-T y_b (diag: g_b y_b / sigma): d mean = sum_b qg_b, d x_b = -qg_b,
 * d L = tril(sum_b qg_b y_b^T) - (sum_b g_b) diag(1 / L_ii), formed by the caller with otvae_gemm_f64 */
int otvae_mvn_logprob_bwd(const double* g, const double* y, const double* L, int nb, int B, int D, int diag, double* qg,
                          void* stream);

/* ---- Adam (model/vae.py:148-151; torch.optim.Adam defaults) over one flat buffer --------------------------- */
/* hyper (device): float[4] = {lr, beta1, beta2, eps}; step (device int32) is the 1-based count of THIS update
 * (incremented by otvae_step_begin). grad_scale multiplies g first (1/world_size for data-parallel means). */
/* dst[i][0..n[i]) = src[i][0..n[i]) for `count` contiguous fp32 ranges in one launch per 32 (pointer tables on the HOST, passed to the
 * kernel by value): the gradients autograd left in p.grad (embeddings, learned tokens) into their slots of the flat gradient buffer
 * that otvae_adam_step reads -- what the reference's optimizer finds in p.grad (model/vae.py:148-151). */
int otvae_copy_batched(int count, const float* const* src, float* const* dst, const int64_t* n, void* stream);

int otvae_step_begin(int32_t* step, void* stream);
int otvae_adam_step(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper,
                    const int32_t* step, float grad_scale, void* stream);
/* the same update with the gradient scale read from device memory (what otvae_grad_clip_coef leaves in out[0]) */
int otvae_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper,
                        const int32_t* step, const float* grad_scale_dev, void* stream);
/* Guarded update: what makes a bad step survivable inside a captured training step (the reference leans on Lightning's
 * loop to stop on a NaN loss, configs/ddp.yaml:1-5; a hipGraph replay has no host in the loop).  The update is applied only
 * when every watched device scalar is finite: watch_loss[0] (the step's loss; a starved Sinkhorn solve poisons it with
 * NaN), and grad_scale_dev[0..1] = {scale, |g|} when given (what otvae_grad_clip_coef leaves; with max_norm <= 0 it only
 * reports the norm).  grad_scale_dev NULL: the host value grad_scale is used.  A skipped step leaves p, m, v unchanged, takes
 * *step back by one, restores state[n_state] from backup (below), and counts itself: guard[0] += 1, guard[1] = the step number
 * that was skipped (device int32[2]). */
int otvae_adam_step_guarded(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, int32_t* step,
                            float grad_scale, const float* grad_scale_dev, const float* watch_loss, int32_t* guard,
                            float* state, const float* backup, int64_t n_state, void* stream);
/* The superset entry: otvae_adam_step_guarded's arguments (guard / watch_loss / state may be NULL: unguarded) plus the parameter
 * moving average of the reference's `ema_decay` option (model/base.py:99,153-190; kept there by the third-party torch_ema package,
 * requirements.txt:10, unpinned): ema_shadow[n] -= (1 - d) (ema_shadow - p_new), d = min(ema_decay, (1 + t) / (10 + t)) with t the
 * device step counter (one update per accepted optimizer step), in the optimizer's own pass.  A refused step leaves the shadow alone. */
int otvae_adam_step_ema(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, int32_t* step,
                        float grad_scale, const float* grad_scale_dev, const float* watch_loss, int32_t* guard, float* state,
                        const float* backup, int64_t n_state, float* ema_shadow, double ema_decay, void* stream);
/* The same moving-average update alone, with the caller's effective decay (a stock optimizer stepped by the reference's own loop) */
int otvae_ema_update(float* shadow, const float* p, int64_t n, double decay, void* stream);
/* The running buffers of a guarded step.  A NaN that reaches a BatchNorm does not stay one (the next ReLU maps the NaN-normalised
 * tensor to zeros), so later layers would fold finite but meaningless batch statistics into their running buffers:
 * otvae_step_begin_guarded increments *step like otvae_step_begin AND copies state[n_state] (every running buffer of the model,
 * one flat fp32 range) to backup; a refused step (otvae_adam_step_guarded) copies it back.  n_state 0: no such buffers. */
int otvae_step_begin_guarded(int32_t* step, const float* state, float* backup, int64_t n_state, void* stream);
/* Round 4: counter + (n > 0) the guard's backup + the zeroing of `zero_words` int64 words (the BatchNorm statistic slots the step's
 * kernels add into; 16-byte aligned, an even count, may be 0) as ONE launch.  otvae_zero_words: the zeroing alone. */
int otvae_step_begin_slots(int32_t* step, const float* state, float* backup, int64_t n, void* zero, int64_t zero_words, void* stream);
int otvae_zero_words(void* zero, int64_t zero_words, void* stream);

/* ---- global-norm gradient clipping (configs/ddp.yaml:4 `gradient_clip_val: 1.0`, applied by Lightning through
 * torch.nn.utils.clip_grad_norm_) over the flat gradient buffer -------------------------------------------------- */
/* g[n] holds the gradient SUM over ranks, grad_scale = 1/world_size.  norm = |g * grad_scale|_2 (fp64 accumulation);
 * out[0] = grad_scale * min(1, max_norm / (norm + 1e-6)) = the scale otvae_adam_step_dev applies to g; out[1] = norm.
 * max_norm <= 0: no clipping (out[0] = grad_scale).  ws: otvae_grad_clip_ws() doubles. */
int otvae_grad_clip_ws(void);
int otvae_grad_clip_coef(const float* g, int64_t n, float grad_scale, float max_norm, double* ws, float* out,
                         void* stream);

/* ---- generic convolution: any stride, square footprints up to 32 x 32 (the fallback behind the tuned kernels, which take strides 1 / 2
 * and footprints up to 7 x 7): what `ConvLayer(down_sample=s)` makes for s >= 4 -- a (2 s) x (2 s) kernel with stride s, e.g. the
 * 8 x 8 / stride 4 layers of CNN(scaling_factor=4) (networks/cnn.py:98-101,605-621).  Plain direct convolutions (no BatchNorm / activation
 * fusion, geom->up must be 1), NHWC activations, HWIO weights, fp32, deterministic. ---- */
int otvae_conv_generic_fwd(const otvae_conv_geom* geom, const float* x, const float* w_hwio, const float* bias, float* y, void* stream);
int otvae_conv_generic_bwd_data(const otvae_conv_geom* geom, const float* gy, const float* w_hwio, float* gx, void* stream);
int64_t otvae_conv_generic_bwd_weight_ws(const otvae_conv_geom* geom, int has_bias); /* floats */
int otvae_conv_generic_bwd_weight(const otvae_conv_geom* geom, const float* x, const float* gy, int has_bias, float* ws,
                                  float* gw_hwio, float* gb, void* stream);

/* ---- standard-normal draws (prior/gaussian.py:93 `q.rsample()`, the prior samples of the minibatch-OT prior) ---------- */
/* out[n] ~ N(0, 1) from a counter-based hash of (key[0] = seed, key[1] = call counter, stream_id, element index): the
 * values do not depend on the launch shape.  key = device int64[3] {seed, counter, 0}; advance != 0: the call's last
 * workgroup increments the counter, so a captured step draws fresh noise on every replay without host work. */
int otvae_normal_fill(float* out, int64_t n, int64_t* key, int stream_id, int advance, void* stream);

/* ---- sinkhorn_log (ot/w2_utils.py:276-319) ---------------------------------------------------------------- */
/* a[nb][N], b[nb][M], C[nb][N][M] -> pi[nb][N][M] (may alias no input), u[nb][N], v[nb][M] potentials.
 * dtype: 0 = fp32, 1 = fp64.  ws: bytes from otvae_sinkhorn_ws.  Stops all problems at the first iteration where
 * min over the batch of (|du|_1+|dv|_1) < threshold, evaluated on the device (no host sync); threshold <= 0
 * runs exactly max_iter iterations.  iters_done (device int32, may be NULL). */
int64_t otvae_sinkhorn_ws(int dtype, int nb, int N, int M);
int otvae_sinkhorn_log(int dtype, const void* a, const void* b, const void* C, int nb, int N, int M,
                       double reg, int max_iter, double threshold, void* ws,
                       void* pi, void* u, void* v, int32_t* iters_done, void* stream);
/* The reference's sinkhorn_log is plain torch arithmetic: autograd differentiates it through every iteration with respect to
 * a, b and C (ot/w2_utils.py:301-319).  These two entry points are that derivative.  otvae_sinkhorn_log_tape: the same solve (one
 * launch per half-iteration, identical arithmetic and stopping rule) that also keeps what the reverse sweep needs in `tape`
 * (otvae_sinkhorn_tape_bytes: Cr, Cr^T, log marginals, the potentials of every iteration, the iteration count).
 * otvae_sinkhorn_log_bwd: gpi[nb][N][M] = dL/dpi -> gC[nb][N][M], ga[nb][N], gb[nb][M] (each may be NULL; ga / gb need a / b),
 * ws: otvae_sinkhorn_bwd_ws bytes.  The iteration count the forward stopped at is read on the device (no host sync). */
int64_t otvae_sinkhorn_tape_bytes(int dtype, int nb, int N, int M, int max_iter);
int64_t otvae_sinkhorn_bwd_ws(int dtype, int nb, int N, int M, int max_iter);
int otvae_sinkhorn_log_tape(int dtype, const void* a, const void* b, const void* C, int nb, int N, int M, double reg,
                            int max_iter, double threshold, void* tape, void* pi, void* u, void* v, int32_t* iters_done,
                            void* stream);
int otvae_sinkhorn_log_bwd(int dtype, const void* gpi, const void* pi, const void* a, const void* b, int nb, int N, int M,
                           double reg, int max_iter, void* tape, void* ws, void* gC, void* ga, void* gb, void* stream);
/* The same solve on C / max(C) per problem, the `cost_matrix / max_per_mat` of batch_ot_gmm (ot/w2_utils.py:265-266), without
 * a pass that finds the maximum or divides: pmax[nb][P] are partial maxima of each problem's C (otvae_sqdist_max leaves them),
 * reduced inside the solver's first kernel; cmax[nb] (may be NULL) receives the maxima.  a / b may be NULL in both entry points:
 * uniform marginals 1/N, 1/M.
 * A persistent solve whose bounded wait ran out (OTVAE_SK_SPIN_LIMIT polls, default 2^22) fills pi, u, v with NaN and reports
 * iters_done = -1: a starved solve cannot pass for a result. */
int otvae_sinkhorn_log_normalized(int dtype, const void* a, const void* b, const void* C, const void* pmax, int P, int nb,
                                  int N, int M, double reg, int max_iter, double threshold, void* ws, void* pi, void* u, void* v,
                                  void* cmax, int32_t* iters_done, void* stream);
/* The minibatch-OT prior's forward (configs[2]/[3]: BASELINE.json; the arithmetic of ot/w2_utils.py:265-269 on
 * C_ij = |z_i - y_j|^2 with uniform marginals) in one call: C[N][M], pi[N][M], u[N], v[M], cost[1] = sum C * pi, cmax[1]
 * (may be NULL).  z[N][D], y[M][D].  cost[cost_rep] = loss_scale * sum C * pi, the same value cost_rep times (the Prior
 * contract returns one loss entry per sample, prior/base.py:74-78; loss_scale = loss_coeff x annealing).
 * ws: otvae_sinkhorn_prior_ws bytes. */
int64_t otvae_sinkhorn_prior_ws(int dtype, int N, int M);
int otvae_sinkhorn_prior_fwd(int dtype, const void* z, const void* y, int N, int M, int D, double reg, int max_iter,
                             double threshold, double loss_scale, int cost_rep, void* ws, void* C, void* pi, void* u, void* v,
                             void* cost, void* cmax, int32_t* iters_done, void* stream);
/* cost[nb] = sum_ij C*pi (fp64 accumulate, fixed order), out dtype = dtype; ws: double[nb*256] */
int otvae_ot_cost(int dtype, const void* C, const void* pi, int nb, int N, int M, double* ws, void* cost, void* stream);
/* pairwise squared euclidean cost C[nb][N][M] = |x_i - y_j|^2, x[nb][N][D], y[nb][M][D] */
int otvae_sqdist(int dtype, const void* x, const void* y, int nb, int N, int M, int D, void* C, void* stream);
/* the same, also leaving the maximum of every output tile in pmax[nb][otvae_sqdist_max_parts(dtype, N, M)] */
int otvae_sqdist_max_parts(int dtype, int N, int M);
int otvae_sqdist_max(int dtype, const void* x, const void* y, int nb, int N, int M, int D, void* C, void* pmax, void* stream);
/* Gradient of scale * sum_ij C_ij pi_ij with respect to z for C_ij = |z_i - y_j|^2 and a fixed plan (the minibatch OT prior's
 * backward, SinkhornPrior): gz[i][d] = 2 scale (sum_q g[q]) sum_j pi_ij (z_id - y_jd); z [N][D], y [M][D], pi [N][M]; g[ng] =
 * the upstream gradients of the ng replicas of the cost that the forward handed out (device memory); gadd[N][D] (may be
 * NULL) is added to the result (the gradient reaching z through its other consumer, the decoder).  fp32 runs on the
 * matrix cores (16 x 16 x 4 MFMA tiles, fixed summation order). */
int otvae_ot_cost_grad(int dtype, const void* z, const void* y, const void* pi, const void* g, int ng, double scale,
                       const void* gadd, int N, int M, int D, void* gz, void* stream);

/* ---- Gaussian W2 with empirical covariance as a loss term (GaussianW2Prior; BASELINE north_star) ------------ */
/* Forward tail of L = w2_gaussian(mean_cov(_stats(z)), N(mut, covt)) (ot/w2_utils.py:40-80, ot/matrix_utils.py:145-158,
 * gaussian_model.py:144-157): lam[D], vt[D][D] = eigenvalues / eigenvector rows of M = covt^1/2 cov covt^1/2 (otvae_eigh_fn,
 * fn 3).  mut == NULL: zero mean; covt == NULL: identity (then M = cov and the make_pd shift of the 'spd' validation,
 * w2_utils.py:661-669, is applied to cov and its spectrum alike).  loss[rep] = scale * L (fp32, the same value rep times: one
 * entry per sample, prior/base.py:74-78);  q[D][D] = diag(lam^-1/4) vt, so that M^-1/2 = q^T q. */
int otvae_w2_prior_tail(const double* mu, const double* mut, const double* cov, const double* covt, const double* lam,
                        const double* vt, int D, double scale, int rep, float* loss, double* q, void* stream);
/* Backward: gz[i][:] = gadd[i][:] + (2 scale sum(g) / B) [ (mu - mut) + (z_i - mu) - W (z_i - mu) ], W[D][D] = covt^1/2 M^-1/2
 * covt^1/2 (symmetric) -- the closed form of the reference's autograd through eigh-based sqrtm.  z, gadd (may be NULL), gz:
 * [B][D] fp32 (dtype 0) or fp64 (1); g[ng] fp32 upstream gradients of the loss replicas. */
int otvae_w2_prior_bwd(int dtype, const void* z, int B, int D, const double* mu, const double* mut, const double* W,
                       const float* g, int ng, double scale, const void* gadd, void* gz, void* stream);

/* ---- GaussianModel statistics (ot/distribution_models/gaussian_model.py:99-108,144-157) ------------------- */
/* samples [nb][B][D] (in_dtype 0=fp32,1=fp64) -> fp64 sum_x[nb][D], sum_xx[nb][D][D] (diag: [nb][D]),
 * accumulated into the running buffers:  run = decay<0 ? run + new : run*decay + new*(1-decay)
 * (utils/__init__.py:204-206); n_obs[nb] likewise with new = B.  With accumulate==0 the raw batch statistics are
 * written instead (the all-reduce path reduces them before the EMA). */
int64_t otvae_gauss_stats_ws(int nb, int B, int D, int diag); /* bytes */
int otvae_gauss_stats(int in_dtype, const void* samples, int nb, int B, int D, int diag, int accumulate,
                      double decay, double* ws, double* n_obs, double* sum_x, double* sum_xx, void* stream);
/* mean_cov (ot/matrix_utils.py:145-158): mean = sum/n, cov = sum_xx/n - mean mean^T */
int otvae_mean_cov(const double* n_obs, const double* sum_x, const double* sum_xx, int nb, int D, int diag,
                   double* mean, double* cov, void* stream);

/* ---- symmetric eigen-decomposition based matrix functions (ot/matrix_utils.py:37-109) --------------------- */
/* A[nb][D][D] fp64 symmetric (lower triangle is read, like eigh(UPLO='L')).  fn: 0 = none (eigvals only),
 * 1 = sqrtm, 2 = invsqrtm: out[nb][D][D] = V f(lambda) V^T; 3 = eigenvectors: out[nb][k][:] is the unit eigenvector of
 * eigvals[nb][k] (so that callers form several functions of one matrix from a single decomposition).  eigvals[nb][D]
 * are not sorted.  D <= 128: one workgroup per matrix in LDS; 128 < D <= 2048: block Jacobi.
 * ws: bytes from otvae_eigh_ws. */
int64_t otvae_eigh_ws(int nb, int D);
int64_t otvae_eigh_onesided_ws(int nb, int D); /* part of otvae_eigh_ws for D <= 128: the one-sided solver's share */
int64_t otvae_eigh_block_onesided_ws(int nb, int D); /* part of otvae_eigh_ws for 128 < D <= 1024 */
int otvae_eigh_fn(const double* A, int nb, int D, int fn, double* out, double* eigvals, void* ws, void* stream);
/* otvae_eigh_fn started from an orthonormal basis the caller already has (Vinit[nb][D][D], row k = vector k; e.g. the eigenvectors
 * of the previous training step's latent covariance, prior/gaussian_w2.py): the iteration begins with the nearly orthogonal columns
 * A v_k and needs 2-4 sweeps instead of ~9; the result is the same decomposition.  A must be symmetric in both triangles; g0:
 * scratch of nb*D*D doubles.  warm (nullable): device int read by the kernels, 0 = "Vinit holds nothing yet" (run cold): lets a
 * captured step decide per replay.  Vinit == NULL, D > 128: the cold solver. */
int otvae_eigh_fn_warm(const double* A, const double* Vinit, const int* warm, int nb, int D, int fn, double* out, double* eigvals,
                       void* ws, double* g0, void* stream);
/* make_psd (ot/matrix_utils.py:123-142) without host sync: shift_b = (any_b lambda_min_b <= thr ? 1 : 0) *
 * (|min(lambda_min_b,0)| + (strict ? 1e-8 : 0)), A_b += shift_b * I.  cond_any: 1 = apply only if some matrix of
 * the batch fails the test (w2_utils.py:667-669), 0 = always (gaussian_model.py:204-214). */
int otvae_make_psd(double* A, const double* eigvals, int nb, int D, int strict, int cond_any, void* stream);
/* Lower Cholesky factor L[nb][D][D] of A[nb][D][D] (fp64, lower triangle read): the factor torch.distributions.MultivariateNormal
 * samples with (the reference's stochastic apply_transport, ot/w2_utils.py:521-525).  info[nb] (may be NULL): 0, or 1 + the index
 * of the first non-positive pivot (then L holds NaN from that column on).  L must not alias A. */
int otvae_cholesky(const double* A, int nb, int D, double* L, int* info, void* stream);
/* C[nb][m][n] = alpha * op(A) op(B) + beta*C, fp64, row-major; transX: 0 = N, 1 = T. bcast flags: operand has
 * batch stride 0 */
int otvae_gemm_f64(int transA, int transB, int nb, int m, int n, int k, double alpha, const double* A, int a_bcast,
                   const double* B, int b_bcast, double beta, double* C, void* stream);
/* the same in fp32: the small weighted sums of the mixture / codebook models (weights @ atoms, probs^T @ samples:
 * ot/distribution_models/base.py:241-251, codebook_model.py:145-148, ot/transport/discrete_transport.py:70-76) */
int otvae_gemm_f32(int transA, int transB, int nb, int m, int n, int k, float alpha, const float* A, int a_bcast,
                   const float* B, int b_bcast, float beta, float* C, void* stream);
/* y[rows][K] = softmax_k(scale * x) and its backward gx = scale * y o (gy - sum_k y gy): the assignment distributions
 * softmax(energy / temperature) (base.py:216-224) and F.gumbel_softmax (:234-235); dtype 0 = fp32, 1 = fp64 */
int otvae_softmax_rows(int dtype, const void* x, int64_t rows, int K, double scale, void* y, void* stream);
int otvae_softmax_rows_bwd(int dtype, const void* y, const void* gy, int64_t rows, int K, double scale, void* gx, void* stream);
/* out[rows] = log sum_k exp(x[r][k]) (torch.logsumexp over the last dimension: the mixture log-density read-out of
 * GaussianMixtureModel, gaussian_model.py:129-132 on a MixtureSameFamily) and its backward gx[r][k] = g[r] exp(x[r][k] - lse[r]) */
int otvae_lse_rows(int dtype, const void* x, int64_t rows, int K, void* out, void* stream);
int otvae_lse_rows_bwd(int dtype, const void* x, const void* lse, const void* g, int64_t rows, int K, void* gx, void* stream);
/* w2_gaussian tail (ot/w2_utils.py:78-80): out[nb] = |ms-mt|^2 + tr(cs + ct - 2*sqrt_mix) */
int otvae_w2_tail(const double* ms, const double* mt, const double* cs, const double* ct, const double* sqrt_mix,
                  int nb, int D, double* out, void* stream);
/* w2_gaussian (ot/w2_utils.py:40-80) and the non-stochastic full-matrix operator of eq. 17 (compute_transport_operators ->
 * _compute_transport_full_mat, :756-769) of the same nb pairs of Gaussians in one call, from the spectra of the two covariances
 * (lam_x[nb][D], vt_x[nb][D][D] with row k = eigenvector k, as otvae_eigh_fn(fn = 3) returns them), including the reference's
 * 'spd' argument validation (:661-669: with make_pd a strictly positive shift |min(lambda_min, 0)| + 1e-8 on a side whose batch
 * holds a matrix that is not positive definite).  w2[nb], T[nb][D][D] = (1 - pg_star) Cs^-1/2 (Cs^1/2 Ct Cs^1/2)^1/2 Cs^-1/2 + pg_star I.
 * No host synchronisation inside: flags[3] (device ints) = {a source covariance is not positive definite, a target covariance is
 * not, Ct^1/2 Cs Ct^1/2 is not symmetric} for the caller to turn into the reference's ValueErrors.  ws: otvae_w2_transport_ws(nb, D)
 * bytes; eigh_ws: otvae_eigh_ws(2 * nb, D) bytes. */
int64_t otvae_w2_transport_ws(int nb, int D);
int otvae_w2_transport(const double* ms, const double* mt, const double* cs, const double* ct, const double* lam_s, const double* vt_s,
                       const double* lam_t, const double* vt_t, int nb, int D, double pg_star, int make_pd, void* ws, void* eigh_ws,
                       double* w2, double* T, int* flags, void* stream);
/* apply_transport (ot/w2_utils.py:517-520): y[nb][B][D] = T[nb] (x - ms) + mt ; x dtype 0/1, y same dtype */
int otvae_apply_transport(int dtype, const void* x, const double* ms, const double* mt, const double* T,
                          int nb, int B, int D, void* y, void* stream);

/* ---- CodebookModel.energy/assign 'argmax'/predict (ot/distribution_models/codebook_model.py:150-160,
 *      base.py:216-233): x[nb][B][d], codebook[nb][K][d] -> idx[nb][B] (int64), enc[nb][B][d] = codebook[idx] */
int otvae_codebook_assign(const float* x, const float* codebook, int nb, int B, int K, int d, float temperature,
                          int64_t* idx, float* enc, void* stream);
/* The assignment distribution itself (base.py:216-224): probs[nb][B][K] = softmax_k(energy/temperature), and
 * (nullable) entropy[nb][B] = -sum_k p log p, which the CodebookPrior 'kl' loss uses (prior/codebook.py:81-82). */
int otvae_codebook_probs(const float* x, const float* codebook, int nb, int B, int K, int d, float temperature,
                         float* probs, float* entropy, void* stream);
/* its backward with respect to the samples (the reference's codebook is a frozen parameter unless update_with_autograd,
 * codebook_model.py:84-86): gprobs[nb][B][K] / gentropy[nb][B] are the upstream gradients (either may be NULL), probs the
 * forward's output; gx[nb][B][d].  K <= 4096. */
int otvae_codebook_probs_bwd(const float* x, const float* codebook, const float* probs, const float* gprobs,
                             const float* gentropy, int nb, int B, int K, int d, float temperature, float* gx, void* stream);

/* The same with the gradient of the atoms too: CodebookModel(update_with_autograd=True) trains its codebook as a parameter
 * (reference ot/distribution_models/codebook_model.py:89 `requires_grad=self.update_with_autograd`; energies :158-160 under
 * torch.autograd).  coef_ws: [nb][B][K] floats of workspace; gc: [nb][K][d], summed over the samples in a fixed order. */
int otvae_codebook_probs_bwd_atoms(const float* x, const float* codebook, const float* probs, const float* gprobs,
                                   const float* gentropy, int nb, int B, int K, int d, float temperature, float* gx,
                                   float* coef_ws, float* gc, void* stream);

/* CodebookModel.energy for the metrics besides the hot path's euclidean p = 2 (ot/distribution_models/codebook_model.py:155-168):
 * metric 0: 1 / (cdist_p(x, c) + 1e-8) for any p > 0; metric 1 ('cosine'): |x . c| / ((sum |x|^p)(sum |c|^p) + 1e-8)^(1/p).
 * E [nb][B][K], raw (MixtureMixin.assign applies topk and the temperature afterwards, base.py:216-224).  _bwd: gx [nb][B][d] and /
 * or gc [nb][K][d] (NULL = not wanted) from gE, summed in a fixed order. */
int otvae_codebook_energy(const float* x, const float* codebook, int nb, int B, int K, int d, int metric, float p, float* E,
                          void* stream);
int otvae_codebook_energy_bwd(const float* x, const float* codebook, const float* gE, int nb, int B, int K, int d, int metric,
                              float p, float* gx, float* gc, void* stream);
/* k-means sufficient statistics for one-hot ('argmax') assignments (MixtureMixin.kmean_iteration, base.py:241-252):
 * counts[nb][K] = number of samples per atom, sums[nb][K][d] = their sum, members added in increasing sample order. */
int otvae_codebook_kmeans(const float* x, const int64_t* idx, int nb, int B, int K, int d, float* counts, float* sums,
                          void* stream);

/* ---- Gaussian mixture (diagonal covariances): GaussianMixtureModel.energy, gassian_mixture_model.py:91-99 ------------
 * energy[nb][B][K] = log N(x_b; mean_k, diag var_k) + log w_k for x [nb][B][d], mean / var [nb][K][d], logw [nb][K];
 * dtype 0 = fp32, 1 = fp64 (all tensors).  The assignment (softmax / argmax over K) and the weighted sufficient
 * statistics that follow are [B][K]-sized library reductions on the caller's side. */
int otvae_gmm_diag_energy(int dtype, const void* x, const void* mean, const void* var, const void* logw, int nb, int B, int K,
                          int d, void* energy, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OTVAE_H */
