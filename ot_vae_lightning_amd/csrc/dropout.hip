// Element-wise dropout with the counter-based masks of dropout_hash.h, optionally fused with the ReLU in front of it:
//   y = keep(row, col) ? act(x) / (1-p) : 0,   act = ReLU or identity
// -- the "dropout(activation(linear1(x)))" in the feed-forward half of a training-mode nn.TransformerEncoderLayer and the
// embedding dropout of PositionalEmbedding (reference networks/vit.py:54-58,157-172).  No mask tensor exists: the backward
// kernel recomputes keep from the call key the forward left in `used`, and reads x for the ReLU gate.
#include "common.h"
#include "dropout_hash.h"

template <bool RELU, bool BWD>
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, const float* __restrict__ gy, int64_t total, int D,
                                                      uint32_t thresh, float inv_keep, const int64_t* __restrict__ key, int stream_id,
                                                      int64_t* __restrict__ used, float* __restrict__ out) {
    uint64_t ck;
    if constexpr (BWD) {
        ck = (uint64_t)key[0];
    } else {
        ck = call_key(key, stream_id);
        if (blockIdx.x == 0 && threadIdx.x == 0) used[0] = (int64_t)ck;
    }
    // D % 4 == 0 (checked by the host): a thread's four consecutive elements share a row
    const int64_t n4 = total >> 2;
    for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int64_t e = i << 2;
        const int64_t row = e / D;
        const int col = (int)(e - row * D);
        const uint32_t rh = row_hash(ck, (uint32_t)row);
        const float4 xv = reinterpret_cast<const float4*>(x)[i];
        float4 sv = BWD ? reinterpret_cast<const float4*>(gy)[i] : xv;  // what is scaled: gy (backward) or x (forward)
        float r[4] = {sv.x, sv.y, sv.z, sv.w};
        const float gate[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            bool on = keep_pair(rh, col + k, thresh);
            if constexpr (RELU) on = on && gate[k] > 0.f;
            r[k] = on ? r[k] * inv_keep : 0.f;
        }
        reinterpret_cast<float4*>(out)[i] = make_float4(r[0], r[1], r[2], r[3]);
    }
}

static int dropout_check(const char* who, int64_t rows, int D, float p) {
    OTVAE_REQUIRE(rows > 0 && D > 0 && rows < ((int64_t)1 << 32), "%s: bad sizes (rows must stay below 2^32)", who);
    OTVAE_REQUIRE(D % 4 == 0, "%s: the row width must be a multiple of 4", who);
    OTVAE_REQUIRE(p >= 0.f && p < 1.f, "%s: dropout probability must be in [0, 1)", who);
    return OTVAE_OK;
}

static int dropout_grid(int64_t total) {
    const int64_t b = cdiv(total >> 2, (int64_t)256);
    return (int)(b < 4096 ? b : 4096);
}

extern "C" int otvae_dropout_fwd(const float* x, int64_t rows, int D, int relu, float p, const int64_t* key, int stream_id, float* y,
                                 int64_t* used, void* stream) {
    OTVAE_REQUIRE(x && y && key && used && stream_id >= 0 && stream_id < 4095, "otvae_dropout_fwd: bad argument");
    int rc = dropout_check("otvae_dropout_fwd", rows, D, p);
    if (rc) return rc;
    const int64_t total = rows * D;
    const uint32_t th = dropout_threshold(p);
    const float ik = 1.f / (1.f - p);
    hipStream_t st = (hipStream_t)stream;
    if (relu) dropout_kernel<true, false><<<dropout_grid(total), 256, 0, st>>>(x, nullptr, total, D, th, ik, key, stream_id, used, y);
    else dropout_kernel<false, false><<<dropout_grid(total), 256, 0, st>>>(x, nullptr, total, D, th, ik, key, stream_id, used, y);
    OTVAE_CHECK_LAUNCH("otvae_dropout_fwd");
    return OTVAE_OK;
}

extern "C" int otvae_dropout_bwd(const float* x, const float* gy, int64_t rows, int D, int relu, float p, const int64_t* used,
                                 float* gx, void* stream) {
    OTVAE_REQUIRE(x && gy && gx && used, "otvae_dropout_bwd: bad argument");
    int rc = dropout_check("otvae_dropout_bwd", rows, D, p);
    if (rc) return rc;
    const int64_t total = rows * D;
    const uint32_t th = dropout_threshold(p);
    const float ik = 1.f / (1.f - p);
    hipStream_t st = (hipStream_t)stream;
    if (relu) dropout_kernel<true, true><<<dropout_grid(total), 256, 0, st>>>(x, gy, total, D, th, ik, used, 0, nullptr, gx);
    else dropout_kernel<false, true><<<dropout_grid(total), 256, 0, st>>>(x, gy, total, D, th, ik, used, 0, nullptr, gx);
    OTVAE_CHECK_LAUNCH("otvae_dropout_bwd");
    return OTVAE_OK;
}
