// Standard-normal draws for the re-parametrisation noise (reference prior/gaussian.py:93 `q.rsample()`) and the prior
// samples of the minibatch-OT prior, generated on the device from a counter-based hash so that a captured training step
// draws fresh noise on every replay without a host-side generator call between replays.
//
// key = device int64[3] {seed, call counter, block ticket}.  Element e of call c is Box-Muller over two 32-bit hashes of
// (call key(seed, c, stream_id), e): the values depend on (seed, counter, stream_id, e) only -- not on the launch shape.
// The LAST block to finish advances the counter (ticket = number of blocks that have read the key and written their part),
// so every block of a call sees the same counter and the next call (stream-ordered) sees counter + 1.
#include "common.h"
#include "dropout_hash.h"

__device__ __forceinline__ float u01_open(uint32_t h) {  // (0, 1]: 24 random bits, never 0 (log below)
    return (float)((h >> 8) + 1u) * (1.0f / 16777216.0f);
}

__global__ __launch_bounds__(256) void normal_fill_kernel(float* __restrict__ out, int64_t n, int64_t* __restrict__ key, int stream_id,
                                                          int advance) {
    const uint64_t ck = call_key(key, stream_id);
    const uint32_t lo = (uint32_t)ck, hi = (uint32_t)(ck >> 32);
    const int64_t pairs = (n + 1) >> 1;
    for (int64_t p = blockIdx.x * (int64_t)256 + threadIdx.x; p < pairs; p += (int64_t)gridDim.x * 256) {
        const uint32_t c0 = (uint32_t)p, c1 = (uint32_t)(p >> 32);
        const uint32_t h0 = mix32(mix32(c0 ^ lo) + c1 * 0x9E3779B9U + hi);
        const uint32_t h1 = mix32(h0 ^ (c0 * 0x85EBCA6BU + 0x632BE5ABU) ^ hi);
        const float r = sqrtf(-2.0f * __logf(u01_open(h0)));
        float s, c;
        __sincosf(6.28318530717958647692f * u01_open(h1), &s, &c);
        out[2 * p] = r * c;
        if (2 * p + 1 < n) out[2 * p + 1] = r * s;
    }
    if (advance) {
        __syncthreads();  // every thread of this block has read the key
        if (threadIdx.x == 0) {
            const unsigned long long t =
                __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(key + 2), 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (t + 1 == (unsigned long long)gridDim.x) {  // all blocks are past their read of key[1]
                key[2] = 0;
                key[1] += 1;
            }
        }
    }
}

extern "C" int otvae_normal_fill(float* out, int64_t n, int64_t* key, int stream_id, int advance, void* stream) {
    OTVAE_REQUIRE(out && key && n > 0 && stream_id >= 0 && stream_id < 4095, "otvae_normal_fill: bad argument");
    const int64_t b = cdiv((n + 1) >> 1, (int64_t)1024);
    normal_fill_kernel<<<(int)(b < 1024 ? b : 1024), 256, 0, (hipStream_t)stream>>>(out, n, key, stream_id, advance);
    OTVAE_CHECK_LAUNCH("otvae_normal_fill");
    return OTVAE_OK;
}
