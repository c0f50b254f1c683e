// Counter-based dropout masks shared by the kernels that drop inside their own arithmetic (attention_dropout.hip,
// layernorm.hip): keep(row, col) is a hash of (call key, row, col), so a backward kernel recomputes the mask of its
// forward instead of reading a stored one.  The call key comes from `key` = {seed, call counter} in DEVICE memory (a
// captured hipGraph must draw fresh masks on every replay: the host advances the counter with a captured add) and a
// per-call-site stream id.
#pragma once
#include <stdint.h>

__device__ __forceinline__ uint32_t mix32(uint32_t x) {  // a full-avalanche 32-bit finaliser
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

__device__ __forceinline__ uint64_t call_key(const int64_t* __restrict__ key, int stream_id) {  // splitmix64 finaliser
    uint64_t s = (uint64_t)key[0] + 0x9E3779B97F4A7C15ull * ((uint64_t)key[1] * 4096ull + (uint64_t)stream_id + 1ull);
    s ^= s >> 30;
    s *= 0xBF58476D1CE4E5B9ull;
    s ^= s >> 27;
    s *= 0x94D049BB133111EBull;
    s ^= s >> 31;
    return s;
}

// row = slice * T + query token
__device__ __forceinline__ uint32_t row_hash(uint64_t ck, uint32_t row) { return mix32(row ^ (uint32_t)ck) ^ (uint32_t)(ck >> 32); }
__device__ __forceinline__ bool keep_pair(uint32_t rh, int s, uint32_t thresh) {
    return mix32(rh + (uint32_t)s * 0x9E3779B9U) >= thresh;
}

// dropout probability -> 32-bit threshold: keep iff hash >= thresh, i.e. with probability 1 - thresh / 2^32
static inline uint32_t dropout_threshold(float p) {
    const double scaled = (double)p * 4294967296.0;
    return scaled >= 4294967295.0 ? 4294967295u : (uint32_t)scaled;
}
