// probe: cost of per-block, per-channel 64-bit integer atomics (agent scope, no return) at the end of a kernel -- the
// building block of an order-independent (exact fixed-point) BatchNorm statistics accumulation without a finalize launch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <bool ATOMIC>
__global__ __launch_bounds__(256) void k(unsigned long long* acc, int C, int stride, float* sink, int work) {
    float v = threadIdx.x;
    for (int i = 0; i < work; ++i) v = fmaf(v, 1.0001f, 0.5f);  // some body work
    if (ATOMIC) {
        for (int c = threadIdx.x; c < C; c += 256)
            __hip_atomic_fetch_add(acc + (size_t)c * stride, (unsigned long long)(blockIdx.x + c), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
    }
    if (v == 12345.f) sink[0] = v;
}

int main() {
    unsigned long long* acc;
    float* sink;
    hipMalloc(&acc, 64 << 20);
    hipMalloc(&sink, 64);
    hipMemset(acc, 0, 64 << 20);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int reps = 200;
    for (int blocks : {256, 512, 2048}) {
        for (int C : {64, 384, 1536}) {  // accumulators per block: e.g. 32 channels x 2 sums x 1, x3 limbs, 256 ch x 2 x 3
            for (int stride : {1, 32, 512}) {
                if ((size_t)C * stride * 8 > (64u << 20)) continue;
                float ms[2];
                for (int a = 0; a < 2; ++a) {
                    for (int w = 0; w < 3; ++w) {
                        if (a) k<true><<<blocks, 256>>>(acc, C, stride, sink, 2000);
                        else k<false><<<blocks, 256>>>(acc, C, stride, sink, 2000);
                    }
                    hipEventRecord(e0);
                    for (int r = 0; r < reps; ++r) {
                        if (a) k<true><<<blocks, 256>>>(acc, C, stride, sink, 2000);
                        else k<false><<<blocks, 256>>>(acc, C, stride, sink, 2000);
                    }
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    hipEventElapsedTime(&ms[a], e0, e1);
                }
                printf("blocks %4d  accumulators/block %4d  stride %4d B: %.2f us without, %.2f us with atomics (+%.2f us)\n",
                       blocks, C, stride * 8, ms[0] / reps * 1e3, ms[1] / reps * 1e3, (ms[1] - ms[0]) / reps * 1e3);
            }
        }
    }
    return 0;
}
