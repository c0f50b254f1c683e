// probe: operand layout and issue cost of v_mfma_f32_4x4x1_16b_f32 on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float4v __attribute__((ext_vector_type(4)));

__global__ void layout_kernel(float* out) {
    const int l = threadIdx.x;
    // a = 1 + lane (row value), b = 100 * (1 + lane) (column value)  ->  D[i][j] of block = a_i * b_j
    float4v c = {0.f, 0.f, 0.f, 0.f};
    float4v d = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(1 + l), 100.f * (float)(1 + l), c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = d[r];
}

template <int MODE>
__global__ __launch_bounds__(256) void cost_kernel(float* out, int iters, float seed) {
    const int l = threadIdx.x;
    float4v acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    float4v negm = {-1.f, -1.f, -1.f, -1.f};
    float q0 = seed * l, q1 = q0 + 1.f, q2 = q0 + 2.f, q3 = q0 + 3.f, k = 0.001f * l, b = 0.5f + l;
    float s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {  // VALU reference: 16 pairs: fma + exp + 4 accumulations
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float p0 = __builtin_amdgcn_exp2f(fmaf(q0, k + r, -1.f));
                float p1 = __builtin_amdgcn_exp2f(fmaf(q1, k + r, -1.f));
                float p2 = __builtin_amdgcn_exp2f(fmaf(q2, k + r, -1.f));
                float p3 = __builtin_amdgcn_exp2f(fmaf(q3, k + r, -1.f));
                acc0[0] += p0; acc0[1] = fmaf(p0, b, acc0[1]); acc0[2] = fmaf(p0, k, acc0[2]); acc0[3] = fmaf(p0, q0, acc0[3]);
                acc1[0] += p1; acc1[1] = fmaf(p1, b, acc1[1]); acc1[2] = fmaf(p1, k, acc1[2]); acc1[3] = fmaf(p1, q0, acc1[3]);
                acc2[0] += p2; acc2[1] = fmaf(p2, b, acc2[1]); acc2[2] = fmaf(p2, k, acc2[2]); acc2[3] = fmaf(p2, q0, acc2[3]);
                acc3[0] += p3; acc3[1] = fmaf(p3, b, acc3[1]); acc3[2] = fmaf(p3, k, acc3[2]); acc3[3] = fmaf(p3, q0, acc3[3]);
            }
            k += 1e-6f;
        } else {  // MFMA: per query set 1 score MFMA + 4 exp + 4 accumulate MFMAs
            float4v sc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(k, q0, negm, 0, 0, 0);
            float4v sc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(k, q1, negm, 0, 0, 0);
            float4v sc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(k, q2, negm, 0, 0, 0);
            float4v sc3 = __builtin_amdgcn_mfma_f32_4x4x1f32(k, q3, negm, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(__builtin_amdgcn_exp2f(sc0[r]), b + r, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(__builtin_amdgcn_exp2f(sc1[r]), b + r, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(__builtin_amdgcn_exp2f(sc2[r]), b + r, acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_4x4x1f32(__builtin_amdgcn_exp2f(sc3[r]), b + r, acc3, 0, 0, 0);
            }
            k += 1e-6f;
        }
    }
    float s = 0;
    for (int r = 0; r < 4; ++r) s += acc0[r] + acc1[r] + acc2[r] + acc3[r];
    out[blockIdx.x * 256 + l] = s + s0 + s1 + s2 + s3;
}

template <int MODE>
__global__ __launch_bounds__(256) void pipe_kernel(float* out, int iters, float seed) {
    const int l = threadIdx.x;
    float4v a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
    typedef float float16v __attribute__((ext_vector_type(4)));
    float x = seed * l, y = 0.5f + l;
    float e0 = x, e1 = x + 1, e2 = x + 2, e3 = x + 3, e4 = x + 4, e5 = x + 5, e6 = x + 6, e7 = x + 7;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 2) {  // 16 MFMA 4x4x1, 8 independent accumulators
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a3, 0, 0, 0);
                a4 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a4, 0, 0, 0);
                a5 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a5, 0, 0, 0);
                a6 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a6, 0, 0, 0);
                a7 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a7, 0, 0, 0);
            }
        } else if (MODE == 3) {  // 16 exps, 8 independent chains
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                e0 = __builtin_amdgcn_exp2f(e0); e1 = __builtin_amdgcn_exp2f(e1); e2 = __builtin_amdgcn_exp2f(e2); e3 = __builtin_amdgcn_exp2f(e3);
                e4 = __builtin_amdgcn_exp2f(e4); e5 = __builtin_amdgcn_exp2f(e5); e6 = __builtin_amdgcn_exp2f(e6); e7 = __builtin_amdgcn_exp2f(e7);
            }
        } else if (MODE == 4) {  // 16 MFMA 16x16x4
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
                a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a4, 0, 0, 0);
                a5 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a5, 0, 0, 0);
                a6 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a6, 0, 0, 0);
                a7 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a7, 0, 0, 0);
            }
        } else if (MODE == 5) {  // 16 fma, 8 chains
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                e0 = fmaf(e0, y, x); e1 = fmaf(e1, y, x); e2 = fmaf(e2, y, x); e3 = fmaf(e3, y, x);
                e4 = fmaf(e4, y, x); e5 = fmaf(e5, y, x); e6 = fmaf(e6, y, x); e7 = fmaf(e7, y, x);
            }
        } else if (MODE == 6) {  // 8 MFMA 4x4x1 + 8 exp interleaved, independent
            a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 0, 0, 0); e0 = __builtin_amdgcn_exp2f(e0);
            a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a1, 0, 0, 0); e1 = __builtin_amdgcn_exp2f(e1);
            a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a2, 0, 0, 0); e2 = __builtin_amdgcn_exp2f(e2);
            a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a3, 0, 0, 0); e3 = __builtin_amdgcn_exp2f(e3);
            a4 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a4, 0, 0, 0); e4 = __builtin_amdgcn_exp2f(e4);
            a5 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a5, 0, 0, 0); e5 = __builtin_amdgcn_exp2f(e5);
            a6 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a6, 0, 0, 0); e6 = __builtin_amdgcn_exp2f(e6);
            a7 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a7, 0, 0, 0); e7 = __builtin_amdgcn_exp2f(e7);
        }
    }
    float s = e0 + e1 + e2 + e3 + e4 + e5 + e6 + e7;
    for (int r = 0; r < 4; ++r) s += a0[r] + a1[r] + a2[r] + a3[r] + a4[r] + a5[r] + a6[r] + a7[r];
    out[blockIdx.x * 256 + l] = s;
}

template <int MODE>
void run_pipe(float* d, const char* what, int nper) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, blocks = 1024;
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        pipe_kernel<MODE><<<blocks, 256>>>(d, iters, 1e-4f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    // SIMD cycles per instruction per wave: 4 waves per SIMD (1024 blocks x 4 waves / 1024 SIMDs)
    double cyc = ms * 1e-3 * 2.4e9 / (4.0 * iters * nper);
    printf("%s: %.3f ms -> %.2f SIMD cycles per instruction @2.4GHz\n", what, ms, cyc);
}

int main() {
    float* d;
    hipMalloc(&d, 1 << 22);
    layout_kernel<<<1, 64>>>(d);
    std::vector<float> h(256);
    hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost);
    for (int l = 0; l < 12; ++l) printf("lane %2d: %10.0f %10.0f %10.0f %10.0f\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    for (int mode = 0; mode < 2; ++mode) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 20000, blocks = 1024;  // 4 blocks (16 waves) per CU
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) cost_kernel<0><<<blocks, 256>>>(d, iters, 1e-4f);
            else cost_kernel<1><<<blocks, 256>>>(d, iters, 1e-4f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            double pairs = (double)blocks * 256 * iters * 16;
            printf("mode %d: %.3f ms, %.2f T pairs/s, %.1f SIMD-cycles/pair-wave @2.4GHz\n", mode, ms, pairs / ms / 1e9,
                   1024.0 * 2.4e9 * 64 / (pairs / (ms * 1e-3)));
        }
    }
    run_pipe<2>(d, "mfma 4x4x1 only (16/iter)", 16);
    run_pipe<3>(d, "v_exp only (16/iter)", 16);
    run_pipe<4>(d, "mfma 16x16x4 only (16/iter)", 16);
    run_pipe<5>(d, "v_fma only (16/iter)", 16);
    run_pipe<6>(d, "8 mfma 4x4x1 + 8 exp (per instruction)", 16);
    return 0;
}
