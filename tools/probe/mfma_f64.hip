// probe: register layout of v_mfma_f64_16x16x4_f64 on gfx950 (which D[i][j] does lane l, register r hold?)
// hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_f64.hip -o tools/probe/mfma_f64 && tools/probe/mfma_f64
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4v __attribute__((ext_vector_type(4)));

__global__ void layout(double* out, int mode) {
    const int l = threadIdx.x;
    double a, b;
    if (mode == 0) {  // assume A[i][k]: i = l % 16, k = l / 16; only k == 0 non-zero: D[i][j] = (i + 1) * 100 (j + 1)
        a = (l / 16 == 0) ? (double)(l % 16 + 1) : 0.0;
        b = (l / 16 == 0) ? 100.0 * (l % 16 + 1) : 0.0;
    } else {          // k-dependence: A[i][k] = 1, B[k][j] = 10^k  ->  D = 1111 everywhere if all four k are summed
        a = 1.0;
        b = (l / 16 == 0) ? 1.0 : (l / 16 == 1) ? 10.0 : (l / 16 == 2) ? 100.0 : 1000.0;
    }
    double4v c = {0.0, 0.0, 0.0, 0.0};
    double4v d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = d[r];
}

int main() {
    double* dev;
    hipMalloc(&dev, 256 * 8);
    double h[256];
    for (int mode = 0; mode < 2; ++mode) {
        layout<<<1, 64>>>(dev, mode);
        hipMemcpy(h, dev, sizeof(h), hipMemcpyDeviceToHost);
        if (mode == 1) {
            printf("k-sum check: %g %g %g\n", h[0], h[100], h[255]);
            continue;
        }
        for (int l : {0, 1, 15, 16, 17, 32, 48, 63}) {
            printf("lane %2d:", l);
            for (int r = 0; r < 4; ++r) {
                const int v = (int)(h[l * 4 + r] + 0.5), i = (v / 100) ? 0 : 0;
                (void)i;
                // v = (i+1) * 100 * (j+1): recover (i, j) knowing j = lane % 16 candidates
                int fi = -1, fj = -1;
                for (int ii = 0; ii < 16; ++ii)
                    for (int jj = 0; jj < 16; ++jj)
                        if ((ii + 1) * 100 * (jj + 1) == v && (jj == l % 16 || ii == l % 16)) { fi = ii; fj = jj; }
                printf("  r%d=%6d (i=%2d,j=%2d)", r, v, fi, fj);
            }
            printf("\n");
        }
    }
    return 0;
}
