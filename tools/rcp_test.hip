// accuracy of v_rcp_f64 and of upr_rcp (estimate + one second-order step) on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "../upright_amd/csrc/upr_common.h"
__global__ void k(const double* x, double* y0, double* y1, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    y0[i] = __builtin_amdgcn_rcp(x[i]);
    y1[i] = upr_rcp(x[i]);
}
int main() {
    const int n = 1 << 20;
    double *x, *y0, *y1;
    hipMallocManaged(&x, n * 8); hipMallocManaged(&y0, n * 8); hipMallocManaged(&y1, n * 8);
    for (int i = 0; i < n; ++i) x[i] = ldexp(1.0 + (double)i / n, (i % 121) - 60);   // 1e-18 .. 1e18: slacks and multipliers of the IPM
    k<<<n / 256, 256>>>(x, y0, y1, n);
    hipDeviceSynchronize();
    double e0 = 0, e1 = 0;
    for (int i = 0; i < n; ++i) {
        long double r = 1.0L / (long double)x[i];
        e0 = fmax(e0, fabs((double)((y0[i] - r) / r))); e1 = fmax(e1, fabs((double)((y1[i] - r) / r)));
    }
    printf("max rel err: v_rcp_f64 %.3e, upr_rcp %.3e\n", e0, e1);
    return 0;
}
