// accuracy of v_rsq_f64 and of one / two Newton steps on it (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double* x, double* y0, double* y1, double* y2, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a = x[i];
    double y = __builtin_amdgcn_rsq(a);
    y0[i] = y;
    y = y * (1.5 - 0.5 * a * y * y); y1[i] = y;
    y = y * (1.5 - 0.5 * a * y * y); y2[i] = y;
}
int main() {
    const int n = 1 << 20;
    double *x, *y0, *y1, *y2;
    hipMallocManaged(&x, n * 8); hipMallocManaged(&y0, n * 8); hipMallocManaged(&y1, n * 8); hipMallocManaged(&y2, n * 8);
    for (int i = 0; i < n; ++i) x[i] = ldexp(1.0 + (double)i / n * 3.0, (i % 41) - 20);
    k<<<n / 256, 256>>>(x, y0, y1, y2, n);
    hipDeviceSynchronize();
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; ++i) {
        long double r = 1.0L / sqrtl((long double)x[i]);
        e0 = fmax(e0, fabs((double)((y0[i] - r) / r))); e1 = fmax(e1, fabs((double)((y1[i] - r) / r))); e2 = fmax(e2, fabs((double)((y2[i] - r) / r)));
    }
    printf("max rel err: rsq %.3e, +1 Newton %.3e, +2 Newton %.3e\n", e0, e1, e2);
    return 0;
}
