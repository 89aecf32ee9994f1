// dependent-issue latencies and issue rates on gfx950 (one wave, nothing else running): shader cycles per step.
// The timers are inline asm with data dependencies on the chain (s_memtime is not ordered against VALU work).
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 512
#define TIMER(t, dep) asm volatile("s_nop 0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "v"(dep) : "memory")
template <int OP>
__global__ void k(double* out, long long* cyc, double a, double b) {
    double x = a + threadIdx.x * 1e-9, y = b;
    double z0 = x, z1 = x + 1, z2 = x + 2, z3 = x + 3, z4 = x + 4, z5 = x + 5, z6 = x + 6, z7 = x + 7;
    long long t0, t1;
    double s0 = x + z7;
    TIMER(t0, s0);
#pragma unroll 16
    for (int i = 0; i < N; ++i) {
        if (OP == 0) x = __builtin_fma(x, y, b);
        if (OP == 1) { int lo = __builtin_amdgcn_readlane(__double2loint(x), 3), hi = __builtin_amdgcn_readlane(__double2hiint(x), 3);
                       x = __builtin_fma(__hiloint2double(hi, lo), y, x); }
        if (OP == 2) x = __builtin_amdgcn_rsq(x) + b;
        if (OP == 3) { int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0xB1, 0xf, 0xf, false), hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0xB1, 0xf, 0xf, false);
                       x = x + __hiloint2double(hi, lo) * y; }
        if (OP == 4) { int ad = (threadIdx.x ^ 4) * 4; int lo = __builtin_amdgcn_ds_bpermute(ad, __double2loint(x)), hi = __builtin_amdgcn_ds_bpermute(ad, __double2hiint(x));
                       x = x + __hiloint2double(hi, lo) * y; }
        if (OP == 5) x = __builtin_amdgcn_rcp(x) + b;
        if (OP == 6) { z0 = __builtin_fma(z0, y, b); z1 = __builtin_fma(z1, y, b); z2 = __builtin_fma(z2, y, b); z3 = __builtin_fma(z3, y, b);
                       z4 = __builtin_fma(z4, y, b); z5 = __builtin_fma(z5, y, b); z6 = __builtin_fma(z6, y, b); z7 = __builtin_fma(z7, y, b); }
        if (OP == 7) { z0 = z0 * y; z1 = z1 * y; z2 = z2 * y; z3 = z3 * y; z4 = z4 + y; z5 = z5 + y; z6 = z6 + y; z7 = z7 + y; }
    }
    x += ((z0 + z1) + (z2 + z3)) + ((z4 + z5) + (z6 + z7));
    TIMER(t1, x);
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    double* out; long long* cyc;
    (void)hipMallocManaged(&out, 64 * 8); (void)hipMallocManaged(&cyc, 8);
    const char* names[] = {"v_fma_f64 chain", "readlane x2 -> fma_f64", "v_rsq_f64 + add", "dpp mov x2 -> mul/add", "ds_bpermute x2 -> mul/add", "v_rcp_f64 + add",
                           "8 independent v_fma_f64", "4 v_mul_f64 + 4 v_add_f64 independent"};
#define RUN(OP) k<OP><<<1, 64>>>(out, cyc, 0.9, 0.5); (void)hipDeviceSynchronize(); k<OP><<<1, 64>>>(out, cyc, 0.9, 0.5); (void)hipDeviceSynchronize(); printf("%-40s %.1f cycles per step\n", names[OP], (double)cyc[0] / N);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7)
    return 0;
}
