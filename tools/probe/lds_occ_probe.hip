// How many 256-lane workgroups fit a CU as a function of the dynamic LDS size (the allocation granularity decides whether
// 160 KiB / 3 can be asked for): hipOccupancyMaxActiveBlocksPerMultiprocessor of a trivial kernel.
//   hipcc --offload-arch=gfx950 -O2 -o tools/probe/lds_occ_probe tools/probe/lds_occ_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256, 3) k(double* out) { extern __shared__ double s[]; s[threadIdx.x] = 1.0; __syncthreads(); out[threadIdx.x] = s[255 - threadIdx.x]; }
int main() {
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int last = -1;
    for (int b = 38 * 1024; b <= 82 * 1024; b += 64) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)k, 256, (size_t)b) != hipSuccess) { printf("error at %d\n", b); return 1; }
        if (n != last) { printf("from %6d B (%.3f KiB): %d workgroups per CU\n", b, b / 1024.0, n); last = n; }
    }
    return 0;
}
