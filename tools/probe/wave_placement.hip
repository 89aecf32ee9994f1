// Where do the four waves of a 256-thread workgroup land?  Same launch shape as the production QP kernel (256 threads,
// ~77 KB of LDS: two workgroups per CU).  Prints, per CU, the SIMD of wave 0 of each co-resident workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ void __launch_bounds__(256, 2) probe(unsigned* out, int spin) {
    extern __shared__ double smem[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    long long t0 = clock64();
    double acc = 0;
    while (clock64() - t0 < spin) { acc += smem[threadIdx.x]; }
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 4 + threadIdx.x / 64) * 2] = hw;
        out[(blockIdx.x * 4 + threadIdx.x / 64) * 2 + 1] = xcc;
    }
    if (acc == 12345.678) out[0] = 0;
}
int main() {
    const int B = 512;
    unsigned* d; hipMalloc(&d, B * 4 * 2 * sizeof(unsigned));
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 77 * 1024);
    hipLaunchKernelGGL(probe, dim3(B), dim3(256), 77 * 1024, 0, d, 2000000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(B * 8);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
    std::map<unsigned, std::vector<int>> percu;
    int hist[4][4] = {};
    for (int b = 0; b < B; ++b) {
        int s[4];
        for (int w = 0; w < 4; ++w) { unsigned hw = h[(b * 4 + w) * 2]; s[w] = (hw >> 4) & 3; hist[w][s[w]]++; }
        unsigned hw = h[b * 8], xcc = h[b * 8 + 1] & 0xf;
        unsigned key = (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf);
        percu[key].push_back(s[0]);
        if (b < 8) printf("wg %d: simd of waves %d %d %d %d  cu key %05x\n", b, s[0], s[1], s[2], s[3], key);
    }
    for (int w = 0; w < 4; ++w) printf("wave %d on simd: %d %d %d %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    int same = 0, pairs = 0, other = 0;
    for (auto& kv : percu) { if (kv.second.size() == 2) { pairs++; same += kv.second[0] == kv.second[1]; } else other++; }
    printf("CUs with two workgroups: %d, of which wave 0 of both on the SAME simd: %d; CUs with another count: %d (of %zu)\n", pairs, same, other, percu.size());
    return 0;
}
