// Where do the waves of two co-resident 256-thread workgroups (79.8 KB LDS each: two per CU) land?  Prints, per CU, the
// workgroups resident together and the SIMD of each of their waves.  hipcc --offload-arch=gfx950 -O2 -o simd_probe simd_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#include <tuple>
__global__ void __launch_bounds__(256, 2) probe(unsigned* out, long long spin) {
    extern __shared__ double smem[];
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);
    const long long t0 = __builtin_readcyclecounter();
    smem[threadIdx.x] = (double)hw;
    while (__builtin_readcyclecounter() - t0 < spin) __builtin_amdgcn_s_sleep(8);
    if ((threadIdx.x & 63) == 0) {
        unsigned* o = out + (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
        o[0] = hw; o[1] = xcc; o[2] = (unsigned)(t0 & 0xffffffffu); o[3] = (unsigned)smem[threadIdx.x];
    }
}
int main() {
    const int B = 1024;
    unsigned* d; hipMalloc(&d, B * 16 * sizeof(unsigned));
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 81712);
    probe<<<B, 256, 81712>>>(d, 2000000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(B * 16);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    std::map<std::tuple<int,int,int,int>, std::vector<int>> cus;
    int hist[4][4] = {};
    for (int b = 0; b < B; ++b) {
        const unsigned hw = h[b * 16], xcc = h[b * 16 + 1] & 0xf;
        cus[{(int)xcc, (int)((hw >> 13) & 7), (int)((hw >> 12) & 1), (int)((hw >> 8) & 15)}].push_back(b);
        for (int w = 0; w < 4; ++w) hist[w][(h[(b * 4 + w) * 4] >> 4) & 3]++;
    }
    printf("wave -> SIMD histogram (rows: wave 0..3, columns: SIMD 0..3)\n");
    for (int w = 0; w < 4; ++w) printf("  %d: %5d %5d %5d %5d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    printf("CUs seen: %zu\n", cus.size());
    int n = 0, same0 = 0, pairs = 0;
    for (auto& kv : cus) {
        auto& v = kv.second;
        if (n++ < 6) {
            printf("xcc %d se %d sh %d cu %2d:", std::get<0>(kv.first), std::get<1>(kv.first), std::get<2>(kv.first), std::get<3>(kv.first));
            for (int b : v) { printf("  wg %4d t0 %10u simd", b, h[b * 16 + 2]); for (int w = 0; w < 4; ++w) printf(" %u", (h[(b * 4 + w) * 4] >> 4) & 3); }
            printf("\n");
        }
        // first-round pair: the two workgroups with the smallest t0
        if (v.size() >= 2) {
            std::vector<std::pair<unsigned,int>> s; for (int b : v) s.push_back({h[b * 16 + 2], b});
            std::sort(s.begin(), s.end());
            ++pairs; if (((h[s[0].second * 16] >> 4) & 3) == ((h[s[1].second * 16] >> 4) & 3)) ++same0;
        }
    }
    printf("first-round pairs whose wave 0 share a SIMD: %d of %d\n", same0, pairs);
    return 0;
}
