// Does a workgroup that re-streams a private set of a few hundred KB get it from the Infinity Cache (MALL) or from DRAM -- and what do
// FETCH_SIZE / WRITE_SIZE count?  The access pattern of the production QP kernel's far arrays: 512 resident workgroups (two per CU),
// each reading (and rewriting a third of) its OWN `set` bytes `passes` times, 8 bytes a lane, coalesced.  Three runs:
//   resident   set = 216 KB per workgroup: 512 x 216 KB = 110 MB live -- misses the 4 MiB L2 of an XCD (13.8 MB per XCD), fits the 256 MiB MALL
//   l2         set = 48 KB per workgroup: 3 MB per XCD -- L2 resident (the floor: no fabric traffic after the first pass)
//   stream     every pass reads a DIFFERENT 216 KB slice of a 14 GB buffer: nothing is ever re-read -- DRAM
// Prints GB/s of each; run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` (and WRITE_SIZE, separate pass) to see that the counters
// report the same bytes for `resident` and `stream` (they sit at the L2's fabric side, in front of the MALL) while the time differs.
//   hipcc --offload-arch=gfx950 -O2 -o tools/probe/mall_probe tools/probe/mall_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void __launch_bounds__(256, 2) restream(double* buf, size_t set_doubles, size_t pass_stride_doubles, int passes, double* sink) {
    double* mine = buf + (size_t)blockIdx.x * set_doubles;
    double acc = 0.0;
    for (int p = 0; p < passes; ++p) {
        double* b = mine + (size_t)p * pass_stride_doubles;
        for (size_t i = threadIdx.x; i < set_doubles; i += 4 * 256) {      // four independent requests a lane in flight
            double v0 = b[i], v1 = (i + 256 < set_doubles) ? b[i + 256] : 0.0, v2 = (i + 512 < set_doubles) ? b[i + 512] : 0.0, v3 = (i + 768 < set_doubles) ? b[i + 768] : 0.0;
            acc += (v0 + v1) + (v2 + v3);
            if ((i / 1024) % 3 == 0) b[i] = v0 + 1.0;                       // a third of the lines are rewritten, as the kernel's far arrays are
        }
        __syncthreads();
    }
    if (acc == 12345.678) sink[0] = acc;
}
static double run(const char* name, double* buf, size_t set_b, size_t stride_b, int passes, int nwg, double* sink) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    restream<<<nwg, 256>>>(buf, set_b / 8, stride_b / 8, 2, sink);   // warm
    hipDeviceSynchronize();
    hipEventRecord(e0);
    restream<<<nwg, 256>>>(buf, set_b / 8, stride_b / 8, passes, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)nwg * set_b * passes;
    printf("%-9s set %4zu KB x %d workgroups = %7.1f MB live, %3d passes: %8.3f ms, read %7.2f GB -> %7.1f GB/s (+ a third of it written back)\n", name, set_b / 1024, nwg,
           stride_b ? -1.0 : nwg * set_b / 1048576.0, passes, ms, bytes / 1e9, bytes / 1e9 / (ms * 1e-3));
    return ms;
}
int main() {
    const int nwg = 512, passes = 50;
    const size_t set_big = 216 * 1024, set_small = 48 * 1024;
    const size_t total = (size_t)nwg * set_big * (passes + 2);             // 'stream': workgroup b, pass p -> slice (p * nwg + b)
    double *buf, *sink; 
    if (hipMalloc(&buf, total) != hipSuccess) { printf("hipMalloc of %.1f GB failed\n", total / 1e9); return 1; }
    hipMalloc(&sink, 8); hipMemset(buf, 0, total);
    run("l2", buf, set_small, 0, passes, nwg, sink);
    run("resident", buf, set_big, 0, passes, nwg, sink);
    run("stream", buf, set_big, (size_t)nwg * set_big, passes, nwg, sink);
    return 0;
}
