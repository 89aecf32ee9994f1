// Probe: can a kernel compiled at run time with hiprtc (a) include the engine's headers, (b) use more than 64 KB of dynamic LDS
// through hipModuleLaunchKernel on gfx950?  hipcc tools/probe/hiprtc_probe.cpp -lhiprtc -o tools/probe/hiprtc_probe
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <chrono>
#include <cstdio>
#include <string>
#include <vector>

#define CK(x) do { auto e_ = (x); if (e_ != 0) { printf("FAIL %s -> %d\n", #x, (int)e_); return 1; } } while (0)

int main(int argc, char** argv) {
    const char* inc = argc > 1 ? argv[1] : "upright_amd/csrc";
    std::string src =
        "#include \"upr_common.h\"\n"
        "extern \"C\" __global__ void probe(double* out, int n) {\n"
        "    extern __shared__ double sm[];\n"
        "    for (int i = threadIdx.x; i < n; i += blockDim.x) sm[i] = (double)i;\n"
        "    __syncthreads();\n"
        "    double s = 0.0; for (int i = threadIdx.x; i < n; i += blockDim.x) s += sm[n - 1 - i];\n"
        "    atomicAdd(out, s * upr_rcp(1.0));\n"
        "}\n";
    hiprtcProgram prog;
    CK(hiprtcCreateProgram(&prog, src.c_str(), "probe.hip", 0, nullptr, nullptr));
    std::string I = std::string("-I") + inc;
    const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", I.c_str(), "-I/opt/rocm/include"};
    auto t0 = std::chrono::steady_clock::now();
    hiprtcResult rc = hiprtcCompileProgram(prog, 5, opts);
    size_t ls = 0; hiprtcGetProgramLogSize(prog, &ls);
    if (ls > 1) { std::string log(ls, 0); hiprtcGetProgramLog(prog, &log[0]); printf("log: %s\n", log.c_str()); }
    CK(rc);
    printf("compiled in %.2f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    size_t cs = 0; CK(hiprtcGetCodeSize(prog, &cs));
    std::vector<char> code(cs); CK(hiprtcGetCode(prog, code.data()));
    hipModule_t mod; hipFunction_t f;
    CK(hipModuleLoadData(&mod, code.data()));
    CK(hipModuleGetFunction(&f, mod, "probe"));
    double* out; CK(hipMalloc(&out, 8)); CK(hipMemset(out, 0, 8));
    for (int kb : {32, 64, 80, 120, 156}) {
        int n = kb * 1024 / 8;
        size_t lds = (size_t)n * 8;
        hipError_t ea = hipSuccess;
        if (lds > 64 * 1024) ea = hipFuncSetAttribute((const void*)f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        CK(hipMemset(out, 0, 8));
        void* args[] = {&out, &n};
        hipError_t el = hipModuleLaunchKernel(f, 1, 1, 1, 256, 1, 1, (unsigned)lds, nullptr, args, nullptr);
        hipError_t es = hipDeviceSynchronize();
        double h = 0; hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
        printf("lds %3d KB: setattr %d launch %d sync %d result %.0f expected %.0f\n", kb, (int)ea, (int)el, (int)es, h, 0.5 * n * (n - 1.0));
    }
    return 0;
}
