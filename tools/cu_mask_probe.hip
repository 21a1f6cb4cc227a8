// Does hipExtStreamCreateWithCUMask work here, and which CUs / XCDs do the mask bits select?  Launches a kernel of many one-wave
// workgroups that record (XCC_ID, SE, CU) on streams with different masks and prints the sets that ran them, plus the time of a
// fixed amount of ALU work (so the CU count is visible as a slowdown).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <vector>
__global__ void probe(unsigned* out, int iters) {
    float x = threadIdx.x;
    for (int i = 0; i < iters; ++i) x = x * 1.0001f + 0.5f;
    if (threadIdx.x == 0) {
        unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));     // HW_REG_HW_ID
        unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));   // HW_REG_XCC_ID
        out[blockIdx.x * 2] = hw; out[blockIdx.x * 2 + 1] = xcc;
    }
    if (x == 12345.f) out[0] = 1;
}
static void run(const char* name, std::vector<uint32_t> mask) {
    hipStream_t s;
    hipError_t e = mask.empty() ? hipStreamCreate(&s) : hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) { printf("%s: stream creation failed: %s\n", name, hipGetErrorString(e)); return; }
    const int n = 4096;
    unsigned* d; hipMalloc(&d, n * 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(probe, dim3(n), dim3(64), 0, s, d, 20000);
    hipStreamSynchronize(s);
    hipEventRecord(a, s);
    hipLaunchKernelGGL(probe, dim3(n), dim3(64), 0, s, d, 20000);
    hipEventRecord(b, s);
    hipStreamSynchronize(s);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned> h(n * 2);
    hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
    std::set<unsigned> cus, xccs;
    int per_xcc[16] = {0};
    std::set<unsigned> cu_of_xcc[16];
    for (int i = 0; i < n; ++i) {
        unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
        unsigned cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5);
        cus.insert(cu | (xcc << 8)); xccs.insert(xcc); cu_of_xcc[xcc].insert(cu);
    }
    printf("%-28s %.3f ms  distinct CUs %3zu  XCCs %zu  per-XCC CUs:", name, ms, cus.size(), xccs.size());
    for (unsigned x = 0; x < 8; ++x) printf(" %zu", cu_of_xcc[x].size());
    printf("\n");
    hipFree(d); hipStreamDestroy(s);
}
int main() {
    run("no mask", {});
    run("all 256 bits", std::vector<uint32_t>(8, 0xffffffffu));
    run("low 128 bits", {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0});
    run("high 128 bits", {0, 0, 0, 0, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu});
    run("even bits", std::vector<uint32_t>(8, 0x55555555u));
    run("low 32 bits", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0});
    run("bits 0-7 of each word", std::vector<uint32_t>(8, 0x000000ffu));
    run("low 64 bits", {0xffffffffu, 0xffffffffu, 0, 0, 0, 0, 0, 0});
    return 0;
}
