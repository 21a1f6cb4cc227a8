// Can HIP events recorded INSIDE a captured graph be timed (hipEventElapsedTime) after the graph has been launched?  bench.py times the
// dominant kernel with events around its launches inside the timed region; capturing the wav2vec2 phase into a graph is only an
// option if those events keep working.  Also prints the idle gap between two dependent kernels launched eagerly vs from a graph.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(float* p, int iters) {
    float x = p[threadIdx.x];
    for (int i = 0; i < iters; ++i) x = x * 1.0001f + 0.5f;
    p[threadIdx.x] = x;
}
int main() {
    float* d; hipMalloc(&d, 4096); hipMemset(d, 0, 4096);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t e[4]; for (auto& x : e) hipEventCreate(&x);
    const int N = 200;
    // eager: N dependent kernels between two events
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e[0], s);
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, d, 2000);
        hipEventRecord(e[1], s);
        hipStreamSynchronize(s);
    }
    float ms = 0; hipEventElapsedTime(&ms, e[0], e[1]);
    printf("eager : %d dependent kernels %.1f us each\n", N, ms * 1e3f / N);
    // graph with event-record nodes in the middle
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    hipError_t r0 = hipEventRecord(e[2], s);
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, d, 2000);
    hipError_t r1 = hipEventRecord(e[3], s);
    hipError_t r2 = hipStreamEndCapture(s, &g);
    hipError_t r3 = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    printf("capture: record %s / %s, end %s, instantiate %s\n", hipGetErrorString(r0), hipGetErrorString(r1), hipGetErrorString(r2), hipGetErrorString(r3));
    if (r2 == hipSuccess && r3 == hipSuccess) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e[0], s);
            hipGraphLaunch(ge, s);
            hipEventRecord(e[1], s);
            hipStreamSynchronize(s);
            float outer = 0, inner = -1;
            hipEventElapsedTime(&outer, e[0], e[1]);
            hipError_t r4 = hipEventElapsedTime(&inner, e[2], e[3]);
            printf("graph : %d dependent kernels %.1f us each (events around the launch); events recorded INSIDE the graph: %s, %.1f us each\n",
                   N, outer * 1e3f / N, hipGetErrorString(r4), inner * 1e3f / N);
        }
    }
    return 0;
}
