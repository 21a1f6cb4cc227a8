// Checks common.h's lane_xor<MASK> (DPP / v_permlane16_swap / v_permlane32_swap forms) against the definition on the hardware:
// out[lane] must be the input of lane ^ MASK for MASK = 1, 2, 4, 8, 16, 32, and wave_sum must equal the __shfl_xor butterfly bit for bit.
//   hipcc --offload-arch=gfx950 -Iartalk_amd/csrc -Iinclude tools/lane_xor_probe.hip -o tools/build/lane_xor_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "common.h"
using namespace artalk;
__global__ void probe(float* o, const float* in) {
    const float v = in[threadIdx.x];
    o[threadIdx.x + 0 * 64] = lane_xor<1>(v); o[threadIdx.x + 1 * 64] = lane_xor<2>(v); o[threadIdx.x + 2 * 64] = lane_xor<4>(v);
    o[threadIdx.x + 3 * 64] = lane_xor<8>(v); o[threadIdx.x + 4 * 64] = lane_xor<16>(v); o[threadIdx.x + 5 * 64] = lane_xor<32>(v);
    o[threadIdx.x + 6 * 64] = wave_sum(v);
    float s = v;
    for (int m = 32; m > 0; m >>= 1) s += __shfl_xor(s, m, 64);
    o[threadIdx.x + 7 * 64] = s;
    o[threadIdx.x + 8 * 64] = wave_max(v);
}
int main() {
    float h[64], out[9 * 64], *din, *dout;
    srand(7);
    for (int i = 0; i < 64; ++i) h[i] = (float)rand() / RAND_MAX * 3.f - 1.f + i * 1e-3f;
    if (hipMalloc(&din, sizeof h) != hipSuccess || hipMalloc(&dout, sizeof out) != hipSuccess) return 2;
    hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dout, din);
    if (hipMemcpy(out, dout, sizeof out, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    int bad = 0;
    for (int k = 0; k < 6; ++k)
        for (int l = 0; l < 64; ++l)
            if (out[k * 64 + l] != h[l ^ (1 << k)]) { if (bad < 8) printf("mask %d lane %d: got %g want %g\n", 1 << k, l, out[k * 64 + l], h[l ^ (1 << k)]); ++bad; }
    for (int l = 0; l < 64; ++l)
        if (memcmp(&out[6 * 64 + l], &out[7 * 64 + l], 4) != 0) { if (bad < 8) printf("wave_sum lane %d: %a vs shfl %a\n", l, out[6 * 64 + l], out[7 * 64 + l]); ++bad; }
    float mx = h[0];
    for (int l = 1; l < 64; ++l) mx = h[l] > mx ? h[l] : mx;
    for (int l = 0; l < 64; ++l) if (out[8 * 64 + l] != mx) ++bad;
    printf(bad ? "lane_xor: %d mismatches\n" : "lane_xor: all masks, wave_sum (bit-identical to the __shfl_xor butterfly) and wave_max correct\n", bad);
    return bad ? 1 : 0;
}
