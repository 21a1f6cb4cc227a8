// Device-side helpers shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels.h"

namespace artalk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int map_row(const RowMap& r, int m) {
    return r.rpb == INT_MAX ? m : (m / r.rpb) * r.bstride + r.off + (m % r.rpb);
}

// torch.nn.functional.gelu (erf form): x * 0.5 * (1 + erf(x / sqrt(2)))
__device__ __forceinline__ float gelu_erf(float x) { return x * 0.5f * (1.0f + erff(x * 0.70710678118654752440f)); }
// gelu(approximate='tanh'): 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))
__device__ __forceinline__ float gelu_tanh(float x) {
    const float kBeta = 0.79788456080286535588f, kKappa = 0.044715f;
    float inner = kBeta * (x + kKappa * x * x * x);
    return 0.5f * x * (1.0f + tanhf(inner));
}
__device__ __forceinline__ float silu(float x) { return x / (1.0f + expf(-x)); }

template <int ACT>
__device__ __forceinline__ float apply_act(float x) {
    if (ACT == ACT_GELU_ERF) return gelu_erf(x);
    if (ACT == ACT_GELU_TANH) return gelu_tanh(x);
    if (ACT == ACT_LEAKY02) return x > 0.f ? x : 0.2f * x;
    return x;
}
__device__ __forceinline__ float apply_act_rt(float x, int act) {
    switch (act) {
        case ACT_GELU_ERF: return gelu_erf(x);
        case ACT_GELU_TANH: return gelu_tanh(x);
        case ACT_LEAKY02: return x > 0.f ? x : 0.2f * x;
        default: return x;
    }
}

// ---- "P8" split format (gemm_f16s.hip): every 8 consecutive elements of a row become 32 bytes [8 x f16 hi][8 x f16 lo],
// x = hi + lo/2048.  Same pitch as fp32.  store_p8x4 writes elements c..c+3 (c % 4 == 0) of a row whose storage starts at `row`.
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_p8x4(float* row, int c, float x0, float x1, float x2, float x3) {
    const float x[4] = {x0, x1, x2, x3};
    f16x4_t h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        h[e] = (_Float16)x[e];
        l[e] = (_Float16)((x[e] - (float)h[e]) * 2048.0f);
    }
    unsigned char* g = reinterpret_cast<unsigned char*>(row) + (c >> 3) * 32 + (c & 4) * 2;
    *reinterpret_cast<f16x4_t*>(g) = h;
    *reinterpret_cast<f16x4_t*>(g + 16) = l;
}

// full-wave (64 lanes) butterfly reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

}  // namespace artalk
