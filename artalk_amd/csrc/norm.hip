// Row LayerNorm variants of the path, one 64-lane wavefront per row (no LDS, no barrier):
//   wav2vec2 conv stack   LN(512, affine, eps 1e-5) + GELU(erf)         hf:291-299
//   wav2vec2 encoder      LN(1024, affine, eps 1e-5)                    hf:631-654,791
//   AR blocks / head      LN(768, no affine, eps 1e-6) * (1+scale)+shift   app/transformer.py:35,40; app/models.py:148
//   VAE blocks            LN(512, affine, eps 1e-5)                     app/modules/bitwise_vae.py:203
//   style encoder         LN(128, affine, eps 1e-5)  (post-LN)          app/modules/style_encoder.py:15-21
// HBM-bound: each row is read once with 16-byte loads and written once.
#include "common.h"

namespace artalk {

template <int D>
__global__ __launch_bounds__(256) void layernorm_kernel(const LnArgs a) {
    constexpr int NV = D / 64;                 // floats per lane
    constexpr int VW = (NV % 4 == 0) ? 4 : 2;  // vector width of one access
    constexpr int NA = NV / VW;                // accesses per lane
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.M) return;
    const float* x = a.X + (long)row * a.ldx;
    float v[NV];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int c = (i * 64 + lane) * VW;
        if (VW == 4) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(x + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[i * VW + e] = t[e];
        } else {
            const float2 t = *reinterpret_cast<const float2*>(x + c);
            v[i * VW] = t.x; v[i * VW + 1] = t.y;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += v[i];
    const float mean = wave_sum(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) { const float d = v[i] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / D) + a.eps);
    const float* sc = nullptr; const float* sh = nullptr;
    if (a.scale) {
        const long mr = map_row(a.mmap, row);
        sc = a.scale + mr * a.ldm; sh = a.shift + mr * a.ldm;
    }
    float* y = a.Y + (long)row * a.ldy;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int c = (i * 64 + lane) * VW;
        float o[VW];
#pragma unroll
        for (int e = 0; e < VW; ++e) {
            float t = (v[i * VW + e] - mean) * rstd;
            if (a.w) t = t * a.w[c + e] + a.b[c + e];
            if (sc) t = t * (sc[c + e] + 1.0f) + sh[c + e];
            o[e] = apply_act_rt(t, a.act);
        }
        if (VW == 4 && a.out_p8) {
            store_p8x4(y, c, o[0], o[1], o[2], o[3]);
        } else if (VW == 4) {
            f32x4 t = {o[0], o[1], o[2], o[3]};
            *reinterpret_cast<f32x4*>(y + c) = t;
        } else {
            *reinterpret_cast<float2*>(y + c) = make_float2(o[0], o[1]);
        }
    }
}

void launch_layernorm(const LnArgs& a, hipStream_t s) {
    if (a.M <= 0) return;
    dim3 grid((a.M + 3) / 4), block(256);
    switch (a.D) {
        case 128: hipLaunchKernelGGL(layernorm_kernel<128>, grid, block, 0, s, a); break;
        case 512: hipLaunchKernelGGL(layernorm_kernel<512>, grid, block, 0, s, a); break;
        case 768: hipLaunchKernelGGL(layernorm_kernel<768>, grid, block, 0, s, a); break;
        case 1024: hipLaunchKernelGGL(layernorm_kernel<1024>, grid, block, 0, s, a); break;
        default: abort();
    }
}

}  // namespace artalk
