// Row LayerNorm variants of the path, one 64-lane wavefront per row (no LDS, no barrier):
//   wav2vec2 conv stack   LN(512, affine, eps 1e-5) + GELU(erf)         hf:291-299
//   wav2vec2 encoder      LN(1024, affine, eps 1e-5)                    hf:631-654,791
//   AR blocks / head      LN(768, no affine, eps 1e-6) * (1+scale)+shift   app/transformer.py:35,40; app/models.py:148
//   VAE blocks            LN(512, affine, eps 1e-5)                     app/modules/bitwise_vae.py:203
//   style encoder         LN(128, affine, eps 1e-5)  (post-LN)          app/modules/style_encoder.py:15-21
// HBM-bound: each row is read once with 16-byte loads and written once.
#include <cstdlib>
#include "common.h"

namespace artalk {

// RPW rows per wave, all of their loads issued before the first reduction (more bytes in flight per wave: the one-row version
// ran at 2.8 TB/s on the 614 400 x 512 conv activations).
template <int D, int RPW>
__global__ __launch_bounds__(256) void layernorm_kernel(const LnArgs a) {
    constexpr int NV = D / 64;                 // floats per lane
    constexpr int VW = (NV % 4 == 0) ? 4 : 2;  // vector width of one access
    constexpr int NA = NV / VW;                // accesses per lane
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (row0 >= a.M) return;
    float v[RPW][NV];
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const int row = min(row0 + rr, a.M - 1);          // tail rows re-read the last row (never stored)
        const float* x = a.X + (long)row * a.ldx;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int c = (i * 64 + lane) * VW;
            if (VW == 4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(x + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[rr][i * VW + e] = t[e];
            } else {
                const float2 t = *reinterpret_cast<const float2*>(x + c);
                v[rr][i * VW] = t.x; v[rr][i * VW + 1] = t.y;
            }
        }
    }
    // affine / modulation vectors are fetched as 16-byte vectors BEFORE the reductions, so their latency (the AdaLN rows come out
    // of a 1.3 GB table) hides behind the row statistics instead of showing in front of the stores
    float wv[NV], bv[NV];
    if (a.w) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int c = (i * 64 + lane) * VW;
            if (VW == 4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(a.w + c), u = *reinterpret_cast<const f32x4*>(a.b + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) { wv[i * VW + e] = t[e]; bv[i * VW + e] = u[e]; }
            } else {
                const float2 t = *reinterpret_cast<const float2*>(a.w + c), u = *reinterpret_cast<const float2*>(a.b + c);
                wv[i * VW] = t.x; wv[i * VW + 1] = t.y; bv[i * VW] = u.x; bv[i * VW + 1] = u.y;
            }
        }
    }
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const int row = row0 + rr;
        if (row >= a.M) break;
        float scv[NV], shv[NV];
        if (a.scale) {
            const long mr = map_row(a.mmap, row);
            const float* sc = a.scale + mr * a.ldm;
            const float* sh = a.shift + mr * a.ldm;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int c = (i * 64 + lane) * VW;
                if (VW == 4) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(sc + c), u = *reinterpret_cast<const f32x4*>(sh + c);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { scv[i * VW + e] = t[e]; shv[i * VW + e] = u[e]; }
                } else {
                    const float2 t = *reinterpret_cast<const float2*>(sc + c), u = *reinterpret_cast<const float2*>(sh + c);
                    scv[i * VW] = t.x; scv[i * VW + 1] = t.y; shv[i * VW] = u.x; shv[i * VW + 1] = u.y;
                }
            }
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) s += v[rr][i];
        const float mean = wave_sum(s) * (1.0f / D);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) { const float d = v[rr][i] - mean; q += d * d; }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / D) + a.eps);
        float* y = a.Y + (long)row * a.ldy;
        const bool junk = a.junk_period && row % a.junk_period >= a.junk_from;      // layout padding row: stored as zeros (kernels.h)
        int* const status = junk ? nullptr : a.status;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int c = (i * 64 + lane) * VW;
            float o[VW];
#pragma unroll
            for (int e = 0; e < VW; ++e) {
                float t = (v[rr][i * VW + e] - mean) * rstd;
                if (a.w) t = t * wv[i * VW + e] + bv[i * VW + e];
                if (a.scale) t = t * (scv[i * VW + e] + 1.0f) + shv[i * VW + e];
                o[e] = junk ? 0.f : apply_act_rt(t, a.act);
            }
            if (VW == 4 && a.out_p8) {
                store_p8x4_pair(y, c, o[0], o[1], o[2], o[3], status, a.p8_exp);      // c = 4 * (64 i + lane): adjacent lanes, adjacent runs
            } else if (VW == 4) {
                f32x4 t = {o[0], o[1], o[2], o[3]};
                *reinterpret_cast<f32x4*>(y + c) = t;
            } else {
                *reinterpret_cast<float2*>(y + c) = make_float2(o[0], o[1]);
            }
        }
    }
}

template <int D, int RPW>
static void launch_ln(const LnArgs& a, hipStream_t s) {
    const int rows_per_block = 4 * RPW;
    ARTALK_LAUNCH((layernorm_kernel<D, RPW>), dim3((a.M + rows_per_block - 1) / rows_per_block), dim3(256), 0, s, a);
}

void launch_layernorm(const LnArgs& a, hipStream_t s) {
    if (a.M <= 0) return;
    const bool big = a.M >= 8192;      // several rows per wave only when there are enough rows to fill the chip anyway
    switch (a.D) {
        case 128: launch_ln<128, 1>(a, s); break;
        case 512: if (big) launch_ln<512, 4>(a, s); else launch_ln<512, 1>(a, s); break;
        case 768: if (big) launch_ln<768, 2>(a, s); else launch_ln<768, 1>(a, s); break;
        case 1024: launch_ln<1024, 1>(a, s); break;      // one row per wave (wav2vec2 encoder: 0.35 ms per step faster than 2, 4 slower still)
        default: abort();
    }
}

}  // namespace artalk
