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

// torch.nn.functional.gelu (erf form): x * 0.5 * (1 + erf(x / sqrt(2))) = x * Phi(x), evaluated as
//     Phi(x) = h            (x < 0)        h = 0.5 * erfc(|x| / sqrt 2) = 0.5 * 2^P(t),  t = min(|x| / sqrt 2, 4)
//            = 1 - h        (x >= 0)       P(t) = log2(erfc(t)) ~ t * Q8(t)   (minimax in erfc-weighted absolute error: 2.2e-9)
// one polynomial, one v_exp_f32, no branch: 14 VALU instructions where libm's two-branch erff (both branches run in a 64-lane wave)
// took about 40 - the largest single item left in the FFN-in / conv epilogues.  Accuracy against float64 over
// |x| <= 7: 1.9 ulp of the result for x > 0 (the libm formula in fp32: 1.6 ulp; the two differ by at most 2 ulp), absolute error
// below 4e-7 everywhere; for x < 0 it has no cancellation (1 + erf loses its leading bits there).  tests/test_ops_gpu.py checks it.
__device__ __forceinline__ float gelu_erf(float x) {
    const float z = x * 0.70710678118654752440f;
    const float t = fminf(fabsf(z), 4.0f);
    float p = 0x1.8564d6p-17f;
    p = fmaf(p, t, -0x1.40ca46p-13f);
    p = fmaf(p, t, 0x1.bcb83cp-11f);
    p = fmaf(p, t, -0x1.2a2936p-9f);
    p = fmaf(p, t, 0x1.63b57cp-14f);
    p = fmaf(p, t, 0x1.c63cep-6f);
    p = fmaf(p, t, -0x1.2fbc0ep-3f);
    p = fmaf(p, t, -0x1.d63e26p-1f);
    p = fmaf(p, t, -0x1.a0be88p+0f);
    const float h = 0.5f * __builtin_amdgcn_exp2f(p * t);
    return x * (z < 0.f ? h : 1.0f - h);
}
// The same on two values at once: the polynomial and the scalings run on the packed-fp32 pipe (v_pk_fma_f32 / v_pk_mul_f32: two
// lanes' worth of fp32 per instruction on gfx950), bit-identical to gelu_erf per element.  For kernels that are VALU-bound on it (conv0).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
    const f32x2 z = x * 0.70710678118654752440f;
    f32x2 t;
    t.x = fminf(fabsf(z.x), 4.0f); t.y = fminf(fabsf(z.y), 4.0f);
    f32x2 p = {0x1.8564d6p-17f, 0x1.8564d6p-17f};
    p = __builtin_elementwise_fma(p, t, (f32x2)(-0x1.40ca46p-13f));
    p = __builtin_elementwise_fma(p, t, (f32x2)(0x1.bcb83cp-11f));
    p = __builtin_elementwise_fma(p, t, (f32x2)(-0x1.2a2936p-9f));
    p = __builtin_elementwise_fma(p, t, (f32x2)(0x1.63b57cp-14f));
    p = __builtin_elementwise_fma(p, t, (f32x2)(0x1.c63cep-6f));
    p = __builtin_elementwise_fma(p, t, (f32x2)(-0x1.2fbc0ep-3f));
    p = __builtin_elementwise_fma(p, t, (f32x2)(-0x1.d63e26p-1f));
    p = __builtin_elementwise_fma(p, t, (f32x2)(-0x1.a0be88p+0f));
    p = p * t;
    f32x2 e;
    e.x = __builtin_amdgcn_exp2f(p.x); e.y = __builtin_amdgcn_exp2f(p.y);
    const f32x2 hpos = __builtin_elementwise_fma(e, (f32x2)(-0.5f), (f32x2)(1.0f)), hneg = e * 0.5f;      // 1 - h, h
    f32x2 phi;
    phi.x = z.x < 0.f ? hneg.x : hpos.x; phi.y = z.y < 0.f ? hneg.y : hpos.y;
    return x * phi;
}
// gelu(approximate='tanh'): 0.5 x (1 + tanh(u)), u = sqrt(2/pi) (x + 0.044715 x^3), through the identity 0.5 (1 + tanh u) = 1 / (1 + e^(-2u)):
// one v_exp_f32 and one v_rcp_f32 (8 instructions; libm's tanhf about 35), no cancellation.  2.3 ulp against float64 for x > 0
// (the libm formula in fp32: 1.7 ulp).  The constants are -2 sqrt(2/pi) log2(e) and that times 0.044715.
__device__ __forceinline__ float gelu_tanh(float x) {
    const float arg = x * fmaf(-0x1.a5a7dp-4f, x * x, -0x1.26aec2p+1f);     // -2u * log2(e)
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(arg));
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

// The same for the 16 values a lane holds of a 32x32 MFMA sub-tile: ONE (wave-uniform) switch around the loops, so that a kernel
// carries each activation's code once per call site instead of once per element behind a branch.

__device__ __forceinline__ void apply_act16(f32x16& v, int act) {
    switch (act) {
        case ACT_GELU_ERF:
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = gelu_erf(v[e]);
            break;
        case ACT_GELU_TANH:
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = gelu_tanh(v[e]);
            break;
        case ACT_LEAKY02:
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = v[e] > 0.f ? v[e] : 0.2f * v[e];
            break;
        default: break;
    }
}

__device__ __forceinline__ void apply_act4(f32x4& v, int act) {
    switch (act) {
        case ACT_GELU_ERF:
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
            break;
        case ACT_GELU_TANH:
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_tanh(v[e]);
            break;
        case ACT_LEAKY02:
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.2f * v[e];
            break;
        default: break;
    }
}

// ---- "P8" split format (gemm_f16s.hip): an operand x is first scaled by a power of two S (activations 2^e with e the SITE
// EXPONENT of the producing site - kActExp = 4, i.e. 16, unless the model's calibration lowered it for a site with outlier
// activations, engine.hip SiteScales -, weights kWScale), then every 8 consecutive elements of a row become 32 bytes
// [8 x f16 hi][8 x f16 lo] with hi = f16(S x) and lo = f16(S x - hi) (UNSCALED residual; gfx950's f16 MFMA honours subnormal inputs -
// tools/mfma_subnormal_probe.py - so a residual below 2^-14 still carries 2^-25 absolute precision).  Same pitch as fp32.  Because
// both halves are in one scale, hi*hi + hi*lo + lo*hi accumulate in ONE fp32 accumulator; the GEMM epilogue multiplies by
// 1 / (SA * SW) (exact: powers of two).  Range: |x| < 65504 / S (4094 at the default 16) for activations, < 255 for weights; an
// overflow raises the status word at the producer (the host recalibrates the site scales, or falls back to exact f32).
// Producers and consumers carry the exponent e (an int: kernels.h GemmArgs::a_exp / c_exp, LnArgs::p8_exp, AttnArgs::qkv_exp / o_exp):
// scale, 1 / scale and the guard threshold are two scalar integer operations away from it.
constexpr float kActScale = 16.0f, kWScale = 256.0f, kOutScale = 1.0f / (16.0f * 256.0f);
__host__ __device__ __forceinline__ float p8_scale_of(int e) { return __builtin_bit_cast(float, (unsigned)(127 + e) << 23); }       // 2^e
__host__ __device__ __forceinline__ float p8_out_scale_of(int a_exp) { return __builtin_bit_cast(float, (unsigned)(127 - a_exp - 8) << 23); }   // 1 / (2^a_exp * kWScale)
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split_f16(float xs, _Float16& hi, _Float16& lo) {   // xs already scaled
    hi = (_Float16)xs;
    lo = (_Float16)(xs - (float)hi);
}
// Range guard of every P8 producer: an activation with |x| * S beyond fp16's largest finite value (or a NaN) would become
// inf in the hi half and poison every product it takes part in; the producer reports it in the model's status word (bit 3,
// include/artalk_hip.h: artalk_get_status) at the place it happens instead of relying on the NaN reaching a bit decision.
constexpr int kStatusP8Range = 8;
// |x| * 2^e <= 65504 (fp16's largest finite value), as a bound on the bit pattern of |x|: finite positive floats order like
// their bit patterns and NaN / inf patterns lie above every finite one, so ONE unsigned maximum + ONE compare covers range, inf
// and NaN (an fmaxf chain would drop NaNs and need a compare per element).  bits(65504 / 2^e) = bits(65504) - (e << 23).
constexpr unsigned int kP8MaxBits = 0x457FE000u;      // bits of 4094.0f = 65504 / 16 (e = kActExp)
__host__ __device__ __forceinline__ unsigned int p8_maxbits_of(int e) { return (unsigned int)(0x477FE000 - e * 0x00800000); }
__device__ __forceinline__ unsigned int abs_bits(float x) { return __float_as_uint(x) & 0x7FFFFFFFu; }
__device__ __forceinline__ void p8_guard(int* status, float a, float b, float c, float d, unsigned int maxbits = kP8MaxBits) {
    if (status) {
        const unsigned int m = max(max(abs_bits(a), abs_bits(b)), max(abs_bits(c), abs_bits(d)));
        if (m > maxbits) atomicOr(status, kStatusP8Range);
    }
}
__device__ __forceinline__ void p8_guard16(int* status, const f32x16& v, unsigned int maxbits = kP8MaxBits) {      // one compare for a whole 32x32 MFMA sub-tile
    if (status) {
        unsigned int m = 0;
#pragma unroll
        for (int e = 0; e < 16; ++e) m = max(m, abs_bits(v[e]));
        if (m > maxbits) atomicOr(status, kStatusP8Range);
    }
}
// store_p8x4 writes activation elements c..c+3 (c % 4 == 0) of a row whose storage starts at `row`, with site exponent e.
__device__ __forceinline__ void store_p8x4(float* row, int c, float x0, float x1, float x2, float x3, int* status = nullptr, int e = kActExp) {
    p8_guard(status, x0, x1, x2, x3, p8_maxbits_of(e));
    const float sc = p8_scale_of(e);
    const float x[4] = {x0, x1, x2, x3};
    f16x4_t h, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) { _Float16 a, b; split_f16(x[i] * sc, a, b); h[i] = a; l[i] = b; }
    unsigned char* g = reinterpret_cast<unsigned char*>(row) + (c >> 3) * 32 + (c & 4) * 2;
    *reinterpret_cast<f16x4_t*>(g) = h;
    *reinterpret_cast<f16x4_t*>(g + 16) = l;
}

// The same for lanes that own ADJACENT runs of 4 columns (lane L: c, lane L+1: c+4 of one 8-group, L even; all 64 lanes active):
// neighbours trade halves so that each issues ONE 16-byte store (even lane the hi chunk, odd lane the lo chunk) instead of two 8-byte ones.
__device__ __forceinline__ void store_p8x4_pair(float* row, int c, float x0, float x1, float x2, float x3, int* status = nullptr, int pe = kActExp) {
    p8_guard(status, x0, x1, x2, x3, p8_maxbits_of(pe));
    const float sc = p8_scale_of(pe);
    const float x[4] = {x0, x1, x2, x3};
    f16x4_t h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) { _Float16 a, b; split_f16(x[e] * sc, a, b); h[e] = a; l[e] = b; }
    const uint2 hu = __builtin_bit_cast(uint2, h), lu = __builtin_bit_cast(uint2, l);
    const bool odd = (c & 4) != 0;
    const unsigned sx = odd ? hu.x : lu.x, sy = odd ? hu.y : lu.y;
    const unsigned rx = __builtin_amdgcn_mov_dpp(sx, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    const unsigned ry = __builtin_amdgcn_mov_dpp(sy, 0xB1, 0xF, 0xF, true);
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const u4 w = odd ? u4{rx, ry, lu.x, lu.y} : u4{hu.x, hu.y, rx, ry};
    *reinterpret_cast<u4*>(reinterpret_cast<unsigned char*>(row) + (c >> 3) * 32 + (odd ? 16 : 0)) = w;
}

// ---- GEMM epilogue for one 32x32 MFMA tile computed with the WEIGHT rows as the A operand and the activation rows as the B
// operand, so that the accumulator is C^T: lane (r = lane & 31, h = lane >> 5) holds
//     v[e] = C[row][col0 + (e & 3) + 8 * (e >> 2) + 4 * h],   row = (tile's first row) + r,
// i.e. four runs of 4 consecutive columns.  Stores, residual and gate reads are therefore 16 bytes per lane (a quarter of the
// vector-memory instructions of the column-per-lane layout: the store tail of a tile is instruction-issue bound, 7.6 us -> see
// DESIGN.md), and a P8 result leaves as whole 32-byte groups after one half-wave exchange.
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
struct EpiCtx {
    const float* bias; float* C; const float* R; bool vec;
};
__device__ __forceinline__ EpiCtx make_epi(const GemmArgs& g, const float* bias, float* C, const float* R) {
    EpiCtx x; x.bias = bias; x.C = C; x.R = R;
    unsigned long long bits = (unsigned long long)(g.N | g.ldc) | ((unsigned long long)C >> 2);
    if (bias) bits |= (unsigned long long)bias >> 2;
    if (R) bits |= (unsigned long long)g.ldr | ((unsigned long long)R >> 2);
    if (g.gate) bits |= (unsigned long long)g.ldg | ((unsigned long long)g.gate >> 2);
    x.vec = (bits & 3) == 0;     // every row start and every run of 4 columns is 16-byte aligned
    return x;
}
// P8OK = false (the exact-fp32 kernels, which never produce a P8 result) compiles the split / guard paths out: with them in, the
// 128x128 fp32 kernel went from 116 to 216 registers + scratch and the f32 mode from 227 to 665 ms per step.
// GUARD = false (the large-grid kernels): no range guard in this epilogue (gemm_f16s.hip, launch_gemm_p8).
// DUAL = true (the small-grid kernel): a fp32 result may be written a second time in the P8 format (g.c2, same pitch and row map).
template <bool P8OK = true, bool GUARD = true, bool DUAL = false>
__device__ __forceinline__ void epilogue_tile32(const GemmArgs& g, const EpiCtx& x, int row, int col0, int h, f32x16& v) {   // v is clobbered
    const bool rok = row < g.M;
    const long crow = rok ? map_row(g.cmap, row) : 0;
    const float* gp = (g.gate && rok) ? g.gate + (long)map_row(g.gmap, row) * g.ldg : nullptr;
    const float* rp = (x.R && rok) ? x.R + crow * g.ldr : nullptr;
    if (x.vec) {
        f32x4 act_gv[4], act_rv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = col0 + 8 * q + 4 * h;
            const bool ok = c < g.N;          // N % 4 == 0: the whole run is inside or outside
            f32x4 b = {0.f, 0.f, 0.f, 0.f}, gv = {1.f, 1.f, 1.f, 1.f}, rv = {0.f, 0.f, 0.f, 0.f};
            if (x.bias && ok) b = *reinterpret_cast<const f32x4*>(x.bias + c);
            if (gp && ok) gv = *reinterpret_cast<const f32x4*>(gp + c);
            if (rp && ok) rv = *reinterpret_cast<const f32x4*>(rp + c);
            if (g.act == ACT_NONE) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = v[4 * q + e] + b[e];
                    if (g.gate) t *= gv[e];
                    v[4 * q + e] = t + rv[e];
                }
            } else {      // (no launch combines an activation with a gate or a residual, but the order is kept: act(x + b) * gate + res)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * q + e] += b[e];
                act_gv[q] = gv; act_rv[q] = rv;
            }
        }
        if (g.act != ACT_NONE) {
            apply_act16(v, g.act);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = v[4 * q + e];
                    if (g.gate) t *= act_gv[q][e];
                    v[4 * q + e] = t + act_rv[q][e];
                }
        }
        const bool dual = DUAL && g.c2 && !g.c_p8;
        if (dual && rok) {      // the fp32 copy first; the P8 copy below takes the same values
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = col0 + 8 * q + 4 * h;
                if (c < g.N) {
                    const f32x4 o = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
                    *reinterpret_cast<f32x4*>(x.C + crow * g.ldc + c) = o;
                }
            }
        }
        float* const p8dst = dual ? g.c2 : x.C;
        if (P8OK && (g.c_p8 || dual)) {   // N % 8 == 0.  Pairs of 8-column groups (k, k+1): lanes h=0 end up with all of group k, lanes h=1 with group k+1
            if (GUARD && rok) {
                const unsigned int mb = p8_maxbits_of(g.c_exp);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (col0 + 8 * q + 4 * h < g.N) p8_guard(g.status, v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3], mb);
            }
            const float csc = p8_scale_of(g.c_exp);
#pragma unroll
            for (int qp = 0; qp < 2; ++qp) {
                unsigned int w[2][4];     // [group of the pair][hi.x, hi.y, lo.x, lo.y] of this lane's 4 columns
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    f16x4_t hh, ll;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        _Float16 a, b;
                        split_f16(v[8 * qp + 4 * k + e] * csc, a, b);
                        hh[e] = a; ll[e] = b;
                    }
                    const uint2 hu = __builtin_bit_cast(uint2, hh), lu = __builtin_bit_cast(uint2, ll);
                    w[k][0] = hu.x; w[k][1] = hu.y; w[k][2] = lu.x; w[k][3] = lu.y;
                }
#pragma unroll
                for (int d = 0; d < 4; ++d) {   // v_permlane32_swap: upper half of w[0] <-> lower half of w[1]
                    const auto sw = __builtin_amdgcn_permlane32_swap(w[0][d], w[1][d], false, false);
                    w[0][d] = sw[0]; w[1][d] = sw[1];
                }
                const int gc = col0 + 16 * qp + 8 * h;
                if (rok && gc < g.N) {
                    unsigned char* o = reinterpret_cast<unsigned char*>(p8dst + crow * g.ldc + gc);
                    const u32x4_t hi = {w[0][0], w[0][1], w[1][0], w[1][1]}, lo = {w[0][2], w[0][3], w[1][2], w[1][3]};
                    *reinterpret_cast<u32x4_t*>(o) = hi;
                    *reinterpret_cast<u32x4_t*>(o + 16) = lo;
                }
            }
        } else if (rok) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = col0 + 8 * q + 4 * h;
                if (c < g.N) {
                    const f32x4 o = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
                    *reinterpret_cast<f32x4*>(x.C + crow * g.ldc + c) = o;
                }
            }
        }
        return;
    }
    if (!rok) return;
#pragma unroll
    for (int e = 0; e < 16; ++e) {      // unaligned shapes (N = 106, odd leading dimensions): one column at a time
        const int col = col0 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (col < g.N) {
            float t = apply_act_rt(v[e] + (x.bias ? x.bias[col] : 0.f), g.act);
            if (gp) t *= gp[col];
            if (rp) t += rp[col];
            if (P8OK && g.c_p8) {
                if (GUARD) p8_guard(g.status, t, 0.f, 0.f, 0.f, p8_maxbits_of(g.c_exp));
                _Float16* o = reinterpret_cast<_Float16*>(x.C + crow * g.ldc + (col & ~7));
                _Float16 hh, ll;
                split_f16(t * p8_scale_of(g.c_exp), hh, ll);
                o[col & 7] = hh;
                o[8 + (col & 7)] = ll;
            } else {
                x.C[crow * g.ldc + col] = t;
            }
        }
    }
}
// All TM x TN sub-tiles of a wave through ONE copy of epilogue_tile32: the sub-tile to finish is always acc[0][0] and the others move
// up behind it (16 register moves per remaining sub-tile and trip) - for kernels whose epilogue is cold or rare code.  Not for the
// fp32 128x128 kernel and the non-persistent two-workgroup kernel: their exposed epilogues overlap the sub-tiles' loads and
// stores when unrolled (looped: f32 mode 222 -> 238 ms per step).  scale: factor applied first (1 for none).
template <bool P8OK, bool GUARD, int TM, int TN, bool DUAL = false>
__device__ __forceinline__ void epilogue_tiles(const GemmArgs& g, const EpiCtx& x, int row0, int col0, int h, f32x16 (&acc)[TM][TN],
                                               float scale = 1.0f) {
    f32x16* a = &acc[0][0];
#pragma nounroll
    for (int t = 0; t < TM * TN; ++t) {
        if (scale != 1.0f) {
#pragma unroll
            for (int e = 0; e < 16; ++e) a[0][e] *= scale;
        }
        epilogue_tile32<P8OK, GUARD, DUAL>(g, x, row0 + (t / TN) * 32, col0 + (t % TN) * 32, h, a[0]);
#pragma unroll
        for (int u = 0; u + 1 < TM * TN; ++u) a[u] = a[u + 1];
    }
}
// Store-only form of epilogue_tile32 for a tile whose bias is already added and that has no gate / residual, every lane valid and
// the 16-byte path available (the deferred epilogue of gemm_p8_2wgp_kernel): activation, then exactly 4 store instructions.
// rok = false (a row beyond M in an edge tile): the lane computes along (the half-wave exchange of a P8 result needs every lane) and
// its stores are masked off - the wave still issues exactly 4 store instructions, which the callers' counted vmcnt waits rely on.
template <bool GUARD = true>
__device__ __forceinline__ void epilogue_tile32_store(const GemmArgs& g, float* C, int row, int col0, int h, f32x16& v, const f32x4* res = nullptr,
                                                      bool rok = true) {
    const long crow = rok ? map_row(g.cmap, row) : 0;
    apply_act16(v, g.act);
    if (res) {      // residual of the deferred tile (4 runs of 4 columns, staged through LDS by the persistent kernel)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * q + e] += res[q][e];
    }
    if (g.c_p8) {
        if (GUARD) p8_guard16(g.status, v, p8_maxbits_of(g.c_exp));
        const float csc = p8_scale_of(g.c_exp);
#pragma unroll
        for (int qp = 0; qp < 2; ++qp) {
            unsigned int w[2][4];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                f16x4_t hh, ll;
#pragma unroll
                for (int e = 0; e < 4; ++e) { _Float16 a, b; split_f16(v[8 * qp + 4 * k + e] * csc, a, b); hh[e] = a; ll[e] = b; }
                const uint2 hu = __builtin_bit_cast(uint2, hh), lu = __builtin_bit_cast(uint2, ll);
                w[k][0] = hu.x; w[k][1] = hu.y; w[k][2] = lu.x; w[k][3] = lu.y;
            }
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const auto sw = __builtin_amdgcn_permlane32_swap(w[0][d], w[1][d], false, false);
                w[0][d] = sw[0]; w[1][d] = sw[1];
            }
            unsigned char* o = reinterpret_cast<unsigned char*>(C + crow * g.ldc + col0 + 16 * qp + 8 * h);
            const u32x4_t hi = {w[0][0], w[0][1], w[1][0], w[1][1]}, lo = {w[0][2], w[0][3], w[1][2], w[1][3]};
            if (rok) {
                *reinterpret_cast<u32x4_t*>(o) = hi;
                *reinterpret_cast<u32x4_t*>(o + 16) = lo;
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 o = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
            if (rok) *reinterpret_cast<f32x4*>(C + crow * g.ldc + col0 + 8 * q + 4 * h) = o;
        }
    }
}

// ---- Coalesced epilogue through wave-private LDS (the LDS-DMA kernels, whose LDS is free after the main loop).  The store tail
// of a tile is bound by vector-memory INSTRUCTIONS whose lanes scatter over many rows (a C^T accumulator tile stores 32 rows x 32 B
// per instruction: 256 KiB of a 256x256 tile took 33 us); transposed through LDS every instruction covers whole row segments.
// epilogue_row4: one lane owns 4 consecutive columns (col % 4 == 0) of one row; needs x.vec (16-byte aligned rows, N % 4 == 0).
// For a P8 result adjacent lanes (col, col+4 of one 8-group) trade halves so that each writes one 16-byte hi or lo chunk; all 64
// lanes must be active when it is called.
template <bool GUARD = true>
__device__ __forceinline__ void epilogue_row4(const GemmArgs& g, const EpiCtx& x, int row, int col, f32x4 v, const f32x4* bpre = nullptr) {
    const bool ok = row < g.M && col < g.N;
    const long crow = ok ? map_row(g.cmap, row) : 0;
    f32x4 b = {0.f, 0.f, 0.f, 0.f}, gv = {1.f, 1.f, 1.f, 1.f}, rv = {0.f, 0.f, 0.f, 0.f};
    if (bpre) b = *bpre;
    else if (x.bias && ok) b = *reinterpret_cast<const f32x4*>(x.bias + col);
    if (g.gate && ok) gv = *reinterpret_cast<const f32x4*>(g.gate + (long)map_row(g.gmap, row) * g.ldg + col);
    if (x.R && ok) rv = *reinterpret_cast<const f32x4*>(x.R + crow * g.ldr + col);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += b[e];
    apply_act4(v, g.act);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float t = v[e];
        if (g.gate) t *= gv[e];
        v[e] = t + rv[e];
    }
    if (g.c_p8) {
        if (GUARD && ok) p8_guard(g.status, v[0], v[1], v[2], v[3], p8_maxbits_of(g.c_exp));
        const float csc = p8_scale_of(g.c_exp);
        f16x4_t hh, ll;
#pragma unroll
        for (int e = 0; e < 4; ++e) { _Float16 a, c; split_f16(v[e] * csc, a, c); hh[e] = a; ll[e] = c; }
        const uint2 hu = __builtin_bit_cast(uint2, hh), lu = __builtin_bit_cast(uint2, ll);
        const bool odd = (col & 4) != 0;                      // second half of the 8-group: keeps lo, gives hi away
        const unsigned sx = odd ? hu.x : lu.x, sy = odd ? hu.y : lu.y;
        const unsigned rx = __builtin_amdgcn_mov_dpp(sx, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]: the neighbour lane's value
        const unsigned ry = __builtin_amdgcn_mov_dpp(sy, 0xB1, 0xF, 0xF, true);
        if (ok) {
            unsigned char* o = reinterpret_cast<unsigned char*>(x.C + crow * g.ldc + (col & ~7)) + (odd ? 16 : 0);
            const u32x4_t w = odd ? u32x4_t{rx, ry, lu.x, lu.y} : u32x4_t{hu.x, hu.y, rx, ry};
            *reinterpret_cast<u32x4_t*>(o) = w;
        }
    } else if (ok) {
        *reinterpret_cast<f32x4*>(x.C + crow * g.ldc + col) = v;
    }
}
// NI x NJ accumulator tiles (C^T layout, already scaled) of one wave = rows row0.. (32 NI) x cols col0.. (32 NJ) -> wave-private LDS
// (>= 32 NI * (32 NJ + 4) floats, nobody else touches it) -> row-major epilogue, 64/(8 NJ) rows per instruction.
template <int NI, int NJ, bool GUARD = false>
__device__ __forceinline__ void epilogue_wave_lds(const GemmArgs& g, const EpiCtx& x, float* lds, int row0, int col0, int lane,
                                                  f32x16 (&acc)[NI][NJ]) {
    constexpr int COLS = 32 * NJ, PITCH = COLS + 4, LPR = COLS / 4, RPI = 64 / LPR;
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 t = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                *reinterpret_cast<f32x4*>(lds + (i * 32 + r) * PITCH + j * 32 + 8 * q + 4 * h) = t;
            }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int rr = lane / LPR, cc = (lane % LPR) * 4;
    // a lane's columns are the same in every iteration: its bias run is fetched once, before the first store (vmcnt counts loads and
    // stores together in issue order, so a load inside the loop queues behind the previous iterations' stores)
    f32x4 bpre = {0.f, 0.f, 0.f, 0.f};
    if (x.bias && col0 + cc < g.N) bpre = *reinterpret_cast<const f32x4*>(x.bias + col0 + cc);
    // four rows in flight per trip: fully unrolled (32 trips of the 256x256 kernel) the epilogue alone was 250 KB of code
#pragma unroll 4
    for (int it = 0; it < 32 * NI / RPI; ++it) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(lds + (it * RPI + rr) * PITCH + cc);
        epilogue_row4<GUARD>(g, x, row0 + it * RPI + rr, col0 + cc, v, &bpre);      // (GUARD: the mid-grid kernel of the AR / VAE body; the large-grid kernels leave the range guard to their consumers)
    }
}

// split-K partial slab of the same tile: raw sums, row-major [M][N]
__device__ __forceinline__ void partial_tile32(const GemmArgs& g, float* P, int row, int col0, int h, const f32x16& v) {
    if (row >= g.M) return;
    float* p = P + (long)row * g.N;
    if (((g.N | (int)((unsigned long long)P >> 2)) & 3) == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = col0 + 8 * q + 4 * h;
            if (c < g.N) { const f32x4 o = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]}; *reinterpret_cast<f32x4*>(p + c) = o; }
        }
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int col = col0 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (col < g.N) p[col] = v[e];
        }
    }
}

// The value of lane (lane ^ MASK), MASK a power of two: what __shfl_xor(v, MASK, 64) returns, without the LDS crossbar.  hipcc turns
// __shfl_xor into ds_bpermute_b32 (an LDS-pipe instruction: address VALU + ~100 cycles + an lgkmcnt wait, and the butterflies are
// dependent chains of them); here masks 1 / 2 / 8 are one DPP move (quad permutes, row rotate by 8), 4 is two (half-row mirror, then
// quad reverse: 7 - l = l ^ 7, ^ 3 = l ^ 4), 16 / 32 are gfx950's v_permlane16_swap / v_permlane32_swap of the value with itself plus a
// select.  Same lanes paired in the same order as before: sums and maxima keep their bits.
template <int MASK>
__device__ __forceinline__ float lane_xor(float v) {
    static_assert(MASK == 1 || MASK == 2 || MASK == 4 || MASK == 8 || MASK == 16 || MASK == 32, "power of two below 64");
    const int u = __builtin_bit_cast(int, v);
    if constexpr (MASK == 1) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(u, u, 0xB1, 0xf, 0xf, false));          // quad_perm [1,0,3,2]
    else if constexpr (MASK == 2) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(u, u, 0x4E, 0xf, 0xf, false));     // quad_perm [2,3,0,1]
    else if constexpr (MASK == 4) {
        const int t = __builtin_amdgcn_update_dpp(u, u, 0x141, 0xf, 0xf, false);                                                    // row_half_mirror
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(t, t, 0x1B, 0xf, 0xf, false));                                  // quad_perm [3,2,1,0]
    } else if constexpr (MASK == 8) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(u, u, 0x128, 0xf, 0xf, false));  // row_ror:8
    else if constexpr (MASK == 16) {
        // swap(a = v, b = v): a' = rows [v0, v0, v2, v2], b' = rows [v1, v1, v3, v3]; an even row takes its neighbour from b', an odd one from a'
        const auto sw = __builtin_amdgcn_permlane16_swap((unsigned)u, (unsigned)u, false, false);
        return __builtin_bit_cast(float, (__lane_id() & 16) ? sw[0] : sw[1]);
    } else {
        // swap(a = v, b = v): a' = [v.lo, v.lo], b' = [v.hi, v.hi]
        const auto sw = __builtin_amdgcn_permlane32_swap((unsigned)u, (unsigned)u, false, false);
        return __builtin_bit_cast(float, (__lane_id() & 32) ? sw[0] : sw[1]);
    }
}
// sum over aligned groups of G consecutive lanes (G a power of two), butterfly from G / 2 down to 1 (the order of the former
// `for (o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o)` loops)
template <int G>
__device__ __forceinline__ float group_sum(float v) {
    if constexpr (G >= 64) v += lane_xor<32>(v);
    if constexpr (G >= 32) v += lane_xor<16>(v);
    if constexpr (G >= 16) v += lane_xor<8>(v);
    if constexpr (G >= 8) v += lane_xor<4>(v);
    if constexpr (G >= 4) v += lane_xor<2>(v);
    if constexpr (G >= 2) v += lane_xor<1>(v);
    return v;
}
// full-wave (64 lanes) butterfly reductions
__device__ __forceinline__ float wave_sum(float v) {
    v += lane_xor<32>(v); v += lane_xor<16>(v); v += lane_xor<8>(v); v += lane_xor<4>(v); v += lane_xor<2>(v); v += lane_xor<1>(v);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, lane_xor<32>(v)); v = fmaxf(v, lane_xor<16>(v)); v = fmaxf(v, lane_xor<8>(v));
    v = fmaxf(v, lane_xor<4>(v)); v = fmaxf(v, lane_xor<2>(v)); v = fmaxf(v, lane_xor<1>(v));
    return v;
}

}  // namespace artalk
