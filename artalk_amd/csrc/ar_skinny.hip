// Skinny split GEMM for the 1- and 5-token scale steps of the AR decoder (reference app/transformer.py:30-43 at M = clips x 1 / 5 rows).
//
// At these sizes a block's four linear layers are pure latency: the tiled kernel needs a split over K to find parallelism, the split
// needs a reduce pass, and the AdaLN-modulated LayerNorm in front of q|k|v and of the FFN is a launch of its own - nine dependent
// launches of 5-9 us per block.  Here one workgroup owns a 16-row x 16-column sliver of the result over the WHOLE K:
//   * wave w takes K slice [192 w, 192 w + 192) (4 waves for K = 768, 16 for K = 3072), all of its operand loads - 12 x 16 bytes of
//     the weight sliver, 12 of the activation rows (+ 24 of the AdaLN scale / shift rows) per lane - are issued before anything is
//     waited for, so a launch costs ONE memory latency; operands go straight to registers in the v_mfma_f32_16x16x32_f16 fragment
//     layout (lane = row & 15 | k-group << 4: a lane's 8 consecutive k are exactly one 32-byte [8 hi | 8 lo] group of the P8 format);
//   * LN mode: the activation rows are the fp32 residual stream; every workgroup recomputes the row statistics of its 16 rows
//     (12 K floats - nothing) and normalises, modulates and splits its slice in registers: the LayerNorm launch and the xmod round
//     trip disappear;
//   * the waves' partial 16x16 tiles are added through LDS in wave order (deterministic) and wave 0 applies bias, activation, gate,
//     residual and stores (fp32 with a row map, or P8).
// Same three products per fp32 product, same operand scaling and the same LayerNorm formula as the tiled path (gemm_f16s.hip,
// norm.hip); only the order of the K summation differs.
//
// MEASURED (round 2, MI355X, 16 clips per group): parity-exact on every golden, but NOT faster - so the engine leaves it off
// (ARTALK_SKINNY_MAX_M=0).  A 16-column sliver puts only N / 16 workgroups to work (48 for the projection and FFN-out), and one
// CU pulls its 200-400 KB of operands at 25-50 GB/s: 6.8 us (K = 768, P8 input), 9.3-10.2 us (LayerNorm fused: x, scale and shift
// rows re-read by every sliver), 15 us (K = 3072 on 16 waves) against 5-9 us per launch of the split tiled path, whose K split
// spreads the weight stream over 4-8 times as many CUs.  1-token step 754 -> 663 us, 5-token step 843 -> 1003 us per 12 blocks.
#include "common.h"

namespace artalk {

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

struct SkinnyArgs {
    const unsigned char* A;      // P8 mode: activation rows in the P8 format (row pitch lda floats); LN mode: nullptr
    const float* X;              // LN mode: fp32 rows (row pitch ldx)
    long lda, ldx;
    const float* scale; const float* shift; long ldm; RowMap mmap; float eps;     // AdaLN rows of the table (LN mode)
    const unsigned char* Wp; long ldw;
    const float* bias;
    float* C; long ldc; RowMap cmap;
    const float* gate; long ldg; RowMap gmap;
    const float* R; long ldr;
    int M, N, K, act, c_p8;
    int* status;
};

template <int NW, bool LN>
__global__ __launch_bounds__(NW * 64) void ar_skinny_kernel(const SkinnyArgs a) {
    constexpr int KS = 6;                         // K steps of 32 per wave: 192 columns
    __shared__ __attribute__((aligned(16))) float red[NW][64][4];
    __shared__ float stat[2][NW][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i16 = lane & 15, kg = lane >> 4;
    const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
    const int mrow = min(m0 + i16, a.M - 1);      // rows past M re-read the last row (never stored)
    const long koff = (long)wave * 192 + kg * 8;  // this lane's first k of K step 0

    // ---- every load of this lane, before any wait ----
    u32x4_t wq[KS][2];
    {
        const unsigned char* wp = a.Wp + ((long)(n0 + i16) * a.ldw + koff) * 4;
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            wq[j][0] = *reinterpret_cast<const u32x4_t*>(wp + j * 128);
            wq[j][1] = *reinterpret_cast<const u32x4_t*>(wp + j * 128 + 16);
        }
    }
    u32x4_t aq[KS][2];                            // P8 mode: [hi | lo]; LN mode: 8 fp32 values as raw bits
    {
        const unsigned char* ap = LN ? reinterpret_cast<const unsigned char*>(a.X + (long)mrow * a.ldx + koff)
                                     : a.A + ((long)mrow * a.lda + koff) * 4;
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            aq[j][0] = *reinterpret_cast<const u32x4_t*>(ap + j * 128);
            aq[j][1] = *reinterpret_cast<const u32x4_t*>(ap + j * 128 + 16);
        }
    }
    f32x4 sc[LN ? KS : 1][2], sh[LN ? KS : 1][2];
    if constexpr (LN) {
        const long mr = map_row(a.mmap, mrow);
        const float* sp = a.scale + mr * a.ldm + koff;
        const float* hp = a.shift + mr * a.ldm + koff;
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            sc[j][0] = *reinterpret_cast<const f32x4*>(sp + j * 32);
            sc[j][1] = *reinterpret_cast<const f32x4*>(sp + j * 32 + 4);
            sh[j][0] = *reinterpret_cast<const f32x4*>(hp + j * 32);
            sh[j][1] = *reinterpret_cast<const f32x4*>(hp + j * 32 + 4);
        }
    }

    __builtin_amdgcn_sched_barrier(0);            // nothing above may sink below: the loads are all in flight before the first wait

    f16x8_t ah[KS], al[KS];
    if constexpr (LN) {
        // row statistics over all of K: this lane's 48 values -> the 4 k-groups of the row -> the NW waves (fixed order)
        float x[KS][8];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < KS; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) { x[j][e] = __uint_as_float(aq[j][e >> 2][e & 3]); s += x[j][e]; }
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if (kg == 0) stat[0][wave][i16] = s;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) tot += stat[0][w][i16];
        const float mean = tot * (1.0f / (NW * 192));
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < KS; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = x[j][e] - mean; q += d * d; }
        q += __shfl_xor(q, 16, 64);
        q += __shfl_xor(q, 32, 64);
        if (kg == 0) stat[1][wave][i16] = q;
        __syncthreads();
        float qt = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) qt += stat[1][w][i16];
        const float rstd = 1.0f / sqrtf(qt * (1.0f / (NW * 192)) + a.eps);
        // y = LN(x) * (1 + scale) + shift (norm.hip), then the P8 split of common.h
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            float t[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = (x[j][e] - mean) * rstd * (sc[j][e >> 2][e & 3] + 1.0f) + sh[j][e >> 2][e & 3];
            p8_guard(a.status, t[0], t[1], t[2], t[3]);
            p8_guard(a.status, t[4], t[5], t[6], t[7]);
#pragma unroll
            for (int e = 0; e < 8; ++e) { _Float16 hi, lo; split_f16(t[e] * kActScale, hi, lo); ah[j][e] = hi; al[j][e] = lo; }
        }
    } else {
#pragma unroll
        for (int j = 0; j < KS; ++j) { ah[j] = __builtin_bit_cast(f16x8_t, aq[j][0]); al[j] = __builtin_bit_cast(f16x8_t, aq[j][1]); }
    }

    // ---- C^T[n][m] += W[n][k] A[m][k]: weight fragment = A operand (gemm_f16s.hip's convention) ----
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < KS; ++j) {
        const f16x8_t wh = __builtin_bit_cast(f16x8_t, wq[j][0]), wl = __builtin_bit_cast(f16x8_t, wq[j][1]);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, ah[j], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, ah[j], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, al[j], acc, 0, 0, 0);
    }
    *reinterpret_cast<f32x4*>(&red[wave][lane][0]) = acc;
    __syncthreads();
    if (wave != 0) return;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < NW; ++w) v += *reinterpret_cast<const f32x4*>(&red[w][lane][0]);

    // ---- epilogue: lane (i16, kg) holds row m0 + i16, columns n0 + 4 kg .. + 3 ----
    const int m = m0 + i16, c = n0 + 4 * kg;
    if (m >= a.M) return;
    f32x4 b = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) b = *reinterpret_cast<const f32x4*>(a.bias + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = v[e] * kOutScale + b[e];
    apply_act4(v, a.act);
    const long crow = map_row(a.cmap, m);
    if (a.gate) {
        const f32x4 gv = *reinterpret_cast<const f32x4*>(a.gate + (long)map_row(a.gmap, m) * a.ldg + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= gv[e];
    }
    if (a.R) {
        const f32x4 rv = *reinterpret_cast<const f32x4*>(a.R + crow * a.ldr + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += rv[e];
    }
    if (a.c_p8) store_p8x4(a.C + crow * a.ldc, c, v[0], v[1], v[2], v[3], a.status);
    else *reinterpret_cast<f32x4*>(a.C + crow * a.ldc + c) = v;
}

// g: the linear layer as the tiled path would get it (A in P8 unless ln; Wp, bias, C / cmap, gate / gmap, R, act, c_p8, status);
// ln: the AdaLN-modulated LayerNorm whose output the layer reads (its X is the layer's input, Y is not written), or nullptr.
bool ar_skinny_eligible(const GemmArgs& g, const LnArgs* ln) {
    auto al16 = [](const void* p) { return ((unsigned long long)p & 15) == 0; };
    if (!g.Wp || g.batch != 1 || g.amode != 0 || g.ngrp != 0 || g.M <= 0) return false;
    if (!(g.K == 768 || (g.K == 3072 && !ln)) || g.N % 16 != 0 || g.ldw % 8 != 0) return false;
    if (!al16(g.Wp) || !al16(g.C) || (g.ldc % 4) != 0 || (g.bias && !al16(g.bias))) return false;
    if (g.gate && (!al16(g.gate) || (g.ldg % 4) != 0)) return false;
    if (g.R && (!al16(g.R) || (g.ldr % 4) != 0)) return false;
    if (g.c_p8 && (g.ldc % 8) != 0) return false;
    if (ln) {
        return ln->D == g.K && ln->M == g.M && ln->scale && ln->shift && !ln->w && ln->act == ACT_NONE && al16(ln->X) && (ln->ldx % 4) == 0 &&
               al16(ln->scale) && al16(ln->shift) && (ln->ldm % 4) == 0;
    }
    return g.a_packed && al16(g.A) && (g.lda % 8) == 0;
}

void launch_ar_skinny(const GemmArgs& g, const LnArgs* ln, hipStream_t s) {
    SkinnyArgs a;
    a.A = ln ? nullptr : reinterpret_cast<const unsigned char*>(g.A);
    a.X = ln ? ln->X : nullptr;
    a.lda = g.lda; a.ldx = ln ? ln->ldx : 0;
    a.scale = ln ? ln->scale : nullptr; a.shift = ln ? ln->shift : nullptr; a.ldm = ln ? ln->ldm : 0;
    a.mmap = ln ? ln->mmap : rowmap_identity(); a.eps = ln ? ln->eps : 0.f;
    a.Wp = reinterpret_cast<const unsigned char*>(g.Wp); a.ldw = g.ldw;
    a.bias = g.bias; a.C = g.C; a.ldc = g.ldc; a.cmap = g.cmap;
    a.gate = g.gate; a.ldg = g.ldg; a.gmap = g.gmap; a.R = g.R; a.ldr = g.ldr;
    a.M = g.M; a.N = g.N; a.K = g.K; a.act = g.act; a.c_p8 = g.c_p8; a.status = g.status;
    const dim3 grid(g.N / 16, (g.M + 15) / 16);
    if (ln) hipLaunchKernelGGL((ar_skinny_kernel<4, true>), grid, dim3(256), 0, s, a);
    else if (g.K == 768) hipLaunchKernelGGL((ar_skinny_kernel<4, false>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((ar_skinny_kernel<16, false>), grid, dim3(1024), 0, s, a);
}

}  // namespace artalk
