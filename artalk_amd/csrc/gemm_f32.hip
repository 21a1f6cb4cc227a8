// fp32 GEMM on the gfx950 matrix cores: C = epilogue(A[M,K] * W[N,K]^T).
//
// v_mfma_f32_32x32x2_f32 is exact f32 (a k-ordered fmaf chain), which is what lets this path meet the
// reference's bit-level decisions.  Both operands are K-contiguous (activations
// row-major, weights in torch's [out,in] layout) so both tiles are staged the same way:
// global (float4, 128-B row segments) -> registers -> LDS [rows][32+4] (the +4 pad makes the
// ds_read_b128 fragment reads conflict-free), double-buffered, one barrier per 32-deep K step.
// Each lane half h holds k in [16h, 16h+16) of the K step: MFMA sums over the two halves, and the
// k order inside a step is free as long as A and B agree.
//
// The same kernel serves the wav2vec2 stride-2 convolutions (channels-last activations make the
// window of output t a contiguous K=k*512 run at row 2t: a plain GEMM with lda = 2*512) and, with
// amode=1, the grouped positional convolution (window rows gathered with zero padding).
#include "common.h"

namespace artalk {


// TAG only separates instantiations: 64x64 launches captured into the AR/VAE hipGraph use TAG=1 so that the eager 64x64
// launches (TAG=0: wav2vec2 encoder + AdaLN table, the dominant kernel of the path) form one kernel symbol whose every launch is
// bracketed by HIP events in bench.py and listed as one row by rocprofv3.
template <int BM, int BN, int WM, int WN, int BK, int AMODE, int TAG = 0>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(const GemmArgs g) {
    constexpr int LDS_LD = BK + 4;                 // floats; the +4 pad keeps the ds_read_b128 fragment reads conflict-free
    constexpr int TPR = BK / 4, RPP = 256 / TPR;   // staging: threads per row, rows per pass
    constexpr int NQ = BK / 8;                     // 16-byte chunks of a fragment per lane (lane half h holds k in [h*BK/2, (h+1)*BK/2))
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int A_LD = BM / RPP, B_LD = BN / RPP;
    static_assert(WM * WN == 4 && TM >= 1 && TN >= 1 && A_LD >= 1 && B_LD >= 1, "tile config");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                     // [2][BM][LDS_LD]
    float* Bs = smem + 2 * BM * LDS_LD;   // [2][BN][LDS_LD]

    const int tid = threadIdx.x;
    // Tile order.  Workgroups are dealt round-robin over the 8 XCDs (each with a private 4 MiB L2), so first give each
    // XCD a contiguous run of the tile sequence (bijective remap), then walk the sequence in groups of GM row-tiles,
    // column-major inside a group: the GM A-panels (GM*BM*K floats) stay L2-resident while the W panels stream past once
    // per group instead of once per row-tile.  Placement only affects speed, never results.
    const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
    int tm, tn;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        const int idx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        constexpr int GM = (BM >= 128) ? 4 : 8;
        const int width = GM * tiles_n;
        const int group = idx / width, first_m = group * GM;
        const int gsz = min(tiles_m - first_m, GM);
        const int in_g = idx - group * width;
        tn = in_g / gsz;
        tm = first_m + (in_g - tn * gsz);
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int z = blockIdx.z;
    const float* __restrict__ A = g.A + z * g.sA;
    const float* __restrict__ W = g.W + z * g.sW;

    f32x4 ra[A_LD], rb[B_LD];
    const int lrow = tid / TPR, lc4 = (tid % TPR) * 4;   // TPR threads cover one BK-float row segment

    auto gload = [&](int kt) {
        const int k = kt * BK + lc4;
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            const int gm = m0 + lrow + i * RPP;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (gm < g.M) {
                if (AMODE == 0) {
                    v = *reinterpret_cast<const f32x4*>(A + (long)gm * g.lda + k);
                } else {
                    const int c = gm / g.pc_tstride, t = gm - c * g.pc_tstride;
                    const int tap = k / g.pc_cin, ci = k - tap * g.pc_cin;
                    const int ts = t + tap - g.pc_pad;
                    if (ts >= 0 && ts < g.pc_T)
                        v = *reinterpret_cast<const f32x4*>(A + ((long)c * g.pc_tstride + ts) * g.lda + ci);
                }
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            const int gn = n0 + lrow + i * RPP;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (gn < g.N) v = *reinterpret_cast<const f32x4*>(W + (long)gn * g.ldw + k);
            rb[i] = v;
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_LD; ++i)
            *reinterpret_cast<f32x4*>(As + (buf * BM + lrow + i * RPP) * LDS_LD + lc4) = ra[i];
#pragma unroll
        for (int i = 0; i < B_LD; ++i)
            *reinterpret_cast<f32x4*>(Bs + (buf * BN + lrow + i * RPP) * LDS_LD + lc4) = rb[i];
    };

    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // K % 32 == 0 is required by the callers, so K / BK is exact for BK = 16 and 32; split-K: this workgroup owns K steps [kt0, kt1)
    const int nk_all = g.K / BK;
    const int kt0 = (int)((long)nk_all * blockIdx.y / g.splitk), nk = (int)((long)nk_all * (blockIdx.y + 1) / g.splitk);
    gload(kt0);
    lstore(0);
    __syncthreads();
    for (int kt = kt0; kt < nk; ++kt) {
        const int buf = (kt - kt0) & 1;
        if (kt + 1 < nk) gload(kt + 1);
        const float* as = As + (buf * BM + wm * (BM / WM) + r) * LDS_LD + h * (BK / 2);
        const float* bs = Bs + (buf * BN + wn * (BN / WN) + r) * LDS_LD + h * (BK / 2);
        f32x4 a[TM][NQ], b[TN][NQ];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int q = 0; q < NQ; ++q) a[i][q] = *reinterpret_cast<const f32x4*>(as + i * 32 * LDS_LD + q * 4);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < NQ; ++q) b[j][q] = *reinterpret_cast<const f32x4*>(bs + j * 32 * LDS_LD + q * 4);
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j][q][e], a[i][q][e], acc[i][j], 0, 0, 0);   // C^T: epilogue_tile32
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: the weight fragment is the A operand, so lane (r, h) holds 4 x 4 consecutive columns of row r (common.h) ----
    if (g.splitk > 1) {   // raw partial sums; bias/act/gate/residual are applied by splitk_reduce_kernel
        float* __restrict__ P = g.partial + (long)blockIdx.y * g.M * g.N;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) partial_tile32(g, P, m0 + wm * (BM / WM) + i * 32 + r, n0 + wn * (BN / WN) + j * 32, h, acc[i][j]);
        return;
    }
    const EpiCtx epi = make_epi(g, g.bias ? g.bias + z * g.sBias : nullptr, g.C + z * g.sC, g.R ? g.R + z * g.sR : nullptr);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) epilogue_tile32<false>(g, epi, m0 + wm * (BM / WM) + i * 32 + r, n0 + wn * (BN / WN) + j * 32, h, acc[i][j]);
}

template <int BM, int BN, int WM, int WN, int BK = 32>
static void launch_cfg(const GemmArgs& g, hipStream_t s) {
    const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
    const size_t lds = 2 * (BM + BN) * (BK + 4) * sizeof(float);
    dim3 grid(tiles, g.splitk, g.batch);
    if (g.amode == 0 && g.graph_tag && BM == 64 && BN == 64)
        ARTALK_LAUNCH((gemm_f32_kernel<BM, BN, WM, WN, BK, 0, 1>), grid, dim3(256), lds, s, g);
    else if (g.amode == 0)
        ARTALK_LAUNCH((gemm_f32_kernel<BM, BN, WM, WN, BK, 0>), grid, dim3(256), lds, s, g);
    else
        ARTALK_LAUNCH((gemm_f32_kernel<BM, BN, WM, WN, BK, 1>), grid, dim3(256), lds, s, g);
}

// Tile choice, from tools/gemm_bench.py on MI355X (profiles/r01_gemm_bench.log).  A register-only MFMA loop sustains
// 148 TF/s with 4 independent accumulators per wave and 130 TF/s with one dependent chain (profiles/r01_mfma_peak.log),
// so the big shapes want 128x128 (4 accumulators) and enough workgroups per CU to hide staging (BK=16: 3 per CU).
//   4: 128x128, BK=16   grids of >= 1024 tiles: wav2vec2 encoder, conv-as-GEMM, AdaLN table (the dominant kernel)   113-125 TF/s
//   2: 64x64,   BK=32   smaller grids with M > 32 (AR / VAE steps; 4 workgroups/CU)                                  50-100 TF/s
//   1: 128x64,  BK=32   grouped positional conv (N = 64 per group)
//   3: 32x128,  BK=32   M <= 32 (first scale step at small batch)
//   0,5,6,7: kept for tuning (128x128 BK=32; 128x64, 64x128, 64x64 with BK=16)
int gemm_config(const GemmArgs& g) {
    if (g.force_cfg >= 0) return g.force_cfg;
    if (g.amode == 1) return 1;
    const long t128 = (long)((g.M + 127) / 128) * ((g.N + 127) / 128);
    if (t128 >= 1024) return 4;
    if (g.M > 32) return 2;
    return 3;
}

void launch_gemm(const GemmArgs& g, hipStream_t s) {
    if (g.M <= 0 || g.N <= 0) return;
    switch (gemm_config(g)) {
        case 0: launch_cfg<128, 128, 2, 2>(g, s); break;
        case 1: launch_cfg<128, 64, 2, 2>(g, s); break;
        case 2: launch_cfg<64, 64, 2, 2>(g, s); break;
        case 3: launch_cfg<32, 128, 1, 4>(g, s); break;
        // experimental BK=16 variants (more workgroups per CU)
        case 4: launch_cfg<128, 128, 2, 2, 16>(g, s); break;
        case 5: launch_cfg<128, 64, 2, 2, 16>(g, s); break;
        case 6: launch_cfg<64, 128, 2, 2, 16>(g, s); break;
        case 7: launch_cfg<64, 64, 2, 2, 16>(g, s); break;
        default: launch_cfg<32, 128, 1, 4>(g, s); break;
    }
}

}  // namespace artalk

// ---- split-K epilogue pass: C = R + gate * act(sum_y partial[y] + bias), partials added in y order (deterministic)
namespace artalk {
// S is a compile-time constant (1..8; 0 = run-time loop for deeper splits) so that the S slab loads of an element are all issued
// before the first add: the slabs were written a moment ago by workgroups on other XCDs, every load is an L2 miss, and a
// run-time loop paid those latencies one after the other (5 us for a 20-row reduce).
template <int S>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmArgs g) {
    const long total = (long)g.M * g.N;
    const EpiCtx epi = make_epi(g, g.bias, g.C, g.R);
    if (epi.vec && (((unsigned long long)g.partial >> 2) & 3) == 0) {   // 4 consecutive columns per thread, shared epilogue (fp32 or P8 result)
        for (long idx = ((long)blockIdx.x * 256 + threadIdx.x) * 4; idx < total; idx += (long)gridDim.x * 1024) {
            const int row = (int)(idx / g.N), col0 = (int)(idx - (long)row * g.N);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (S > 0) {
                f32x4 t[S > 0 ? S : 1];
#pragma unroll
                for (int y = 0; y < S; ++y) t[y] = *reinterpret_cast<const f32x4*>(g.partial + (long)y * total + idx);
#pragma unroll
                for (int y = 0; y < S; ++y) v += t[y];
            } else {
                for (int y = 0; y < g.splitk; ++y) v += *reinterpret_cast<const f32x4*>(g.partial + (long)y * total + idx);
            }
            epilogue_row4(g, epi, row, col0, v);
        }
        return;
    }
    for (long idx = ((long)blockIdx.x * 256 + threadIdx.x) * 4; idx < total; idx += (long)gridDim.x * 1024) {
        const int row = (int)(idx / g.N), col0 = (int)(idx - (long)row * g.N);
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        const int nv = min(4, (int)(total - idx));
        for (int y = 0; y < g.splitk; ++y) {
            const float* P = g.partial + (long)y * total + idx;
            for (int e = 0; e < nv; ++e) v[e] += P[e];
        }
        for (int e = 0; e < nv; ++e) {
            int rr = row, cc = col0 + e;
            if (cc >= g.N) { rr += cc / g.N; cc %= g.N; }
            float x = v[e] + (g.bias ? g.bias[cc] : 0.f);
            x = apply_act_rt(x, g.act);
            if (g.gate) x *= g.gate[(long)map_row(g.gmap, rr) * g.ldg + cc];
            const long crow = map_row(g.cmap, rr);
            if (g.R) x += g.R[crow * g.ldr + cc];
            g.C[crow * g.ldc + cc] = x;
        }
    }
}
void launch_splitk_reduce(const GemmArgs& g, hipStream_t s) {
    const long total = (long)g.M * g.N;
    const long blocks = (total / 4 + 255) / 256;
    const dim3 grid((unsigned)(blocks < 2048 ? (blocks > 0 ? blocks : 1) : 2048));
    switch (g.splitk) {
        case 2: ARTALK_LAUNCH(splitk_reduce_kernel<2>, grid, dim3(256), 0, s, g); break;
        case 3: ARTALK_LAUNCH(splitk_reduce_kernel<3>, grid, dim3(256), 0, s, g); break;
        case 4: ARTALK_LAUNCH(splitk_reduce_kernel<4>, grid, dim3(256), 0, s, g); break;
        case 6: ARTALK_LAUNCH(splitk_reduce_kernel<6>, grid, dim3(256), 0, s, g); break;
        case 8: ARTALK_LAUNCH(splitk_reduce_kernel<8>, grid, dim3(256), 0, s, g); break;
        default: ARTALK_LAUNCH(splitk_reduce_kernel<0>, grid, dim3(256), 0, s, g); break;
    }
}

// The same pass for a GEMM whose result is the 768-wide residual stream x of an AR block and is followed by the AdaLN-modulated
// LayerNorm of the next GEMM's input (app/transformer.py:35,40; app/models.py:148): one wavefront per row sums the slabs, applies
// bias / gate / residual, writes x, and - still holding the row in registers - writes LN(x) * (1 + scale) + shift for the consumer
// (P8 or fp32).  Replaces splitk_reduce + layernorm (two launches, one extra trip of x through memory) on the small scale steps.
template <int S>
__global__ __launch_bounds__(256) void splitk_reduce_ln768_kernel(const GemmArgs g, const LnArgs ln) {
    constexpr int D = 768, NA = 3;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= g.M) return;
    const long total = (long)g.M * D;
    const long crow = map_row(g.cmap, row);
    const float* gp = g.gate ? g.gate + (long)map_row(g.gmap, row) * g.ldg : nullptr;
    const long mr = map_row(ln.mmap, row);
    const float* sc = ln.scale + mr * ln.ldm;
    const float* sh = ln.shift + mr * ln.ldm;
    f32x4 t[NA][S], bv[NA], gv[NA], rv[NA], scv[NA], shv[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int c = (i * 64 + lane) * 4;
#pragma unroll
        for (int y = 0; y < S; ++y) t[i][y] = *reinterpret_cast<const f32x4*>(g.partial + (long)y * total + (long)row * D + c);
        bv[i] = g.bias ? *reinterpret_cast<const f32x4*>(g.bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        gv[i] = gp ? *reinterpret_cast<const f32x4*>(gp + c) : f32x4{1.f, 1.f, 1.f, 1.f};
        rv[i] = g.R ? *reinterpret_cast<const f32x4*>(g.R + crow * g.ldr + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        scv[i] = *reinterpret_cast<const f32x4*>(sc + c);
        shv[i] = *reinterpret_cast<const f32x4*>(sh + c);
    }
    float v[NA][4];
    float s1 = 0.f;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int y = 0; y < S; ++y) a += t[i][y];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x = apply_act_rt(a[e] + bv[i][e], g.act);
            if (g.gate) x *= gv[i][e];
            v[i][e] = x + rv[i][e];
            s1 += v[i][e];
        }
        const f32x4 o = {v[i][0], v[i][1], v[i][2], v[i][3]};
        *reinterpret_cast<f32x4*>(g.C + crow * g.ldc + (i * 64 + lane) * 4) = o;
    }
    // same arithmetic as layernorm_kernel<768> (norm.hip)
    const float mean = wave_sum(s1) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / D) + ln.eps);
    float* y = ln.Y + (long)row * ln.ldy;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int c = (i * 64 + lane) * 4;
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * (scv[i][e] + 1.0f) + shv[i][e];
        if (ln.out_p8) store_p8x4_pair(y, c, o[0], o[1], o[2], o[3], ln.status, ln.p8_exp);
        else { const f32x4 w = {o[0], o[1], o[2], o[3]}; *reinterpret_cast<f32x4*>(y + c) = w; }
    }
}
bool splitk_reduce_ln_eligible(const GemmArgs& g, const LnArgs& ln) {
    const int S = g.splitk;
    auto al16 = [](const void* p) { return ((unsigned long long)p & 15) == 0; };
    return g.N == 768 && ln.D == 768 && ln.M == g.M && (S == 2 || S == 3 || S == 4 || S == 6 || S == 8) && ln.scale && ln.shift && !ln.w &&
           ln.act == ACT_NONE && g.c_p8 == 0 && (g.ldc % 4) == 0 && (g.ldr % 4) == 0 && (g.ldg % 4) == 0 && (ln.ldm % 4) == 0 && (ln.ldy % 8) == 0 &&
           al16(g.partial) && al16(g.C) && al16(g.R) && al16(g.gate) && al16(g.bias) && al16(ln.scale) && al16(ln.shift) && al16(ln.Y);
}
void launch_splitk_reduce_ln(const GemmArgs& g, const LnArgs& ln, hipStream_t s) {
    const dim3 grid((g.M + 3) / 4);
    switch (g.splitk) {
        case 2: ARTALK_LAUNCH(splitk_reduce_ln768_kernel<2>, grid, dim3(256), 0, s, g, ln); break;
        case 3: ARTALK_LAUNCH(splitk_reduce_ln768_kernel<3>, grid, dim3(256), 0, s, g, ln); break;
        case 4: ARTALK_LAUNCH(splitk_reduce_ln768_kernel<4>, grid, dim3(256), 0, s, g, ln); break;
        case 6: ARTALK_LAUNCH(splitk_reduce_ln768_kernel<6>, grid, dim3(256), 0, s, g, ln); break;
        default: ARTALK_LAUNCH(splitk_reduce_ln768_kernel<8>, grid, dim3(256), 0, s, g, ln); break;
    }
}
int gemm_tile_count(const GemmArgs& g, bool f16s) {
    int bm, bn;
    if (f16s) { if (gemm_f16s_config(g) == 0) { bm = 128; bn = 128; } else { bm = 64; bn = 64; } }
    else switch (gemm_config(g)) {
        case 0: case 4: bm = 128; bn = 128; break;
        case 1: case 5: bm = 128; bn = 64; break;
        case 6: bm = 64; bn = 128; break;
        case 2: case 7: bm = 64; bn = 64; break;
        default: bm = 32; bn = 128; break;
    }
    return ((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn);
}
}  // namespace artalk

// ---- calibration: register-only v_mfma_f32_32x32x2_f32 loop (no memory traffic) on non-trivial data.  Gives the
// fp32-MFMA rate the chip sustains at the clock it holds under this load: the ceiling the GEMM is compared with.
namespace artalk {
template <int NACC>
__global__ __launch_bounds__(256) void mfma_f32_peak_kernel(float* out, int iters, float seed) {
    f32x16 acc[NACC];
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a = seed * (float)(lane + 1) * 1.0009765625f, b = 0.5f + seed * (float)(63 - lane);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32 / NACC; ++u) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
            a = -a * 0.999f; b = b * 1.0001f;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// fp16 matrix cores, register-only, on non-trivial data: the rate (and the clock the chip holds: out[...] also receives the
// in-kernel clock = d s_memtime / d s_memrealtime x 100 MHz) that bounds the f16x3 split GEMM.  SHAPE 0: v_mfma_f32_32x32x16_f16
// (4 accumulators of 16 registers), 1: v_mfma_f32_16x16x32_f16 (16 accumulators of 4 registers): equal cycles per flop.
typedef _Float16 pk_h8 __attribute__((ext_vector_type(8)));
typedef float pk_f4 __attribute__((ext_vector_type(4)));
template <int SHAPE>
__global__ __launch_bounds__(256) void mfma_f16_peak_kernel(float* out, int iters, float seed) {
    const int lane = threadIdx.x & 63;
    pk_h8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(seed * (float)(lane + 1 + e) * 0.013f); b[e] = (_Float16)(0.5f - seed * (float)(63 - lane + e) * 0.011f); }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
    if (SHAPE == 0) {
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
            }
            a[it & 7] = -a[it & 7]; b[(it + 3) & 7] = b[(it + 3) & 7] * (_Float16)0.999f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) sum += acc[i][e];
    } else {
        pk_f4 acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) { const pk_f4 z = {0.f, 0.f, 0.f, 0.f}; acc[i] = z; }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
            }
            a[it & 7] = -a[it & 7]; b[(it + 3) & 7] = b[(it + 3) & 7] * (_Float16)0.999f;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) sum += acc[i][e];
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float v = sum;
    if (threadIdx.x == 0) v = (float)((double)(c1 - c0) / (double)(r1 - r0) * 0.1);     // GHz (s_memrealtime ticks at 100 MHz)
    out[blockIdx.x * 256 + threadIdx.x] = v;
}
// The same two shapes on operands that CHANGE from MFMA to MFMA, as a GEMM's do: 12 + 8 pseudo-random fp16 fragments in registers
// (hi-like values of order 1 and lo-like values of order 2^-11, as the split operands are), every MFMA takes another pair.  The
// constant-operand loops above barely toggle the multipliers' inputs; the clock the chip holds depends on what the pipes switch.
__device__ __forceinline__ unsigned peak_hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int SHAPE>
__global__ __launch_bounds__(256) void mfma_f16_peak_rnd_kernel(float* out, int iters, float seed) {
    const int lane = threadIdx.x & 63;
    pk_h8 a[12], b[8];
#pragma unroll
    for (int f = 0; f < 12; ++f)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const unsigned hsh = peak_hash((unsigned)(lane * 977 + f * 131 + e * 7 + blockIdx.x * 7919) + (unsigned)(seed * 1000.f));
            const float v = ((float)(hsh & 0xffff) / 32768.f - 1.f) * (f % 3 == 2 ? 4.8828125e-4f : 1.f);       // every third fragment lo-like
            a[f][e] = (_Float16)v;
        }
#pragma unroll
    for (int f = 0; f < 8; ++f)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const unsigned hsh = peak_hash((unsigned)(lane * 331 + f * 1009 + e * 13 + blockIdx.x * 104729 + 5));
            const float v = ((float)(hsh & 0xffff) / 32768.f - 1.f) * (f & 1 ? 4.8828125e-4f : 1.f) * 0.03f;
            b[f][e] = (_Float16)v;
        }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
    if (SHAPE == 0) {
        f32x16 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[(i + 3 * u) & 7], a[(i * 5 + u * 7) % 12], acc[i], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) sum += acc[i][e];
    } else {
        pk_f4 acc[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) { const pk_f4 z = {0.f, 0.f, 0.f, 0.f}; acc[i] = z; }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
#pragma unroll
                for (int i = 0; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[(i + 3 * u) & 7], a[(i * 5 + u * 7) % 12], acc[i], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 32; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) sum += acc[i][e];
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float v = sum;
    if (threadIdx.x == 0) v = (float)((double)(c1 - c0) / (double)(r1 - r0) * 0.1);
    out[blockIdx.x * 256 + threadIdx.x] = v;
}
// returns FLOPs issued by one launch; nacc = independent accumulators per wave (1 = one dependent chain, like the 64x64 GEMM);
// nacc 16 / 17: the fp16 kernels above (32x32x16 / 16x16x32)
double launch_mfma_f32_peak(float* out, int blocks, int iters, int nacc, hipStream_t s) {
    if (nacc == 16 || nacc == 17) {
        if (nacc == 16) ARTALK_LAUNCH(mfma_f16_peak_kernel<0>, dim3(blocks), dim3(256), 0, s, out, iters, 0.37f);
        else ARTALK_LAUNCH(mfma_f16_peak_kernel<1>, dim3(blocks), dim3(256), 0, s, out, iters, 0.37f);
        // per iteration and wave: 32 MFMAs of 32x32x16 (32768 flops each) or 64 of 16x16x32 (16384 flops each)
        return (double)blocks * 4 * iters * 32.0 * (32.0 * 32 * 16 * 2);
    }
    if (nacc == 18 || nacc == 19) {      // the fp16 shapes on changing pseudo-random operands
        if (nacc == 18) ARTALK_LAUNCH(mfma_f16_peak_rnd_kernel<0>, dim3(blocks), dim3(256), 0, s, out, iters, 0.37f);
        else ARTALK_LAUNCH(mfma_f16_peak_rnd_kernel<1>, dim3(blocks), dim3(256), 0, s, out, iters, 0.37f);
        return (double)blocks * 4 * iters * 32.0 * (32.0 * 32 * 16 * 2);
    }
    if (nacc == 1) ARTALK_LAUNCH(mfma_f32_peak_kernel<1>, dim3(blocks), dim3(256), 0, s, out, iters, 0.37f);
    else ARTALK_LAUNCH(mfma_f32_peak_kernel<4>, dim3(blocks), dim3(256), 0, s, out, iters, 0.37f);
    return (double)blocks * 4 /*waves*/ * iters * 32.0 /*mfma per iter*/ * (32.0 * 32 * 2 * 2);
}
}  // namespace artalk
