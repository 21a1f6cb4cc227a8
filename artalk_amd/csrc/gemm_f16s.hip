// fp32-accurate GEMM on the fp16 matrix cores by operand splitting ("f16x3"): C = epilogue(A[M,K] * W[N,K]^T).
//
// Every fp32 operand x is scaled by a power of two (activations 16, weights 256: common.h) and represented as hi + lo with
// hi = f16(Sx) and lo = f16(Sx - hi): 22 significand bits (the residual may be an fp16 subnormal, which the gfx950 MFMA
// honours).  A product a*b then needs three exact fp16 products  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi, all accumulated into ONE
// fp32 accumulator by v_mfma_f32_32x32x16_f16 (both halves share a scale); the dropped a_lo*b_lo term is 2^-22 relative and
// the epilogue removes the operand scales with one exact multiply.  Measured on random data (K = 1024): error a few 1e-7 of
// the output scale, at the accumulation-order noise of any fp32 GEMM, and every reference golden stays decision-exact
// (tests/test_e2e_gpu.py runs both precision modes).  Three fp16 MFMAs cover 16 k in 96 cycles where v_mfma_f32_32x32x2_f32
// needs 512: the MFMA time drops 5.3x and the kernels become staging-bound, which is why the accumulator count matters: one
// accumulator per output tile (not main + cross) lets a wave own twice the output and halves the operand bytes per flop.
// fp16 range: |x| < 4094 for activations, < 255 for weights (all GEMM inputs of the path are LayerNorm/GELU/attention outputs
// or weights); an overflow becomes inf and is caught by the model's status word (exact-f32 re-run).
//
// Memory layout is unchanged: activations stay fp32 in HBM and are split while they are staged into LDS (or arrive already
// split from their producer kernel); weights are pre-split once at load into the "P8" format (every 8 elements -> 16 bytes of
// hi halves + 16 bytes of lo halves), so a packed matrix has the same size, pitch and 16-byte loads as the fp32 one and a
// 16-byte chunk is an MFMA fragment.  LDS rows (register-staged kernel) hold the hi plane (BK halves) followed by the lo
// plane, padded to 144 bytes (conflict-free ds_read_b128 for the 32x32x16 operand map: lane (r,h) holds k = 8h..8h+7).
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include "common.h"

namespace artalk {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split_f32x4(const f32x4 x, const float scale, u32x2& hi, u32x2& lo) {
    f16x4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) { _Float16 a, b; split_f16(x[e] * scale, a, b); h[e] = a; l[e] = b; }
    hi = __builtin_bit_cast(u32x2, h);
    lo = __builtin_bit_cast(u32x2, l);
}
// Register-staged kernel: thread (row, chunk c = 0..7) of a K tile of 32 handles one 16-byte chunk.  For an fp32 operand the
// chunk is elements 4c..4c+3 (split here: 8 B hi + 8 B lo); for a P8 operand chunk c is already a fragment: group c>>1,
// hi (c even) or lo (c odd), copied as is.  LDS row: [hi plane: 4 groups x 16 B][lo plane: 4 groups x 16 B] + 16 B pad.
__device__ __forceinline__ int p8_lds_offset(int c) { return (c & 1) * 64 + (c >> 1) * 16; }

// fp32 -> "P8" split format, 8 elements per thread: each group of 8 consecutive elements becomes 32 bytes
// [8 x f16 hi][8 x f16 lo] (same size as the 8 floats it replaces, so a P8 matrix keeps the fp32 matrix's pitch and
// indexing).  A 16-byte chunk of a P8 row is therefore directly an MFMA operand fragment (k = 8h .. 8h+7).
__global__ __launch_bounds__(256) void pack_split_kernel(const float* __restrict__ w, unsigned int* __restrict__ out, long n, float scale,
                                                         int* __restrict__ status, unsigned int maxbits) {
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += (long)gridDim.x * 2048) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(w + i), b = *reinterpret_cast<const f32x4*>(w + i + 4);
        if (status) {      // activations only (scale = 2^p8_exp): range guard of the P8 format
            p8_guard(status, a[0], a[1], a[2], a[3], maxbits);
            p8_guard(status, b[0], b[1], b[2], b[3], maxbits);
        }
        u32x2 h0, l0, h1, l1;
        split_f32x4(a, scale, h0, l0);
        split_f32x4(b, scale, h1, l1);
        u32x4 hi = {h0[0], h0[1], h1[0], h1[1]}, lo = {l0[0], l0[1], l1[0], l1[1]};
        *reinterpret_cast<u32x4*>(out + i) = hi;
        *reinterpret_cast<u32x4*>(out + i + 4) = lo;
    }
}
void launch_pack_split(const float* w, unsigned int* out, long n, bool is_weight, hipStream_t s, int* status, int p8_exp) {
    if (n <= 0) return;
    if (is_weight) status = nullptr;
    const long blocks = (n / 8 + 255) / 256;   // n % 8 == 0 (every packed tensor has an inner dimension that is a multiple of 32)
    ARTALK_LAUNCH(pack_split_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, s, w, out, n, is_weight ? kWScale : p8_scale_of(p8_exp), status, p8_maxbits_of(p8_exp));
}

// AMODE 1: grouped positional-conv window gather (see gemm_f32.hip), grid.z = group, fp32 A only.
template <int BM, int BN, int WM, int WN, int APK, int TAG, int AMODE = 0>
__global__ __launch_bounds__(256, 2) void gemm_f16s_kernel(const GemmArgs g) {
    constexpr int BK = 32;
    constexpr int ROWB = 144;                      // bytes per LDS row: 64 (hi) + 64 (lo) + 16 pad
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int A_LD = BM / 32, B_LD = BN / 32;  // 8 threads stage one 32-element row segment; 32 rows per pass
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_h[];
    unsigned char* As = smem_h;                    // [2][BM][ROWB]
    unsigned char* Bs = smem_h + 2 * BM * ROWB;    // [2][BN][ROWB]

    const int tid = threadIdx.x;
    const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
    int tm, tn;
    {   // XCD-contiguous, grouped column-major tile order (see gemm_f32.hip)
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
    const unsigned int* __restrict__ Wp = g.Wp + z * g.sW;

    u32x4 ra[A_LD];   // fp32 bits (APK=0: split while staging) or packed (hi | lo<<16) words (APK=1: producer already split)
    u32x4 rb[B_LD];
    const int lrow = tid >> 3, lc4 = (tid & 7) * 4;

    auto gload = [&](int kt) {
        const int k = kt * BK + lc4;
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            const int gm = m0 + lrow + i * 32;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (gm < g.M) {
                if (AMODE == 0) {
                    v = *reinterpret_cast<const u32x4*>(A + (long)gm * g.lda + k);
                } else {
                    const int c = gm / g.pc_tstride, t = gm - c * g.pc_tstride;
                    const int tap = k / g.pc_cin, ci = k - tap * g.pc_cin;
                    const int ts = t + tap - g.pc_pad;
                    if (ts >= 0 && ts < g.pc_T) v = *reinterpret_cast<const u32x4*>(A + ((long)c * g.pc_tstride + ts) * g.lda + ci);
                }
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            const int gn = n0 + lrow + i * 32;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (gn < g.N) v = *reinterpret_cast<const u32x4*>(Wp + (long)gn * g.ldw + k);
            rb[i] = v;
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            unsigned char* p = As + (buf * BM + lrow + i * 32) * ROWB;
            if (APK) {
                *reinterpret_cast<u32x4*>(p + p8_lds_offset(tid & 7)) = ra[i];
            } else {
                u32x2 hi, lo;
                {
                    const f32x4 av = __builtin_bit_cast(f32x4, ra[i]);
                    p8_guard(g.status, av[0], av[1], av[2], av[3], p8_maxbits_of(g.a_exp));
                }
                split_f32x4(__builtin_bit_cast(f32x4, ra[i]), p8_scale_of(g.a_exp), hi, lo);
                *reinterpret_cast<u32x2*>(p + lc4 * 2) = hi;
                *reinterpret_cast<u32x2*>(p + lc4 * 2 + 64) = lo;
            }
        }
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            *reinterpret_cast<u32x4*>(Bs + (buf * BN + lrow + i * 32) * ROWB + p8_lds_offset(tid & 7)) = rb[i];
        }
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

    const int nk_all = g.K / BK;   // split-K: this workgroup owns K steps [kt0, nk)
    const int kt0 = (int)((long)nk_all * blockIdx.y / g.splitk), nk = (int)((long)nk_all * (blockIdx.y + 1) / g.splitk);
    gload(kt0);
    lstore(0);
    __syncthreads();
    for (int kt = kt0; kt < nk; ++kt) {
        const int buf = (kt - kt0) & 1;
        if (kt + 1 < nk) gload(kt + 1);
        const unsigned char* as = As + (buf * BM + wm * (BM / WM) + r) * ROWB + h * 16;
        const unsigned char* bs = Bs + (buf * BN + wn * (BN / WN) + r) * ROWB + h * 16;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {   // two 16-deep MFMA steps per 32-deep K tile
            f16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[i] = *reinterpret_cast<const f16x8*>(as + i * 32 * ROWB + kb * 32);
                al[i] = *reinterpret_cast<const f16x8*>(as + i * 32 * ROWB + kb * 32 + 64);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bh[j] = *reinterpret_cast<const f16x8*>(bs + j * 32 * ROWB + kb * 32);
                bl[j] = *reinterpret_cast<const f16x8*>(bs + j * 32 * ROWB + kb * 32 + 64);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    // weight fragment as the A operand: the accumulator is C^T (common.h, epilogue_tile32)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], ah[i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[j], ah[i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], al[i], acc[i][j], 0, 0, 0);
                }
        }
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }

    if (g.splitk > 1) {   // raw partial sums; the epilogue runs in splitk_reduce_kernel
        float* __restrict__ P = g.partial + (long)blockIdx.y * g.M * g.N;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] *= p8_out_scale_of(g.a_exp);
                partial_tile32(g, P, m0 + wm * (BM / WM) + i * 32 + r, n0 + wn * (BN / WN) + j * 32, h, acc[i][j]);
            }
        return;
    }
    const EpiCtx epi = make_epi(g, g.bias ? g.bias + z * g.sBias : nullptr, g.C + z * g.sC, g.R ? g.R + z * g.sR : nullptr);
    epilogue_tiles<true, true, TM, TN>(g, epi, m0 + wm * (BM / WM) + r, n0 + wn * (BN / WN), h, acc, p8_out_scale_of(g.a_exp));
}

// ------------------------------------------------------------------------------------------------------------------
// LDS-DMA kernels: BOTH operands already in P8 (A written in P8 by its producer kernel), so staging is a pure byte copy and goes
// global -> LDS directly (global_load_lds_dwordx4: no staging registers) behind counted vmcnt waits and ONE raw s_barrier per K step.
//   LDS stage: A rows x 128 B, then W rows x 128 B (BK = 32), rows unpadded (a DMA wave-instruction writes 1 KiB = 8 rows);
//   16-byte chunk c of row r lives at physical chunk c ^ ((r >> 1) & 7): the XOR is applied to the per-lane SOURCE address
//   of the DMA and again on the fragment read (same involution), which makes the ds_read_b128 fragment reads conflict-free.
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// Fragment reads and their waits are hand-placed: while an LDS-DMA is pending hipcc treats it as a FLAT access to both memories
// and turns every wait it inserts for a ds_read result into lgkmcnt(0), which drains the reads issued for the NEXT half step in
// front of MFMAs that do not need them.  The sched_barrier keeps register-only MFMAs from being hoisted above the wait.
template <int N>
__device__ __forceinline__ void wait_lgkmcnt() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <int OFF>
__device__ __forceinline__ f16x8 lds_read128(unsigned addr) {
    f16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
template <int UNIT>
__device__ __forceinline__ void wait_vmcnt_units(int units) {   // at most `units` groups of UNIT DMA instructions stay in flight
    static_assert(7 * UNIT <= 63, "vmcnt is a 6-bit counter");
    if (units >= 7) wait_vmcnt<7 * UNIT>();
    else if (units == 6) wait_vmcnt<6 * UNIT>();
    else if (units == 5) wait_vmcnt<5 * UNIT>();
    else if (units == 4) wait_vmcnt<4 * UNIT>();
    else if (units == 3) wait_vmcnt<3 * UNIT>();
    else if (units == 2) wait_vmcnt<2 * UNIT>();
    else if (units == 1) wait_vmcnt<UNIT>();
    else wait_vmcnt<0>();
}

// ------------------------------------------------------------------------------------------------------------------
// 256x256 tile variant of the LDS-DMA kernel.  A 128x128 tile is bound by what the LDS-DMA path of a CU delivers
// (~60 GB/s: 32 KiB per K step in 0.53 us with the MFMAs removed); a 256x256 tile needs half the operand
// bytes per flop, which one fp32 accumulator per output (common.h: both operand halves share a scale) makes affordable:
// 8 waves as 2 x 4, wave tile 128 x 64 = 4 x 2 MFMA tiles = 128 accumulator registers.
//   LDS: 2 stages of 64 KiB (A 256 rows x 128 B, then W 256 rows x 128 B, same XOR swizzle as above).  One K step (32 deep)
//   is 48 MFMAs per wave (~1.5 us per CU), long enough that a two-stage ring covers the L2/HBM latency:
//     top of K step kt:  vmcnt(0) (stage kt, issued during step kt-1, has landed for this wave) + ONE barrier (landed for
//                        all waves, and all waves are done reading stage kt-1, whose buffer is refilled next)
//     body:              the 8 DMA pieces of stage kt+1 go out one per two MFMAs at the start of the step;
//                        fragments roll through registers: B fragments of a k block (2 tiles) are double buffered, A tiles
//                        (4 per k block) stream through two slots, each read issued 6 MFMAs before its first use and
//                        waited for with a counted lgkmcnt.
template <int ABL = 0>
__global__ __launch_bounds__(512, 1) void gemm_p8_256_kernel(const GemmArgs g) {
    constexpr int BM = 256, BN = 256, BK = 32;
    constexpr int STAGE_BYTES = (BM + BN) * 128;     // 64 KiB
    constexpr int NDMA = 8;                           // 1-KiB pieces per wave per stage: 4 of A, 4 of W
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_p8[];

    const int tid = threadIdx.x;
    unsigned long long* stamps = nullptr;
    if constexpr (ABL == 6) {
        stamps = reinterpret_cast<unsigned long long*>(g.partial) + (long)blockIdx.x * 8;
        if (tid == 0) {
            stamps[0] = wall_clock64();
            stamps[4] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
            stamps[5] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));
        }
    }
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
    int tm, tn;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        const int idx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        constexpr int GM = 8;      // row tiles per XCD tile group (8 x 4 tiles per XCD measured 0.27 ms per step better than 2 x 16)
        const int width = GM * tiles_n;
        const int group = idx / width, first_m = group * GM;
        const int gsz = min(tiles_m - first_m, GM);
        const int in_g = idx - group * width;
        tn = in_g / gsz;
        tm = first_m + (in_g - tn * gsz);
    }
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- DMA addressing: piece = 8 rows x 128 B; wave w owns A rows 32w..32w+31 and W rows 32w..32w+31 of the tile ----
    const int prow = lane >> 3, pchunk = lane & 7;
    const unsigned char* src[NDMA];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int ra = wave * 32 + q * 8 + prow;
        const int gm = min(m0 + ra, g.M - 1);          // clamp: rows >= M are never stored
        const int gn = min(n0 + ra, g.N - 1);
        const int sw = (pchunk ^ ((ra >> 1) & 7)) << 4;
        src[q] = reinterpret_cast<const unsigned char*>(g.A) + ((long)gm * g.lda) * 4 + sw;
        src[4 + q] = reinterpret_cast<const unsigned char*>(g.Wp) + ((long)gn * g.ldw) * 4 + sw;
    }
    auto issue_piece = [&](int q, int kt, int buf) {
        unsigned char* dst = smem_p8 + buf * STAGE_BYTES + (q < 4 ? 0 : BM * 128) + (wave * 4 + (q & 3)) * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[q] + (long)kt * (BK * 4)),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };

    const int wm = wave >> 2, wn = wave & 3;
    const int r = lane & 31, h = lane >> 5;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_p8;
    unsigned a_off[2][2], w_off[2][2];               // [kb][hi/lo]; further tiles are +4096 B per 32 rows (same swizzle key)
    {
        const int arow = wm * 128 + r, wrow = wn * 64 + r;
        const int akey = (arow >> 1) & 7, wkey = (wrow >> 1) & 7;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int lo = 0; lo < 2; ++lo) {
                const int c = (kb * 2 + h) * 2 + lo;
                a_off[kb][lo] = lds0 + arow * 128 + ((c ^ akey) << 4);
                w_off[kb][lo] = lds0 + BM * 128 + wrow * 128 + ((c ^ wkey) << 4);
            }
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = g.K / BK;
#pragma unroll
    for (int q = 0; q < NDMA; ++q) issue_piece(q, 0, 0);

    f16x8 bh[2][2], bl[2][2];      // [k block parity][n tile]
    f16x8 ah[2], al[2];            // two slots, m tile i lives in slot i & 1
    auto read_b = [&](unsigned sb, int kb) {
        bh[kb][0] = lds_read128<0>(w_off[kb][0] + sb);
        bl[kb][0] = lds_read128<0>(w_off[kb][1] + sb);
        bh[kb][1] = lds_read128<4096>(w_off[kb][0] + sb);
        bl[kb][1] = lds_read128<4096>(w_off[kb][1] + sb);
    };
    auto read_a = [&](unsigned sb, int kb, int i) {
        const unsigned hp = a_off[kb][0] + sb, lp = a_off[kb][1] + sb;
        if (i == 0) { ah[0] = lds_read128<0>(hp); al[0] = lds_read128<0>(lp); }
        else if (i == 1) { ah[1] = lds_read128<4096>(hp); al[1] = lds_read128<4096>(lp); }
        else if (i == 2) { ah[0] = lds_read128<8192>(hp); al[0] = lds_read128<8192>(lp); }
        else { ah[1] = lds_read128<12288>(hp); al[1] = lds_read128<12288>(lp); }
    };
    // one K step from ring buffer `buf`; ISSUE: fetch stage kt+1 into the other buffer meanwhile.  The eight (k block, m tile)
    // sub-steps are spelled out with compile-time indices (fragment slots and DMA pieces are register arrays).
    auto substep = [&](int kt, int buf, unsigned sb, auto issue_tag, auto kb_tag, auto i_tag) {
        constexpr bool ISSUE = decltype(issue_tag)::value;
        constexpr int kb = decltype(kb_tag)::value, i = decltype(i_tag)::value;
        // prefetch the next fragments, then wait for everything older than them
        if constexpr (i < 3) { read_a(sb, kb, i + 1); wait_lgkmcnt<2>(); }
        else if constexpr (kb == 0) { read_b(sb, 1); read_a(sb, 1, 0); wait_lgkmcnt<6>(); }
        else wait_lgkmcnt<0>();
        constexpr int sl = i & 1;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (t == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[kb][j], ah[sl], acc[i][j], 0, 0, 0);
                else if (t == 1) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[kb][j], ah[sl], acc[i][j], 0, 0, 0);
                else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[kb][j], al[sl], acc[i][j], 0, 0, 0);
                const int piece = i * 3 + t;     // one DMA piece per two MFMAs over the first 16 MFMAs of the step
                if (ISSUE && kb == 0 && j == 1 && piece < NDMA) {
                    __builtin_amdgcn_sched_barrier(0);
                    issue_piece(piece, kt + 1, buf ^ 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto kstep = [&](int kt, int buf, auto issue_tag) {
        const unsigned sb = buf * STAGE_BYTES;
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        read_b(sb, 0);
        read_a(sb, 0, 0);
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
        substep(kt, buf, sb, issue_tag, I0{}, I0{});
        substep(kt, buf, sb, issue_tag, I0{}, I1{});
        substep(kt, buf, sb, issue_tag, I0{}, I2{});
        substep(kt, buf, sb, issue_tag, I0{}, I3{});
        substep(kt, buf, sb, issue_tag, I1{}, I0{});
        substep(kt, buf, sb, issue_tag, I1{}, I1{});
        substep(kt, buf, sb, issue_tag, I1{}, I2{});
        substep(kt, buf, sb, issue_tag, I1{}, I3{});
    };
    int kt = 0;
    for (; kt + 1 < nk; ++kt) {
        if (ABL == 6 && kt == 1) { if (tid == 0) stamps[1] = wall_clock64(); }
        kstep(kt, kt & 1, std::true_type{});
    }
    kstep(kt, kt & 1, std::false_type{});
    if constexpr (ABL == 6) { if (tid == 0) stamps[2] = wall_clock64(); }

    const EpiCtx epi = make_epi(g, g.bias, g.C, g.R);
    if (epi.vec && ABL != 1) {      // coalesced: two passes (m tiles 0-1, then 2-3) through this wave's 17 KiB slice of LDS (ABL 1: tuning, direct stores)
        __builtin_amdgcn_s_barrier();      // every wave has consumed its last fragments
        constexpr int SLICE = 64 * 68;
        float* lds = reinterpret_cast<float*>(smem_p8) + wave * SLICE;
        f32x16 (&lo2)[2][2] = *reinterpret_cast<f32x16 (*)[2][2]>(&acc[0]);
        f32x16 (&hi2)[2][2] = *reinterpret_cast<f32x16 (*)[2][2]>(&acc[2]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] *= p8_out_scale_of(g.a_exp);
        epilogue_wave_lds<2, 2>(g, epi, lds, m0 + wm * 128, n0 + wn * 64, lane, lo2);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        epilogue_wave_lds<2, 2>(g, epi, lds, m0 + wm * 128 + 64, n0 + wn * 64, lane, hi2);
    } else {
        epilogue_tiles<true, false, 4, 2>(g, epi, m0 + wm * 128 + r, n0 + wn * 64, h, acc, p8_out_scale_of(g.a_exp));
    }
    if constexpr (ABL == 6) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (tid == 0) stamps[3] = wall_clock64();
    }
}


// ------------------------------------------------------------------------------------------------------------------
// Persistent big-tile kernel: (64 TM) x 256 tiles, TM = 4 (256 x 256, the tile of gemm_p8_256_kernel) or 5 (320 x 256), one
// workgroup of 8 waves (2 x 4, wave tile 32 TM x 64 = TM x 2 MFMA tiles = 128 / 160 accumulator registers) per CU walking the
// tile list with stride gridDim.x.  What it adds to gemm_p8_256_kernel:
//   * the LDS-DMA ring runs ACROSS tile boundaries: stage 0 of the next tile is fetched during the last K step of this one and its
//     stage 1 right after that step's barrier, so a tile has no prologue (3.3 us of 90) and the DMA latency of the first two stages
//     sits under the epilogue;
//   * the epilogue stores straight from the accumulators (no LDS transposition: the ring is busy) and the next tile starts behind
//     counted vmcnt waits that leave those stores in flight for its first two K steps;
//   * the bias row of a tile arrives by one more DMA instruction per wave with its stage 0 (256 B per wave, two slots) and is read with
//     ds_read in the epilogue: no vector-memory LOAD sits between the stores (vmcnt counts loads and stores in issue order, a load
//     in the epilogue would drain every store in front of it);
//   * 320 x 256 tiles: the M = 19200 GEMMs of batch 32 cut into 60 row tiles, so N = 1024 (out-projection, FFN-out) is ONE round of
//     240 workgroups on 256 CUs instead of 2.34 rounds of the 128x128 kernel's 512, and q|k|v 2.81 rounds instead of 3.52.
// Accumulation order per output element is that of every other split kernel (k ascending; hi*hi, lo*hi, hi*lo per 16-deep block), so
// results are bit-identical to theirs.
template <int U, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (U < N) {
        f(std::integral_constant<int, U>{});
        static_for<U + 1, N>(f);
    }
}
template <int TM, bool RES, int TAG = 0>      // RES: tiles with a residual (g.R != nullptr), a compile-time property so that the epilogue is straight-line code
__global__ __launch_bounds__(512, 1) void gemm_p8_big_kernel(const GemmArgs g) {
    constexpr int BM = 64 * TM, BN = 256, BK = 32;
    constexpr int STAGE_BYTES = (BM + BN) * 128;
    constexpr int NDMA = TM + 4;                      // 1-KiB pieces per wave per stage: TM of A, 4 of W
    constexpr int NSUB = 2 * TM;                      // sub-steps (k block, m tile) of a K step
    constexpr int BIAS_OFF = 2 * STAGE_BYTES;         // two slots of 8 x 256 B behind the ring
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_p8[];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
    const int ntiles = tiles_n * tiles_m;
    const int wm = wave >> 2, wn = wave & 3;
    const int nk = g.K / BK;
    // Everything derived from the lane index (fragment addresses, DMA source offsets) is RE-derived at the phase boundaries of a tile
    // (derive(): the lane index passes through an empty asm, so nothing computed from the old copy can be reused): the main loop's
    // address registers are then dead in the epilogue and the epilogue's in the main loop, instead of 30 registers living through both -
    // with 128 / 160 accumulator registers of 256 that is the difference between no spills and hundreds.
    int lane = tid & 63, r, h, prow, pchunk;
    // LDS byte addresses of this lane's hi / lo fragments of k block 0 in ring buffer 0 (the dynamic LDS starts at address 0: rows are
    // 128-byte aligned); k block 1 is the same address with bit 6 flipped (logical chunk + 4 under the XOR swizzle), ring buffer 1 is
    // + STAGE_BYTES, further tiles + 4096 B per 32 rows (same swizzle key, an immediate).  Four registers instead of sixteen.
    unsigned a_off[2], w_off[2];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_p8;
    auto derive = [&]() __attribute__((always_inline)) {
        asm volatile("" : "+v"(lane));
        r = lane & 31; h = lane >> 5; prow = lane >> 3; pchunk = lane & 7;
        const int arow = wm * (32 * TM) + r, wrow = wn * 64 + r;
        const int akey = (arow >> 1) & 7, wkey = (wrow >> 1) & 7;
#pragma unroll
        for (int lo = 0; lo < 2; ++lo) {
            const int c = h * 2 + lo;
            a_off[lo] = lds0 + arow * 128 + ((c ^ akey) << 4);
            w_off[lo] = lds0 + BM * 128 + wrow * 128 + ((c ^ wkey) << 4);
        }
    };
    derive();

    // tile index -> origin: XCD-contiguous (workgroups b and b + 8 share an XCD and gridDim.x % 8 == 0, so tile t runs on XCD t & 7),
    // grouped column-major inside groups of GM row tiles
    auto tile_origin = [&](int t, int& m0, int& n0) __attribute__((always_inline)) {
        const int xcd = t & 7, q = ntiles >> 3, rr = ntiles & 7;
        const int idx = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (t >> 3);
        // row tiles per group: 8 (an XCD then works on ~8 x 4 tiles at a time) - or ALL of them when there are few (the AdaLN table:
        // 23 row tiles x 222 column tiles; with groups of 8 every XCD walked all 222 column tiles and pulled the whole 233 MB weight
        // matrix, 2.0 GB read per launch; column-major over all rows an XCD reads an eighth of it)
        const int GM = tiles_m <= 32 ? tiles_m : 8;
        const int width = GM * tiles_n;
        const int group = idx / width, first_m = group * GM;
        const int gsz = min(tiles_m - first_m, GM);
        const int in_g = idx - group * width;
        const int tn = in_g / gsz;
        m0 = (first_m + (in_g - tn * gsz)) * BM;
        n0 = tn * BN;
    };

    // DMA source addressing: piece q of a wave covers tile rows (wave * TM + q) * 8 + prow (A) / (wave * 4 + q) * 8 + prow (W); lane =
    // (row in piece, physical 16-byte chunk).  The pieces are BUFFER loads to LDS (buffer_load_dwordx4 ... offen lds): the operand is
    // described once by a resource descriptor in scalar registers, the lane's byte offset is 32 bits (the launcher takes this kernel
    // only for operands below 4 GiB), the K offset rides in the instruction's scalar offset, and rows beyond the operand's last row
    // (edge tiles) are dropped by the hardware's range check and land as zeros - no clamp.  A piece costs two VALU instructions
    // (row offset of piece q, swizzle), one scalar add for M0 and the load; the flat form (64-bit address per lane, min(), carry
    // chain) took eleven instructions per piece, and this kernel is bound by what its two waves per SIMD issue BETWEEN their MFMAs
    // (hardware counters: the matrix pipe busy 75 % of the cycles with every wait and barrier removed).
    // Only the offset of the lane's row in piece 0 and its swizzle term live in registers - nine stored offsets were what pushed
    // the 320-row tile into spilling.  Swizzle: chunk ^ ((row >> 1) & 7) with row = 8 * piece + prow, i.e. (piece & 1) * 4 + (prow >> 1).
    unsigned a_row0 = 0, w_row0 = 0, swz = 0;        // byte offsets of the lane's row of piece 0; (pchunk ^ (prow >> 1)) << 4
    const unsigned a_step = (unsigned)(8 * g.lda * 4), w_step = (unsigned)(8 * g.ldw * 4);
    // descriptors: base, stride 0 (raw), bytes up to the end of the last row's K run, 32-bit data format
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0,
                                                                            (int)(((unsigned)(g.M - 1) * (unsigned)g.lda + (unsigned)g.K) * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned int*>(g.Wp), 0,
                                                                            (int)(((unsigned)(g.N - 1) * (unsigned)g.ldw + (unsigned)g.K) * 4u), 0x00020000);
    const unsigned char* const Bbase = g.bias ? reinterpret_cast<const unsigned char*>(g.bias) : reinterpret_cast<const unsigned char*>(g.Wp);
    int bias_col0 = 0;                               // (wave-uniform) first bias column of this wave for the tile `src` points at
    auto set_src = [&](int m0, int n0) __attribute__((always_inline)) {
        a_row0 = (unsigned)(m0 + wave * TM * 8 + prow) * (unsigned)(g.lda * 4);
        w_row0 = (unsigned)(n0 + wave * 32 + prow) * (unsigned)(g.ldw * 4);
        swz = (unsigned)((pchunk ^ (prow >> 1)) << 4);
        bias_col0 = n0 + wn * 64;
    };
    auto issue_piece = [&](int q, int kt, int buf) __attribute__((always_inline)) {
        unsigned char* dst = smem_p8 + buf * STAGE_BYTES + (q < TM ? (wave * TM + q) * 1024 : BM * 128 + (wave * 4 + q - TM) * 1024);
        const int p = q < TM ? wave * TM + q : wave * 4 + q - TM;             // piece index inside the operand's tile (wave-uniform)
        const unsigned off = (q < TM ? a_row0 + (unsigned)q * a_step : w_row0 + (unsigned)(q - TM) * w_step) + (swz ^ (unsigned)((p & 1) << 6));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(q < TM ? a_rsrc : w_rsrc, (__attribute__((address_space(3))) void*)dst, 16, (int)off, kt * (BK * 4), 0, 0);
    };
    auto issue_bias = [&](int slot) __attribute__((always_inline)) {
        // this wave's 64 bias values, one float per lane (without a bias: any valid address, the slot is then unused)
        const unsigned off = g.bias ? (unsigned)min(bias_col0 + lane, g.N - 1) * 4u : (unsigned)lane * 4u;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Bbase + off),
                                         (__attribute__((address_space(3))) void*)(smem_p8 + BIAS_OFF + slot * 2048 + wave * 256), 4, 0, 0);
    };

    f32x16 acc[TM][2];
    f16x8 bh[2][2], bl[2][2];      // [k block][n tile]
    f16x8 ah[2], al[2];            // two slots, sub-step u lives in slot u & 1 (requesting fragments two sub-steps ahead measured no gain)
    auto read_b = [&](unsigned sb, int kb) __attribute__((always_inline)) {
        const unsigned hp = (w_off[0] + sb) ^ (kb << 6), lp = (w_off[1] + sb) ^ (kb << 6);
        bh[kb][0] = lds_read128<0>(hp);
        bl[kb][0] = lds_read128<0>(lp);
        bh[kb][1] = lds_read128<4096>(hp);
        bl[kb][1] = lds_read128<4096>(lp);
    };
    // Top of a K step that reads ring buffer `buf`: its stage has landed for this wave (pre: vector-memory instructions that may stay in
    // flight - everything OLDER than them has landed), then for every wave, and every wave is done reading the other buffer (barrier);
    // then the fragments of the step's first sub-step are requested.
    auto top = [&](int buf, int pre) __attribute__((always_inline)) {
        const unsigned sb = buf * STAGE_BYTES;
        // the largest count of the ladder that does not exceed `pre` (waiting for more than asked is always safe)
        if (pre >= 8 * TM + 1 + NDMA) wait_vmcnt<8 * TM + 1 + NDMA>();    // later tiles, step 0: bias piece + stage 1 + the previous tile's stores
        else if (pre >= 8 * TM + NDMA) wait_vmcnt<8 * TM + NDMA>();       // later tiles, step 1: those stores + step 0's re-fetched pieces
        else if (pre >= 16 + NDMA) wait_vmcnt<16 + NDMA>();               // the same with a residual (the last few sub-tiles' stores)
        else if (pre >= 16) wait_vmcnt<16>();
        else if (pre >= 1 + NDMA) wait_vmcnt<1 + NDMA>();                 // first tile, step 0: bias piece + stage 1
        else if (pre >= NDMA) wait_vmcnt<NDMA>();                         // first tile, step 1: step 0's re-fetched pieces
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#ifndef BIG_ABL_NOLDS
        read_b(sb, 0);
        ah[0] = lds_read128<0>(a_off[0] + sb);
        al[0] = lds_read128<0>(a_off[1] + sb);
#endif
    };
    // One K step out of ring buffer `buf`; its top() has run.  The pieces of K tile kt_issue of the tile `src` points at go into the
    // other buffer meanwhile, one per two MFMAs - in EVERY step, so that the step is straight-line code (ONE copy serves the whole
    // kernel, no branch per piece): the two steps that have nothing new to fetch re-fetch what the other buffer holds or will not
    // need (the callers' comments).  next_top: the NEXT step's top() runs in front of this step's last sub-step - by then every
    // fragment of this stage is in registers, the next stage (fetched during the first third of this step) has had 40+ MFMAs to land,
    // and the next step's first fragments ride under the last six MFMAs instead of stalling both waves of a SIMD at the step boundary
    // (same-process A/B, interleaved rounds, profiles/r03_gemm_early_top_ab.log: 1.4 - 3.4 % of the whole launch).
    auto kstep = [&](int buf, int kt_issue, bool next_top, int next_pre) __attribute__((always_inline)) {
        const unsigned sb = buf * STAGE_BYTES;
        static_for<0, NSUB>([&](auto u_tag) __attribute__((always_inline)) {
            constexpr int u = decltype(u_tag)::value, kb = u / TM, i = u % TM, sl = u & 1;
            if constexpr (u + 1 < NSUB) {       // prefetch the next sub-step's fragments, then wait for everything older than them
                constexpr int nkb = (u + 1) / TM, ni = (u + 1) % TM, nsl = (u + 1) & 1;
#ifndef BIG_ABL_NOLDS      // (timing builds, results wrong: BIG_ABL_NOLDS no fragment reads, BIG_ABL_NODMA no DMA pieces, BIG_ABL_ONEPROD one MFMA product of three, BIG_ABL_NOSTORE stores masked)
                if constexpr (ni == 0) read_b(sb, nkb);
                ah[nsl] = lds_read128<ni * 4096>((a_off[0] + sb) ^ (nkb << 6));
                al[nsl] = lds_read128<ni * 4096>((a_off[1] + sb) ^ (nkb << 6));
#endif
                if constexpr (ni == 0) wait_lgkmcnt<6>(); else wait_lgkmcnt<2>();
            } else {
                wait_lgkmcnt<0>();
                if (next_top) top(buf ^ 1, next_pre);      // (slot 0 and the k-block-0 weight fragments are free: this sub-step uses slot 1 / k block 1)
            }
#ifdef BIG_SETPRIO      // (experiment: the wave whose fragments are in gets the issue slots for its six MFMAs)
            __builtin_amdgcn_s_setprio(BIG_SETPRIO);
#endif
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (t == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[kb][j], ah[sl], acc[i][j], 0, 0, 0);
#ifndef BIG_ABL_ONEPROD
                    else if (t == 1) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[kb][j], ah[sl], acc[i][j], 0, 0, 0);
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[kb][j], al[sl], acc[i][j], 0, 0, 0);
#endif
                    const int piece = u * 3 + t;     // one DMA piece per two MFMAs from the start of the step
#ifdef BIG_ABL_NODMA
                    if (false) {
#else
                    if (j == 1 && piece < NDMA) {
#endif
                        __builtin_amdgcn_sched_barrier(0);
                        issue_piece(piece, kt_issue, buf ^ 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#ifdef BIG_SETPRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    int t = blockIdx.x;
    if (t >= ntiles) return;
#ifdef BIG_ABL_NOLDS
    read_b(0, 0); read_b(0, 1);
    ah[0] = lds_read128<0>(a_off[0]); al[0] = lds_read128<0>(a_off[1]); ah[1] = lds_read128<4096>(a_off[0]); al[1] = lds_read128<4096>(a_off[1]);
#endif
    int m0, n0;
    tile_origin(t, m0, n0);
    set_src(m0, n0);
#pragma unroll
    for (int q = 0; q < NDMA; ++q) issue_piece(q, 0, 0);
    issue_bias(0);
    if (nk > 1) {
#pragma unroll
        for (int q = 0; q < NDMA; ++q) issue_piece(q, 1, 1);
    }
    // vector-memory instructions issued after the stage that K step 0 / K step 1 of the current tile reads (see kstep's `pre`)
    int pre0 = 1 + (nk > 1 ? NDMA : 0), pre1 = 0;       // (+ NDMA for step 1: the pieces step 0 re-fetches)
    int gs = 0, tile_no = 0;        // ring parity (K steps run so far), bias slot parity
    while (true) {
        if (tile_no > 0) { derive(); set_src(m0, n0); }      // (see derive(): the epilogue's registers end here)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        const int t_next = t + gridDim.x;
#ifdef BIG_NO_NEXT
        const bool has_next = false;
#else
        const bool has_next = t_next < ntiles;
#endif
        int nm0 = 0, nn0 = 0;
        if (has_next) tile_origin(t_next, nm0, nn0);
        // Stages 0 and 1 of a tile are in the ring when it starts.  Step kt reads buffer gs & 1 and fetches K tile kt + 1 into the
        // other one; for step 0 that is the K tile the other buffer already holds (K tile 2 would need the buffer step 0 itself reads):
        // the same bytes land on themselves, which is harmless under the reads of the next step's first fragments, and 1 / nk more
        // traffic buys a branch-free step.  The last step runs the ring on into the next tile: its stage 0 goes into the buffer of the
        // step before; the very last step of the workgroup re-fetches K tile 0 of its own tile into that (free) buffer.
        top(gs & 1, nk > 1 ? pre0 : 0);      // (nk == 1: stage 0 only, waited for in full)
#ifdef BIG_PHASE      // timing build (results wrong): the first tile of a workgroup runs a phase-dependent share of its K steps, so the CUs' epilogues stop coinciding
#ifdef BIG_PHASE_XCD     // the CUs of an XCD stay in phase (they share operand fetches through their L2), the eight XCDs are spread over a tile period
        const int nk_cur = tile_no == 0 ? max(2, (nk * ((int)(blockIdx.x & 7) + 1)) / 8) : nk;
#else
        const int nk_cur = tile_no == 0 ? max(2, (nk * ((int)((blockIdx.x >> 3) % BIG_PHASE) + 1)) / BIG_PHASE) : nk;
#endif
#else
        const int nk_cur = nk;
#endif
#pragma nounroll
        for (int kt = 0; kt < nk_cur; ++kt) {
            const bool last = kt + 1 == nk_cur;
            if (last && has_next) set_src(nm0, nn0);
            kstep(gs & 1, last ? 0 : kt + 1, !last, kt == 0 ? pre1 + NDMA : 0);
            ++gs;
        }
        if (has_next) issue_bias((tile_no + 1) & 1);
        // stage 1 of the next tile into the buffer the last step has just read (every wave is done with it after this barrier)
        int behind = 0;                 // vector-memory instructions issued behind the next tile's stage 0
        if (has_next) {
            behind = 1;                 // the bias piece (above)
            if (nk > 1) {
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < NDMA; ++q) issue_piece(q, 1, gs & 1 ? 0 : 1);
                __builtin_amdgcn_sched_barrier(0);
                behind += NDMA;
            }
        }
        // ---- epilogue of tile (m0, n0): bias from this wave's LDS slot (written by its own DMA, which came with the tile's stage 0 and is
        // older than everything waited for since), activation, exactly 4 store instructions per sub-tile (rows beyond M masked).  The
        // launcher guarantees the 16-byte path (make_epi's test), no gate and N % 256 == 0. ----
        derive();       // the main loop's address registers end here
        const unsigned char* bslot = smem_p8 + BIAS_OFF + (tile_no & 1) * 2048 + wave * 256;
        constexpr bool has_res = RES;      // (never together with an activation or a P8 result: p8_big_ok)
        auto finish = [&](auto i_tag, auto j_tag, const f32x4 (&res)[4]) __attribute__((always_inline)) {      // sub-tile (i, j)
            constexpr int i = decltype(i_tag)::value, j = decltype(j_tag)::value;
            const int row = m0 + wm * (32 * TM) + i * 32 + r;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 b = {0.f, 0.f, 0.f, 0.f};
                if (g.bias) b = *reinterpret_cast<const f32x4*>(bslot + (j * 32 + 8 * q + 4 * h) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] = acc[i][j][4 * q + e] * p8_out_scale_of(g.a_exp) + b[e];
            }
            if constexpr (has_res) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] += res[q][e];
            }
#ifdef BIG_ABL_NOSTORE
            epilogue_tile32_store<false>(g, g.C, row, n0 + wn * 64 + j * 32, h, acc[i][j], nullptr, row < (g.force_cfg == 12345 ? g.M : -1));
#else
            epilogue_tile32_store<false>(g, g.C, row, n0 + wn * 64 + j * 32, h, acc[i][j], nullptr, row < g.M);
#endif
        };
        using J0 = std::integral_constant<int, 0>; using J1 = std::integral_constant<int, 1>;
        // Residual tiles (the encoder's out-projection and FFN-out, x += ... in place; fp32 result): the residual runs of sub-tile s + 1
        // are requested before sub-tile s is stored (vmcnt counts loads and stores in issue order: requested after the store, they would
        // wait for it).  The sub-tiles are spelled out (no loop): in straight-line code hipcc counts its waits exactly - the one in front
        // of sub-tile s leaves the 4 stores of s - 1 and the 4 loads of s + 1 in flight (vmcnt(8) in the disassembly) - while at a loop
        // back-edge it falls back to vmcnt(0) and drains the stores every time (and raw asm loads are no way out: the compiler takes
        // their results for complete and may recycle the destination registers while the data is still on its way).
        f32x4 r0[4], r1[4];
        auto load_res = [&](f32x4 (&dst)[4], int i, int j) __attribute__((always_inline)) {
            const int row = min(m0 + wm * (32 * TM) + i * 32 + r, g.M - 1);          // (rows beyond M: any valid address, the stores are masked)
            const float* rp = g.R + (long)map_row(g.cmap, row) * g.ldr + n0 + wn * 64 + j * 32 + 4 * h;
#pragma unroll
            for (int q = 0; q < 4; ++q) dst[q] = *reinterpret_cast<const f32x4*>(rp + 8 * q);
        };
        if constexpr (has_res) load_res(r0, 0, 0);
        static_for<0, TM>([&](auto i_tag) __attribute__((always_inline)) {
            constexpr int i = decltype(i_tag)::value;
            if constexpr (has_res) load_res(r1, i, 1);
            finish(i_tag, J0{}, r0);
            if constexpr (has_res && i + 1 < TM) load_res(r0, i + 1, 0);
            finish(i_tag, J1{}, r1);
        });
        // vector-memory instructions of this epilogue that may still be in flight when the next tile starts: all its stores.  With a
        // residual every load of the epilogue has been waited for, and with it everything older - the next tile's stage 0, stage 1
        // and bias piece included; what is left are the stores behind the last load (at most three sub-tiles'), which 16 covers.
        const int stores = has_res ? 16 : 4 * NSUB;
        if (has_res) behind = 0;
        if (!has_next) break;      // (the re-fetched pieces of the last step are still in flight: waited for below)
        // the next tile's K step 0 needs its stage 0: everything issued behind it may stay in flight if it can be counted
        pre0 = behind + stores; pre1 = stores;
        t = t_next; m0 = nm0; n0 = nn0; ++tile_no;
    }
    wait_vmcnt<0>();      // no LDS-DMA of this workgroup may outlive it (the LDS goes to the CU's next workgroup)
}

// ------------------------------------------------------------------------------------------------------------------
// Two-workgroups-per-CU variant: 128x128 tile, 4 waves (2 x 2, wave tile 64 x 64 = 64 accumulator registers), 2 stages of
// 32 KiB = 64 KiB of LDS per workgroup, so two workgroups are co-resident on a CU and run out of phase: while one sits in its
// epilogue (a tile's 64 KiB of stores drain in ~3.6 us whatever their shape: the write path, not instruction issue, bounds
// them), in its prologue, at its barrier or waiting for a DMA stage, the other one owns the matrix cores.  Same two-stage
// K-step structure as the 256x256 kernel: vmcnt(0) + one barrier at the top of a step, the 8 DMA pieces of the next stage
// issued in the MFMA gaps of this one, fragments rolling through registers behind counted lgkmcnt waits.
// Tile order of the persistent 128x128 kernel: tile index t -> origin.
// XCD-contiguous (workgroups b and b + 8 share an XCD, so consecutive tile indices of one XCD walk one tile group) and grouped
// column-major inside groups of GM row tiles, so that an XCD's L2 sees few distinct weight / activation rows at a time.
__device__ __forceinline__ void p8_tile_origin128(int t, int ntiles, int tiles_m, int tiles_n, int& m0, int& n0, int GM = 4) {
    const int xcd = t & 7, q = ntiles >> 3, rr = ntiles & 7;
    const int idx = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (t >> 3);
    const int width = GM * tiles_n;
    const int group = idx / width, first_m = group * GM;
    const int gsz = min(tiles_m - first_m, GM);
    const int in_g = idx - group * width;
    const int tn = in_g / gsz;
    m0 = (first_m + (in_g - tn * gsz)) * 128;
    n0 = tn * 128;
}

// ------------------------------------------------------------------------------------------------------------------
// Persistent form of the two-workgroup kernel with a DEFERRED epilogue: 512 workgroups (two per CU) walk the tile list with
// stride 512; a finished tile's 64 result registers (scaled, bias added) are kept and its four 32x32 sub-tiles are finished
// (activation, P8 split) and stored one per K step during the first four K steps of the NEXT tile, so the stores trickle out
// under matrix work instead of blocking the CU's memory pipeline in a burst (the store tail costs 13-20 % of a K = 1024 GEMM,
// DESIGN.md).  Deferred only where it is free of loads and of partial lanes: full tiles, 16-byte aligned rows, no residual, no
// gate (everything else takes the immediate epilogue).  vmcnt counts loads and stores together in issue order, so the wait at
// the top of a K step that follows a deferred sub-tile leaves exactly its 4 store instructions in flight.
template <int TAG = 0>
__global__ __launch_bounds__(256, 2) void gemm_p8_2wgp_kernel(const GemmArgs g) {
    constexpr int BM = 128, BN = 128, BK = 32;
    constexpr int STAGE_BYTES = (BM + BN) * 128;     // 32 KiB
    constexpr int NDMA = 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_p8[];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
    const int ntiles = tiles_n * tiles_m;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int prow = lane >> 3, pchunk = lane & 7;
    const int nk = g.K / BK;

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_p8;
    unsigned a_off[2][2], w_off[2][2];
    {
        const int arow = wm * 64 + r, wrow = wn * 64 + r;
        const int akey = (arow >> 1) & 7, wkey = (wrow >> 1) & 7;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int lo = 0; lo < 2; ++lo) {
                const int c = (kb * 2 + h) * 2 + lo;
                a_off[kb][lo] = lds0 + arow * 128 + ((c ^ akey) << 4);
                w_off[kb][lo] = lds0 + BM * 128 + wrow * 128 + ((c ^ wkey) << 4);
            }
    }
    EpiCtx epi = make_epi(g, g.bias, g.C, g.R);
    const bool can_defer = epi.vec && !g.R && !g.gate && nk >= 5;      // four K steps before the last one carry the stores
    // Residual tiles (out-projection, FFN-out) are deferred too when the launcher added 16 KiB of LDS (g.res_lds): the kept sub-tile's
    // residual values come in by LDS-DMA at the top of its K step (4 x 1 KiB per wave, lane-private slots: each lane later reads back
    // exactly the 16 bytes it asked for), so they cost no registers while the step's MFMAs run; immediate, such a tile cost the
    // out-projection 15 us and FFN-out 20 us per launch
    const bool defer_res = g.res_lds && epi.vec && g.R && !g.gate && !g.c_p8 && nk >= 5;
    unsigned char* const res_lds = smem_p8 + 2 * STAGE_BYTES + wave * 4096;
    bool pending_res = false;
    float* const C0 = g.C;
    const float* const bias0 = g.bias;

    f32x16 acc[2][2], keep[2][2];
    bool pending = false;
    int pm0 = 0, pn0 = 0;
    float* pC = g.C;                 // result base of the kept (deferred) tile: its column group may differ from the current tile's
    const unsigned char* src[NDMA];
    f16x8 bh[2][2], bl[2][2], ah[2], al[2];

    auto issue_piece = [&](int q, int kt, int buf) {
        unsigned char* dst = smem_p8 + buf * STAGE_BYTES + (q < 4 ? 0 : BM * 128) + (wave * 4 + (q & 3)) * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[q] + (long)kt * (BK * 4)),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    auto read_b = [&](unsigned sb, int kb) {
        bh[kb][0] = lds_read128<0>(w_off[kb][0] + sb);
        bl[kb][0] = lds_read128<0>(w_off[kb][1] + sb);
        bh[kb][1] = lds_read128<4096>(w_off[kb][0] + sb);
        bl[kb][1] = lds_read128<4096>(w_off[kb][1] + sb);
    };
    auto read_a = [&](unsigned sb, int kb, int i) {
        const unsigned hp = a_off[kb][0] + sb, lp = a_off[kb][1] + sb;
        if (i == 0) { ah[0] = lds_read128<0>(hp); al[0] = lds_read128<0>(lp); }
        else { ah[1] = lds_read128<4096>(hp); al[1] = lds_read128<4096>(lp); }
    };
    auto rotate_keep = [&]() {
#pragma unroll
        for (int e = 0; e < 16; ++e) { keep[0][0][e] = keep[0][1][e]; keep[0][1][e] = keep[1][0][e]; keep[1][0][e] = keep[1][1][e]; }
    };
    auto substep = [&](int kt_issue, int buf, unsigned sb, auto issue_tag, auto kb_tag, auto i_tag) {
        constexpr bool ISSUE = decltype(issue_tag)::value;
        constexpr int kb = decltype(kb_tag)::value, i = decltype(i_tag)::value;
        if constexpr (i == 0) { read_a(sb, kb, 1); wait_lgkmcnt<2>(); }
        else if constexpr (kb == 0) { read_b(sb, 1); read_a(sb, 1, 0); wait_lgkmcnt<6>(); }
        else wait_lgkmcnt<0>();
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (t == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[kb][j], ah[i], acc[i][j], 0, 0, 0);
                else if (t == 1) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[kb][j], ah[i], acc[i][j], 0, 0, 0);
                else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[kb][j], al[i], acc[i][j], 0, 0, 0);
                const int piece = (kb * 2 + i) * 3 + t;
                if (ISSUE && j == 1 && piece < NDMA) {
                    __builtin_amdgcn_sched_barrier(0);
                    issue_piece(piece, kt_issue, buf ^ 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        __builtin_amdgcn_sched_barrier(0);
    };
    // behind: vector-memory instructions issued after the DMA pieces of the stage this step reads (0; the 4 stores of the deferred
    // sub-tile of the previous step; the 16 stores of an immediate full-tile epilogue); dq: sub-tile of the kept tile to finish at the
    // end of this step (-1: none); kt_issue: K tile (of the tile `src` points at) fetched into the other buffer meanwhile
    auto kstep = [&](int buf, auto issue_tag, int behind, int dq, int kt_issue) {
        const unsigned sb = buf * STAGE_BYTES;
        if (behind == 4) wait_vmcnt<4>(); else if (behind == 16) wait_vmcnt<16>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (dq >= 0 && pending_res) {      // (wave-uniform) residual of the sub-tile this step finishes: 4 DMA pieces, ahead of the stage's 8
            const long crow = map_row(g.cmap, pm0 + wm * 64 + (dq >> 1) * 32 + r);
            const float* rp = g.R + crow * g.ldr + pn0 + wn * 64 + (dq & 1) * 32 + 4 * h;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rp + 8 * q),
                                                 (__attribute__((address_space(3))) void*)(res_lds + q * 1024), 16, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        read_b(sb, 0);
        read_a(sb, 0, 0);
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        substep(kt_issue, buf, sb, issue_tag, I0{}, I0{});
        substep(kt_issue, buf, sb, issue_tag, I0{}, I1{});
        substep(kt_issue, buf, sb, issue_tag, I1{}, I0{});
        substep(kt_issue, buf, sb, issue_tag, I1{}, I1{});
        if (dq >= 0) {
            // ONE copy of the store code: the sub-tile to finish is always keep[0][0], the other three move up behind it (48 register
            // moves per sub-tile; with a copy per sub-tile and per K-step variant the kernel was 197 KB of code, three times the
            // instruction cache two CUs share, and a byte-identical second instantiation ran 27 % slower than the first)
            if (pending_res) {
                wait_vmcnt<8>();       // everything older than this step's 8 stage pieces: the 4 residual pieces have landed
                __builtin_amdgcn_sched_barrier(0);
                f32x4 rv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) rv[q] = *reinterpret_cast<const f32x4*>(res_lds + q * 1024 + lane * 16);
                epilogue_tile32_store<false>(g, pC, pm0 + wm * 64 + (dq >> 1) * 32 + r, pn0 + wn * 64 + (dq & 1) * 32, h, keep[0][0], rv);
            } else {
                epilogue_tile32_store<false>(g, pC, pm0 + wm * 64 + (dq >> 1) * 32 + r, pn0 + wn * 64 + (dq & 1) * 32, h, keep[0][0]);
            }
            rotate_keep();
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto tile_origin = [&](int t, int& m0, int& n0) { p8_tile_origin128(t, ntiles, tiles_m, tiles_n, m0, n0); };
    const int t_end = ntiles;
    auto set_src = [&](int m0, int n0) {
        const int grp = g.ngrp ? n0 / g.ngrp : 0;          // column group (kernels.h): global column indices stay, the bases move
        const long wgrp_off = grp ? (long)grp * (g.grpW - (long)g.ngrp * g.ldw) : 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ra = wave * 32 + q * 8 + prow;
            const int gm = min(m0 + ra, g.M - 1);
            const int gn = min(n0 + ra, g.N - 1);
            const int sw = (pchunk ^ ((ra >> 1) & 7)) << 4;
            src[q] = reinterpret_cast<const unsigned char*>(g.A) + ((long)gm * g.lda) * 4 + sw;
            src[4 + q] = reinterpret_cast<const unsigned char*>(g.Wp) + ((long)gn * g.ldw + wgrp_off) * 4 + sw;
        }
    };
    auto set_epi = [&](int n0) {     // result / bias bases of the tile whose accumulators are about to be finished
        if (g.ngrp) {
            const int grp = n0 / g.ngrp;
            epi.C = C0 + (long)grp * (g.grpC - g.ngrp);
            epi.bias = bias0 ? bias0 + (long)grp * (g.grpB - g.ngrp) : nullptr;
        }
    };

    int t = blockIdx.x;
    if (t >= t_end) return;
    int m0, n0;
    tile_origin(t, m0, n0);
    set_src(m0, n0);
#pragma unroll
    for (int q = 0; q < NDMA; ++q) issue_piece(q, 0, 0);
    int gs = 0, behind = 0;          // running K-step count (ring parity), see kstep
    while (true) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        const int t_next = t + gridDim.x;
        const bool has_next = t_next < t_end;
        int nm0 = 0, nn0 = 0;
        if (has_next) tile_origin(t_next, nm0, nn0);
        for (int kt = 0; kt + 1 < nk; ++kt) {
            const int dq = (pending && kt < 4) ? kt : -1;
            kstep(gs & 1, std::true_type{}, behind, dq, kt + 1);
            behind = dq >= 0 ? 4 : 0;
            ++gs;
        }
        pending = false;
        // last K step of the tile: the ring runs on into the next tile (its first stage is fetched now; the source pointers of
        // this tile are no longer needed)
        if (has_next) { set_src(nm0, nn0); kstep(gs & 1, std::true_type{}, behind, -1, 0); }
        else kstep(gs & 1, std::false_type{}, behind, -1, 0);
        ++gs;
        behind = 0;
        const bool full = m0 + BM <= g.M && n0 + BN <= g.N;
        set_epi(n0);
        {   // (the launcher takes this kernel only when the 16-byte epilogue path applies: epi.vec)
            // results move to the kept registers (scaled, bias added); the accumulators are dead from here on
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = n0 + wn * 64 + j * 32 + 8 * q + 4 * h;
                    f32x4 b = {0.f, 0.f, 0.f, 0.f};
                    if (epi.bias && c < g.N) b = *reinterpret_cast<const f32x4*>(epi.bias + c);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int e = 0; e < 4; ++e) keep[i][j][4 * q + e] = acc[i][j][4 * q + e] * p8_out_scale_of(g.a_exp) + b[e];
                }
            if ((can_defer || defer_res) && full) {
                pending = true; pending_res = defer_res; pm0 = m0; pn0 = n0; pC = epi.C;
            } else {                 // edge tile, gate, P8 result with a residual: finish it now (bias is already in)
                EpiCtx e2 = epi;
                e2.bias = nullptr;
#pragma nounroll
                for (int dq = 0; dq < 4; ++dq) {      // one copy of the code here too (see kstep; unrolled it measured 0.6 ms per step slower)
                    epilogue_tile32<true, false>(g, e2, m0 + wm * 64 + (dq >> 1) * 32 + r, n0 + wn * 64 + (dq & 1) * 32, h, keep[0][0]);
                    rotate_keep();
                }
                // a full fp32 tile issued exactly 16 store instructions behind the next tile's first stage (P8 results and
                // partial tiles: wait for everything)
                behind = (full && !g.c_p8) ? 16 : 0;
            }
        }
        if (!has_next) break;
        t = t_next; m0 = nm0; n0 = nn0;
    }
    if (pending) {
        EpiCtx e2 = epi;
        e2.bias = nullptr; e2.C = pC;
#pragma nounroll
        for (int dq = 0; dq < 4; ++dq) {
            const int row = pm0 + wm * 64 + (dq >> 1) * 32 + r, col = pn0 + wn * 64 + (dq & 1) * 32;
            if (pending_res) epilogue_tile32<true, false>(g, e2, row, col, h, keep[0][0]);      // last tile of this workgroup: plain residual loads
            else epilogue_tile32_store<false>(g, pC, row, col, h, keep[0][0]);
            rotate_keep();
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Mid-grid kernel for the 50- / 100-token AR scale steps and the VAE stacks (M = 800 ... 3200 rows, K = 768 / 3072): 128x128 tiles,
// ONE workgroup of 8 waves (2 x 4, wave tile 64 x 32) per CU and a ring of 4 stages of 32 KiB, three K tiles (96 KiB) in flight.
// At these sizes a launch is one or two rounds and a round is bound by DMA latency per K step: the 64x64 small-grid kernel
// (6 MFMAs per wave per step against a ~1.2 us fetch, three 16 KiB stages in flight per workgroup) takes 24 K steps x ~0.5 us per
// round of 512 tiles, and 900 tiles (q|k|v at M = 1600) are two rounds; one 128x128 tile per CU has four times the matrix work per
// fetched byte and the same bytes in flight.  Split-K capable (partial slabs as the small-grid kernel) for the N = 768 GEMMs.
template <int STAGES, int TAG>
__global__ __launch_bounds__(512, 1) void gemm_p8_mid_kernel(const GemmArgs g) {
    constexpr int BM = 128, BN = 128, BK = 32;
    constexpr int WN_WAVES = (BM == 128) ? 4 : 2;          // waves along N
    constexpr int TN = BN / WN_WAVES / 32;                 // n-tiles per wave (1 or 2); m-tiles per wave = 2
    constexpr int APIECES = BM / 64, WPIECES = 2;          // 1-KiB DMA pieces per wave per stage
    constexpr int NDMA = APIECES + WPIECES;
    constexpr int STAGE_BYTES = (BM + BN) * 128;
    static_assert(STAGES * STAGE_BYTES <= 160 * 1024, "LDS");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_p8[];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
    int tm, tn;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        const int idx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        constexpr int GM = (BM == 128) ? 4 : 2;
        const int width = GM * tiles_n;
        const int group = idx / width, first_m = group * GM;
        const int gsz = min(tiles_m - first_m, GM);
        const int in_g = idx - group * width;
        tn = in_g / gsz;
        tm = first_m + (in_g - tn * gsz);
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int nk_all_ = g.K / BK, kt0 = (int)((long)nk_all_ * blockIdx.y / g.splitk);     // split-K: first K tile of this workgroup

    // ---- DMA addressing: 1-KiB pieces (8 rows x 128 B); lane = (row in piece, physical chunk) ----
    const int prow = lane >> 3, pchunk = lane & 7;
    const unsigned char* asrc[APIECES];
    const unsigned char* wsrc[WPIECES];
#pragma unroll
    for (int q = 0; q < APIECES; ++q) {
        const int ra = (wave * APIECES + q) * 8 + prow;                 // tile row 0..BM-1
        const int gm = min(m0 + ra, g.M - 1);                           // clamp: rows >= M are never stored
        asrc[q] = reinterpret_cast<const unsigned char*>(g.A) + ((long)gm * g.lda) * 4 + ((pchunk ^ ((ra >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int q = 0; q < WPIECES; ++q) {
        const int rw = (wave * WPIECES + q) * 8 + prow;
        const int gn = min(n0 + rw, g.N - 1);
        wsrc[q] = reinterpret_cast<const unsigned char*>(g.Wp) + ((long)gn * g.ldw) * 4 + ((pchunk ^ ((rw >> 1) & 7)) << 4);
    }
    // piece q of this wave for K tile kt into ring buffer buf: q < APIECES -> A rows, else W rows
    auto issue_piece = [&](int q, int kt, int buf) {
        unsigned char* base = smem_p8 + buf * STAGE_BYTES;
        const long koff = (long)(kt0 + kt) * (BK * 4);
        if (q < APIECES)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[q] + koff),
                                             (__attribute__((address_space(3))) void*)(base + (wave * APIECES + q) * 1024), 16, 0, 0);
        else
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[q - APIECES] + koff),
                                             (__attribute__((address_space(3))) void*)(base + BM * 128 + (wave * WPIECES + q - APIECES) * 1024), 16, 0, 0);
    };
    auto issue = [&](int kt, int buf) {
#pragma unroll
        for (int q = 0; q < NDMA; ++q) issue_piece(q, kt, buf);
    };

    const int wm = wave / WN_WAVES, wn = wave % WN_WAVES;
    const int r = lane & 31, h = lane >> 5;
    // LDS byte addresses of this lane's fragments inside a stage, [kb][hi/lo]; the second m tile (and n tile) is 32 rows = 4096 B
    // further on with the same swizzle key, which goes into the instruction's immediate offset
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_p8;
    unsigned a_off[2][2], w_off[2][2];
    {
        const int arow = wm * 64 + r, wrow = wn * (32 * TN) + r;
        const int akey = (arow >> 1) & 7, wkey = (wrow >> 1) & 7;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int lo = 0; lo < 2; ++lo) {
                const int c = (kb * 2 + h) * 2 + lo;     // logical 16-byte chunk: hi fragment, lo = the next one
                a_off[kb][lo] = lds0 + arow * 128 + ((c ^ akey) << 4);
                w_off[kb][lo] = lds0 + BM * 128 + wrow * 128 + ((c ^ wkey) << 4);
            }
    }

    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = (int)((long)nk_all_ * (blockIdx.y + 1) / g.splitk) - kt0;      // K tiles [kt0, kt0 + nk); split-K workgroups write raw partial slabs
#pragma unroll
    for (int st = 0; st < STAGES - 1; ++st)
        if (st < nk) issue(st, st);

    // Fragment registers are one half-step ahead of the MFMAs, so LDS-read latency and the stage hand-over (vmcnt wait +
    // barrier) sit under matrix work instead of in front of it.  Steady-state iteration kt (branch-free, so that hipcc counts
    // its lgkmcnt waits exactly instead of draining at a join):
    //   read kb=1 frags of stage kt | 6 MFMA kb=0 | vmcnt: stage kt+1 landed | barrier |
    //   read kb=0 frags of stage kt+1 | 6 MFMA kb=1 with the DMA pieces of stage kt+S-1 issued one per MFMA gap
    // The DMA reuses the buffer of stage kt-1, whose fragment reads were all consumed by MFMAs issued before this barrier, so no
    // lgkmcnt drain is needed in front of the barrier and the kb=1 reads of stage kt stay in flight across it.
    constexpr int NRD = 4 + 2 * TN;    // ds_read_b128 per half step
    auto read_frags = [&](int buf, int kb, f16x8 (&ah)[2], f16x8 (&al)[2], f16x8 (&bh)[TN], f16x8 (&bl)[TN]) {
        const unsigned sb = buf * STAGE_BYTES;
        const unsigned ahp = a_off[kb][0] + sb, alp = a_off[kb][1] + sb, whp = w_off[kb][0] + sb, wlp = w_off[kb][1] + sb;
#ifndef MID_ABL_NOB      // (timing build: the weight operand neither fetched nor read from LDS after the first K step)
        bh[0] = lds_read128<0>(whp);
        bl[0] = lds_read128<0>(wlp);
#endif
        ah[0] = lds_read128<0>(ahp);
        al[0] = lds_read128<0>(alp);
        ah[1] = lds_read128<4096>(ahp);
        al[1] = lds_read128<4096>(alp);
#ifndef MID_ABL_NOB
        if constexpr (TN == 2) {
            bh[TN - 1] = lds_read128<4096>(whp);
            bl[TN - 1] = lds_read128<4096>(wlp);
        }
#endif
    };
    constexpr int NMF = 2 * TN * 3;    // MFMAs per half step
    auto mfma_slot = [&](int sidx, const f16x8 (&ah)[2], const f16x8 (&al)[2], const f16x8 (&bh)[TN], const f16x8 (&bl)[TN]) {
        // term-major order: consecutive MFMAs go to different accumulators.  Weight fragment = A operand: C^T, see epilogue_tile32
        const int t = sidx / (2 * TN), i = (sidx / TN) % 2, j = sidx % TN;
#ifdef MID_ABL_NOMFMA      // (timing builds, results wrong: MID_ABL_NOMFMA one MFMA of twelve per half step, MID_ABL_NODMA no DMA after the prologue, MID_ABL_NOBAR no barriers)
        if (sidx == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], ah[i], acc[i][j], 0, 0, 0);
#else
        if (t == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], ah[i], acc[i][j], 0, 0, 0);
        else if (t == 1) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[j], ah[i], acc[i][j], 0, 0, 0);
        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], al[i], acc[i][j], 0, 0, 0);
#endif
    };
    auto mfmas = [&](const f16x8 (&ah)[2], const f16x8 (&al)[2], const f16x8 (&bh)[TN], const f16x8 (&bl)[TN]) {
#pragma unroll
        for (int sidx = 0; sidx < NMF; ++sidx) mfma_slot(sidx, ah, al, bh, bl);
    };

    f16x8 ah0[2], al0[2], bh0[TN], bl0[TN], ah1[2], al1[2], bh1[TN], bl1[TN];
    wait_vmcnt_units<NDMA>(min(STAGES - 2, nk - 1));   // stage 0 landed (the prologue left up to STAGES-2 younger stages in flight)
    __builtin_amdgcn_s_barrier();
    read_frags(0, 0, ah0, al0, bh0, bl0);
#ifdef MID_ABL_NOB
    bh0[0] = lds_read128<0>(w_off[0][0]); bl0[0] = lds_read128<0>(w_off[0][1]); bh1[0] = lds_read128<0>(w_off[1][0]); bl1[0] = lds_read128<0>(w_off[1][1]);
#endif
    int kt = 0, buf = 0;                               // buf = kt % STAGES
    for (; kt + STAGES - 1 < nk; ++kt) {               // steady state: stage kt+S-1 still to be fetched
        const int nbuf = (buf + 1 == STAGES) ? 0 : buf + 1;
        const int fbuf = (buf == 0) ? STAGES - 1 : buf - 1;     // (kt + S - 1) % S
        read_frags(buf, 1, ah1, al1, bh1, bl1);
        wait_lgkmcnt<NRD>();                           // the kb=0 fragments (issued half a step ago) are in; the kb=1 reads fly on
        mfmas(ah0, al0, bh0, bl0);
        __builtin_amdgcn_sched_barrier(0);
#ifdef MID_ABL_NOB
        wait_vmcnt<(STAGES - 3) * APIECES>();
#else
        wait_vmcnt<(STAGES - 3) * NDMA>();             // stage kt+1 landed for this wave (stages kt+2 .. kt+S-2 may still fly)
#endif
#ifndef MID_ABL_NOBAR
        __builtin_amdgcn_s_barrier();                  // ... and for every wave; every wave is done with stage kt-1
#endif
        __builtin_amdgcn_sched_barrier(0);
        read_frags(nbuf, 0, ah0, al0, bh0, bl0);
        wait_lgkmcnt<NRD>();                           // kb=1 fragments of stage kt
#pragma unroll
        for (int sidx = 0; sidx < NMF; ++sidx) {
            mfma_slot(sidx, ah1, al1, bh1, bl1);
#ifdef MID_ABL_NODMA
            if (false) {
#elif defined(MID_ABL_NOB)
            if (sidx < APIECES) {
#else
            if (sidx < NDMA) {
#endif
                __builtin_amdgcn_sched_barrier(0);
                issue_piece(sidx, kt + STAGES - 1, fbuf);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        buf = nbuf;
    }
    for (; kt < nk; ++kt) {                            // drain: the last S-1 stages are in flight or landed
        const int nbuf = (buf + 1 == STAGES) ? 0 : buf + 1;
        read_frags(buf, 1, ah1, al1, bh1, bl1);
        wait_lgkmcnt<NRD>();
        mfmas(ah0, al0, bh0, bl0);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) {
            wait_vmcnt_units<NDMA>(min(STAGES - 3, nk - 2 - kt));
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            read_frags(nbuf, 0, ah0, al0, bh0, bl0);
            wait_lgkmcnt<NRD>();
        } else {
            wait_lgkmcnt<0>();
        }
        mfmas(ah1, al1, bh1, bl1);
        __builtin_amdgcn_sched_barrier(0);
        buf = nbuf;
    }

#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] *= p8_out_scale_of(g.a_exp);
    if (g.splitk > 1) {      // raw sums; splitk_reduce[_ln768]_kernel finishes them
        float* __restrict__ P = g.partial + (long)blockIdx.y * g.M * g.N;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) partial_tile32(g, P, m0 + wm * 64 + i * 32 + r, n0 + wn * (32 * TN) + j * 32, h, acc[i][j]);
        return;
    }
    const EpiCtx epi = make_epi(g, g.bias, g.C, g.R);
    if (epi.vec) {      // coalesced: transpose through this wave's slice of the (now idle) stage ring
        __builtin_amdgcn_s_barrier();      // every wave has consumed its last fragments
        constexpr int SLICE = 64 * (32 * TN + 4);
        static_assert(8 * SLICE * 4 <= STAGES * STAGE_BYTES, "epilogue LDS");
        epilogue_wave_lds<2, TN, true>(g, epi, reinterpret_cast<float*>(smem_p8) + wave * SLICE, m0 + wm * 64, n0 + wn * (32 * TN), lane, acc);
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) epilogue_tile32<true, true>(g, epi, m0 + wm * 64 + i * 32 + r, n0 + wn * (32 * TN) + j * 32, h, acc[i][j]);
    }
}


// ------------------------------------------------------------------------------------------------------------------
// Ping-pong kernel for the one- and two-round launches of the AR / VAE body (round 5): BM x BN tile, ONE workgroup of 8
// waves (WMW x WNW, wave tile 32 TMW x 32 TNW) per CU.  What differs from gemm_p8_mid_kernel:
//   * the two waves of a SIMD (waves w and w + 4: groups X = 0-3, Y = 4-7) run the SAME program one phase apart: a K step of a wave is
//     a LOAD phase L(k) - the DMA pieces of stage k + S - 1 first, then ALL fragments of stage k into registers (ds_read_b128 x
//     4 (TMW + TNW)) - and a COMPUTE phase C(k) - its 6 TMW TNW MFMAs back to back out of registers.  X runs L(k) while Y runs C(k - 1),
//     then X runs C(k) while Y runs L(k): one wave's matrix work sits under the other's fragment reads, address arithmetic, DMA issue
//     and waits (in the mid-grid kernel both waves of a SIMD reach their reads, their waits and the barrier together:
//     0.62 - 0.7 us per K step where the MFMAs need 0.37);
//   * a stage lives in registers one phase after it was read, so its LDS buffer is free again TWO phases after it landed: a ring of 3
//     stages keeps two K steps in flight (the in-phase structure reads a stage over a whole step and needs 4 buffers for that), which
//     is what makes a 256 x 128 tile (48 KB per stage: 0.75 of the operand bytes per flop of 128 x 128) fit the 160 KB of LDS;
//   * DMA pieces are buffer loads to LDS (descriptor in SGPRs, 32-bit lane offsets, hardware range check instead of a row clamp).
// Synchronisation per K step: ONE mandatory barrier O_k (behind a counted vmcnt wait: stage k + 1 has landed for every wave, every wave
// has read stage k) and, with BAR2, a second one E_k between the two phases of a step (keeps the groups exactly one phase apart;
// without it they pair up by themselves).  Accumulation order per output element is that of every other split kernel.
template <int BM, int BN, int WMW, int WNW, int STAGES, int BAR2, int TAG>
__global__ __launch_bounds__(512, 1) void gemm_p8_pp_kernel(const GemmArgs g) {
    constexpr int BK = 32;
    static_assert(WMW * WNW == 8, "8 waves");
    constexpr int TMW = BM / WMW / 32, TNW = BN / WNW / 32;
    constexpr int APIECES = BM / 64, WPIECES = BN / 64, NP = APIECES + WPIECES;      // 1-KiB DMA pieces per wave per stage
    constexpr int STAGE_BYTES = (BM + BN) * 128;
    static_assert(STAGES >= 3 && STAGES * STAGE_BYTES <= 160 * 1024, "LDS");
    static_assert((STAGES - 1) * NP <= 63, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_p8[];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const bool grp_y = wave >= 4;      // (wave-uniform) the half of the workgroup that runs one phase behind
    const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
    int tm, tn;
    {   // XCD-contiguous, grouped column-major tile order (as gemm_p8_mid_kernel)
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        const int idx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        constexpr int GM = (BM >= 256) ? 2 : 4;
        const int width = GM * tiles_n;
        const int group = idx / width, first_m = group * GM;
        const int gsz = min(tiles_m - first_m, GM);
        const int in_g = idx - group * width;
        tn = in_g / gsz;
        tm = first_m + (in_g - tn * gsz);
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int nk_all = g.K / BK, kt0 = (int)((long)nk_all * blockIdx.y / g.splitk);
    const int nk = (int)((long)nk_all * (blockIdx.y + 1) / g.splitk) - kt0;      // K tiles [kt0, kt0 + nk)

    // ---- DMA: piece = 8 rows x 128 B; wave w owns A pieces w * APIECES .. and W pieces w * WPIECES ..; lane = (row in piece, chunk) ----
    const int prow = lane >> 3, pchunk = lane & 7;
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0,
                                                                            (int)(((unsigned)(g.M - 1) * (unsigned)g.lda + (unsigned)g.K) * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned int*>(g.Wp), 0,
                                                                            (int)(((unsigned)(g.N - 1) * (unsigned)g.ldw + (unsigned)g.K) * 4u), 0x00020000);
    unsigned poff[NP];      // byte offset of this lane's 16 bytes of piece q at K tile 0 (rows beyond the operand: dropped by the range check -> zeros)
#pragma unroll
    for (int q = 0; q < APIECES; ++q) {
        const int ra = (wave * APIECES + q) * 8 + prow;
        poff[q] = (unsigned)(m0 + ra) * (unsigned)(g.lda * 4) + (unsigned)((pchunk ^ ((ra >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int q = 0; q < WPIECES; ++q) {
        const int rw = (wave * WPIECES + q) * 8 + prow;
        poff[APIECES + q] = (unsigned)(n0 + rw) * (unsigned)(g.ldw * 4) + (unsigned)((pchunk ^ ((rw >> 1) & 7)) << 4);
    }
    auto issue_stage = [&](int kt, int buf) __attribute__((always_inline)) {
        unsigned char* base = smem_p8 + buf * STAGE_BYTES;
        const int koff = (kt0 + kt) * (BK * 4);
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            unsigned char* dst = q < APIECES ? base + (wave * APIECES + q) * 1024 : base + BM * 128 + (wave * WPIECES + q - APIECES) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(q < APIECES ? a_rsrc : w_rsrc, (__attribute__((address_space(3))) void*)dst, 16, (int)poff[q], koff, 0, 0);
        }
    };

    const int wm = wave / WNW, wn = wave % WNW;
    const int r = lane & 31, h = lane >> 5;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_p8;
    unsigned a_off[2][2], w_off[2][2];      // [kb][hi / lo]; further 32-row tiles are + 4096 B (same swizzle key: an immediate)
    {
        const int arow = wm * (32 * TMW) + r, wrow = wn * (32 * TNW) + r;
        const int akey = (arow >> 1) & 7, wkey = (wrow >> 1) & 7;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int lo = 0; lo < 2; ++lo) {
                const int c = (kb * 2 + h) * 2 + lo;
                a_off[kb][lo] = lds0 + arow * 128 + ((c ^ akey) << 4);
                w_off[kb][lo] = lds0 + BM * 128 + wrow * 128 + ((c ^ wkey) << 4);
            }
    }

    f32x16 acc[TMW][TNW];
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < TNW; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    f16x8 ah[2][TMW], al[2][TMW], bh[2][TNW], bl[2][TNW];      // all fragments of one K step: [kb][tile]
    constexpr int NRD_KB = 2 * (TMW + TNW);                       // ds_read_b128 per k block (lgkmcnt is a 4-bit counter: with 16 reads of a phase in flight the 16th waits at issue)
    // LOAD phase of K step k: the pieces of stage k + S - 1 first (a Y wave's pieces have three phases to land, an X wave's four), then
    // every fragment of stage k, k block 0 first
    // (timing builds, results wrong: PP_ABL_NOMFMA one MFMA of a phase's 6 TMW TNW, PP_ABL_NODMA no DMA behind the prologue, PP_ABL_NOLDS
    // fragments read in the first K step only; tools/pp_ablation.sh)
    auto load_phase = [&](int k) __attribute__((always_inline)) {
#ifndef PP_ABL_NODMA
        if (k >= 1 && k + STAGES - 1 < nk) issue_stage(k + STAGES - 1, (k - 1) % STAGES);
#endif
        __builtin_amdgcn_sched_barrier(0);
#ifdef PP_ABL_NOLDS
        if (k > 0) return;
#endif
        const unsigned sb = (unsigned)((k % STAGES) * STAGE_BYTES);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const unsigned ahp = a_off[kb][0] + sb, alp = a_off[kb][1] + sb, whp = w_off[kb][0] + sb, wlp = w_off[kb][1] + sb;
            static_for<0, TNW>([&](auto j_tag) __attribute__((always_inline)) {
                constexpr int j = decltype(j_tag)::value;
                bh[kb][j] = lds_read128<j * 4096>(whp);
                bl[kb][j] = lds_read128<j * 4096>(wlp);
            });
            static_for<0, TMW>([&](auto i_tag) __attribute__((always_inline)) {
                constexpr int i = decltype(i_tag)::value;
                ah[kb][i] = lds_read128<i * 4096>(ahp);
                al[kb][i] = lds_read128<i * 4096>(alp);
            });
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // COMPUTE phase: k block 0 as soon as its fragments are in (the k block 1 reads fly on), then k block 1
    auto compute_phase = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            __builtin_amdgcn_sched_barrier(0);      // (the k block 0 MFMAs stay in front of the k block 1 wait)
            if (kb == 0) wait_lgkmcnt<(NRD_KB <= 15 ? NRD_KB : 15)>(); else wait_lgkmcnt<0>();
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int i = 0; i < TMW; ++i)
#pragma unroll
                    for (int j = 0; j < TNW; ++j) {      // weight fragment = A operand: C^T, see epilogue_tile32
#ifdef PP_ABL_NOMFMA
                        if (t + i + j + kb > 0) continue;
#endif
                        if (t == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[kb][j], ah[kb][i], acc[i][j], 0, 0, 0);
                        else if (t == 1) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[kb][j], ah[kb][i], acc[i][j], 0, 0, 0);
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[kb][j], al[kb][i], acc[i][j], 0, 0, 0);
                    }
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    {
        const int pro = min(STAGES, nk);
        for (int st = 0; st < pro; ++st) issue_stage(st, st);
        wait_vmcnt_units<NP>(pro - 1);      // stage 0 has landed for this wave ...
        __builtin_amdgcn_s_barrier();       // ... and for every wave
        __builtin_amdgcn_sched_barrier(0);
    }
    // ONE program for both groups (no divergent path around the fragment and accumulator registers: the asm reads' results may not cross
    // a control-flow join before their wait): L(k) | C(k) per K step; what differs is where a group's barriers sit.
    //   BAR2 = 0: X has its barrier O_k behind C(k), Y behind L(k) - so Y runs C(k - 1) | L(k) between two barriers while X runs L(k) | C(k);
    //   BAR2 = 1: a barrier behind every phase; Y takes one extra barrier first (it starts one phase late), X one extra at the end.
    // O_k (X: behind C(k), Y: behind L(k)) is taken behind the wait for this wave's pieces of stage k + 1: at most the stages issued
    // behind it, k + 2 .. min(k + S - 1, nk - 1), stay in flight.  A load phase's reads have returned before its barrier, so that
    // nobody's DMA can overwrite a buffer under a read in flight.
    if (BAR2 && grp_y) __builtin_amdgcn_s_barrier();
#pragma nounroll
    for (int k = 0; k < nk; ++k) {
        const bool more = k + 1 < nk;
        const int units = min(STAGES - 2, nk - 2 - k);
        load_phase(k);
        if (BAR2 || (grp_y && more)) {
            if (grp_y) {
                wait_lgkmcnt<0>();
                if (more) wait_vmcnt_units<NP>(units);
            }
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        compute_phase();
        if (BAR2 || (!grp_y && more)) {
            if (!grp_y && more) wait_vmcnt_units<NP>(units);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (BAR2 && !grp_y) __builtin_amdgcn_s_barrier();

#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < TNW; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] *= p8_out_scale_of(g.a_exp);
    if (g.splitk > 1) {      // raw sums; splitk_reduce[_ln768]_kernel finishes them
        float* __restrict__ P = g.partial + (long)blockIdx.y * g.M * g.N;
#pragma unroll
        for (int i = 0; i < TMW; ++i)
#pragma unroll
            for (int j = 0; j < TNW; ++j) partial_tile32(g, P, m0 + wm * (32 * TMW) + i * 32 + r, n0 + wn * (32 * TNW) + j * 32, h, acc[i][j]);
        return;
    }
    const EpiCtx epi = make_epi(g, g.bias, g.C, g.R);
    if (epi.vec) {      // coalesced: transpose through this wave's slice of the (now idle) stage ring
        __builtin_amdgcn_s_barrier();      // every wave has read its last fragments (no DMA is in flight: every stage was waited for)
        constexpr int SLICE = 32 * TMW * (32 * TNW + 4);
        static_assert(8 * SLICE * 4 <= STAGES * STAGE_BYTES, "epilogue LDS");
        epilogue_wave_lds<TMW, TNW, true>(g, epi, reinterpret_cast<float*>(smem_p8) + wave * SLICE, m0 + wm * (32 * TMW), n0 + wn * (32 * TNW), lane, acc);
    } else {
#pragma unroll
        for (int i = 0; i < TMW; ++i)
#pragma unroll
            for (int j = 0; j < TNW; ++j) epilogue_tile32<true, true>(g, epi, m0 + wm * (32 * TMW) + i * 32 + r, n0 + wn * (32 * TNW) + j * 32, h, acc[i][j]);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Small-grid variant for the AR/VAE scale steps (M = clips x 1..100 tokens): these launches are bound by the latency of ONE
// tile, and the register-staged kernel's single K tile of prefetch costs ~1 us per K step (a 64x64 tile over K = 768 took
// 23-28 us whatever M was).  Here a 64x64 (or 128x64) tile is fed by the same LDS-DMA ring as the big kernels, STAGES-1 K
// tiles in flight, 4 waves (2 x 2), split-K over blockIdx.y with the partial slabs of gemm_f32.hip.
template <int BM, int BN, int STAGES, int TAG>
__global__ __launch_bounds__(256, 2) void gemm_p8_sm_kernel(const GemmArgs g) {      // (, 2: a wave capped at 256 registers - hipcc then takes the VGPR form of the MFMAs, no accumulator moves out of AGPRs in the epilogue)
    constexpr int BK = 32;
    constexpr int TM = BM / 64, TN = BN / 64;              // 32x32 MFMA tiles per wave
    constexpr int APIECES = BM / 32, WPIECES = BN / 32;    // 1-KiB DMA pieces per wave per stage
    constexpr int NDMA = APIECES + WPIECES;
    constexpr int STAGE_BYTES = (BM + BN) * 128;
    constexpr int NRD = 2 * TM + 2 * TN, NMF = 3 * TM * TN;
    static_assert(STAGES >= 3 && STAGES * STAGE_BYTES <= 160 * 1024, "LDS");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_p8[];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
    int tm, tn;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        const int idx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        constexpr int GM = 8;
        const int width = GM * tiles_n;
        const int group = idx / width, first_m = group * GM;
        const int gsz = min(tiles_m - first_m, GM);
        const int in_g = idx - group * width;
        tn = in_g / gsz;
        tm = first_m + (in_g - tn * gsz);
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int nk_all = g.K / BK;   // split-K: this workgroup owns K tiles [kt0, kt0 + nk)
    const int kt0 = (int)((long)nk_all * blockIdx.y / g.splitk);
    const int nk = (int)((long)nk_all * (blockIdx.y + 1) / g.splitk) - kt0;

    const int prow = lane >> 3, pchunk = lane & 7;
    const unsigned char* src[NDMA];
#pragma unroll
    for (int q = 0; q < APIECES; ++q) {
        const int ra = (wave * APIECES + q) * 8 + prow;
        const int gm = min(m0 + ra, g.M - 1);
        src[q] = reinterpret_cast<const unsigned char*>(g.A) + ((long)gm * g.lda + (long)kt0 * BK) * 4 + ((pchunk ^ ((ra >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int q = 0; q < WPIECES; ++q) {
        const int rw = (wave * WPIECES + q) * 8 + prow;
        const int gn = min(n0 + rw, g.N - 1);
        src[APIECES + q] = reinterpret_cast<const unsigned char*>(g.Wp) + ((long)gn * g.ldw + (long)kt0 * BK) * 4 + ((pchunk ^ ((rw >> 1) & 7)) << 4);
    }
    const bool w_nt = g.w_nt != 0;      // (tuning) weight pieces with the non-temporal cache policy: they are read once per launch
    auto issue_piece = [&](int q, int kt, int buf) {
        unsigned char* dst = smem_p8 + buf * STAGE_BYTES +
                             (q < APIECES ? (wave * APIECES + q) * 1024 : BM * 128 + (wave * WPIECES + q - APIECES) * 1024);
        if (q >= APIECES && w_nt)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[q] + (long)kt * (BK * 4)),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 2);
        else
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[q] + (long)kt * (BK * 4)),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };

    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_p8;
    unsigned a_off[2][2], w_off[2][2];
    {
        const int arow = wm * (BM / 2) + r, wrow = wn * (BN / 2) + r;
        const int akey = (arow >> 1) & 7, wkey = (wrow >> 1) & 7;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int lo = 0; lo < 2; ++lo) {
                const int c = (kb * 2 + h) * 2 + lo;
                a_off[kb][lo] = lds0 + arow * 128 + ((c ^ akey) << 4);
                w_off[kb][lo] = lds0 + BM * 128 + wrow * 128 + ((c ^ wkey) << 4);
            }
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

#pragma unroll
    for (int st = 0; st < STAGES - 1; ++st)
        if (st < nk) {
#pragma unroll
            for (int q = 0; q < NDMA; ++q) issue_piece(q, st, st);
        }

    auto read_frags = [&](int buf, int kb, f16x8 (&ah)[TM], f16x8 (&al)[TM], f16x8 (&bh)[TN], f16x8 (&bl)[TN]) {
        const unsigned sb = buf * STAGE_BYTES;
        const unsigned ahp = a_off[kb][0] + sb, alp = a_off[kb][1] + sb, whp = w_off[kb][0] + sb, wlp = w_off[kb][1] + sb;
        bh[0] = lds_read128<0>(whp);
        ah[0] = lds_read128<0>(ahp);
        bl[0] = lds_read128<0>(wlp);
        al[0] = lds_read128<0>(alp);
        if constexpr (TM == 2) { ah[TM - 1] = lds_read128<4096>(ahp); al[TM - 1] = lds_read128<4096>(alp); }
        if constexpr (TN == 2) { bh[TN - 1] = lds_read128<4096>(whp); bl[TN - 1] = lds_read128<4096>(wlp); }
    };
    auto mfma_slot = [&](int sidx, const f16x8 (&ah)[TM], const f16x8 (&al)[TM], const f16x8 (&bh)[TN], const f16x8 (&bl)[TN]) {
        const int t = sidx / (TM * TN), i = (sidx / TN) % TM, j = sidx % TN;
        if (t == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], ah[i], acc[i][j], 0, 0, 0);
        else if (t == 1) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[j], ah[i], acc[i][j], 0, 0, 0);
        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], al[i], acc[i][j], 0, 0, 0);
    };
    auto mfmas = [&](const f16x8 (&ah)[TM], const f16x8 (&al)[TM], const f16x8 (&bh)[TN], const f16x8 (&bl)[TN]) {
#pragma unroll
        for (int sidx = 0; sidx < NMF; ++sidx) mfma_slot(sidx, ah, al, bh, bl);
    };

    f16x8 ah0[TM], al0[TM], bh0[TN], bl0[TN], ah1[TM], al1[TM], bh1[TN], bl1[TN];
    wait_vmcnt_units<NDMA>(min(STAGES - 2, nk - 1));
    __builtin_amdgcn_s_barrier();
    read_frags(0, 0, ah0, al0, bh0, bl0);
    int kt = 0, buf = 0;
    for (; kt + STAGES - 1 < nk; ++kt) {               // steady state
        const int nbuf = (buf + 1 == STAGES) ? 0 : buf + 1;
        const int fbuf = (buf == 0) ? STAGES - 1 : buf - 1;
        read_frags(buf, 1, ah1, al1, bh1, bl1);
        wait_lgkmcnt<NRD>();
        mfmas(ah0, al0, bh0, bl0);
        __builtin_amdgcn_sched_barrier(0);
        wait_vmcnt<(STAGES - 3) * NDMA>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        read_frags(nbuf, 0, ah0, al0, bh0, bl0);
        wait_lgkmcnt<NRD>();
#pragma unroll
        for (int sidx = 0; sidx < (NMF > NDMA ? NMF : NDMA); ++sidx) {
            if (sidx < NMF) mfma_slot(sidx, ah1, al1, bh1, bl1);
            if (sidx < NDMA) {
                __builtin_amdgcn_sched_barrier(0);
                issue_piece(sidx, kt + STAGES - 1, fbuf);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        buf = nbuf;
    }
    for (; kt < nk; ++kt) {                            // drain
        const int nbuf = (buf + 1 == STAGES) ? 0 : buf + 1;
        read_frags(buf, 1, ah1, al1, bh1, bl1);
        wait_lgkmcnt<NRD>();
        mfmas(ah0, al0, bh0, bl0);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) {
            wait_vmcnt_units<NDMA>(min(STAGES - 3, nk - 2 - kt));
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            read_frags(nbuf, 0, ah0, al0, bh0, bl0);
            wait_lgkmcnt<NRD>();
        } else {
            wait_lgkmcnt<0>();
        }
        mfmas(ah1, al1, bh1, bl1);
        __builtin_amdgcn_sched_barrier(0);
        buf = nbuf;
    }

#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] *= p8_out_scale_of(g.a_exp);
    if (g.splitk > 1) {
        float* __restrict__ P = g.partial + (long)blockIdx.y * g.M * g.N;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                partial_tile32(g, P, m0 + wm * (BM / 2) + i * 32 + r, n0 + wn * (BN / 2) + j * 32, h, acc[i][j]);
        return;
    }
    const EpiCtx epi = make_epi(g, g.bias, g.C, g.R);
    epilogue_tiles<true, true, TM, TN, true>(g, epi, m0 + wm * (BM / 2) + r, n0 + wn * (BN / 2), h, acc);
}

// ------------------------------------------------------------------------------------------------------------------
// Grouped positional convolution of wav2vec2 (hf:360-368: Conv1d(1024, 1024, k = 128, padding 64, groups 16) + GELU, added to its
// input) as a split GEMM whose activation operand never leaves the CU.  One workgroup = one 4-second chunk x one group of 64
// channels: the 200 output frames need input frames -64 .. 262 of the same 64 channels, i.e. a 327-row window of 256 bytes per row
// in the P8 format.  The window is split once into LDS (zero rows outside the chunk = the convolution's padding) and the operand of
// tap j is simply the window shifted by j rows; only the weights (2 MB per group, shared by all chunks of the group and kept in
// the XCD's L2 by the block -> (chunk, group) map) stream through a 4-stage LDS-DMA ring of 8 KB stages.  24 MFMAs per wave per
// 32-deep K step against 8 KB of DMA: bound by the matrix cores (the register-staged kernel that gathered the window from
// global memory for every tap ran at 160 TF/s, 2.0 ms per step).
//   8 waves, 4 (rows) x 2 (columns): wave tile 64 x 32 = 2 accumulators; 8 row tiles cover 256 rows (200 used)
//   window rows have a pitch of 272 bytes (256 + 16): the 16 lanes a ds_read_b128 serves together read 16 consecutive rows at one
//   chunk, i.e. 16 different bank quads, without any swizzle - so a lane needs ONE address register per K step (its row at this
//   tap) and every fragment is that register plus a compile-time offset (channel half, k block, hi / lo, second row tile);
//   the address arithmetic of an XOR-swizzled window cost more vector issue slots than the MFMAs leave free
constexpr int PC_WIN_ROWS = 384, PC_PITCH = 272, PC_WIN_BYTES = PC_WIN_ROWS * PC_PITCH, PC_STAGE = 64 * 128, PC_STAGES = 4;
template <int OFF>
__device__ __forceinline__ f16x8 lds_read128_big(unsigned addr) {      // offset up to 65535 (the DS instructions' 16-bit immediate)
    f16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
__global__ __launch_bounds__(512) void posconv_p8_kernel(const GemmArgs g, int T, int Ts) {
    constexpr int CG = 64, KT = 128, PAD = 64, NSTEP = KT * 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_p8[];
    unsigned char* win = smem_p8;
    unsigned char* ring = smem_p8 + PC_WIN_BYTES;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    // blocks b and b + 8 share an XCD: XCD x gets groups 2x and 2x + 1 only (4 MB of weights, its L2)
    const int bid = blockIdx.x, xcd = bid & 7, jj = bid >> 3;
    const int grp = xcd * 2 + (jj & 1), c = jj >> 1;
    const long row0 = (long)c * Ts;

    // ---- weight ring: piece = 8 rows x 128 B; wave w (of 8) owns rows 8w .. 8w+7 of the 64 output channels ----
    const int prow = lane >> 3, pchunk = lane & 7;
    const unsigned char* wsrc;
    {
        const int rw = wave * 8 + prow;
        wsrc = reinterpret_cast<const unsigned char*>(g.Wp) + ((long)(grp * CG + rw) * g.ldw) * 4 + ((pchunk ^ ((rw >> 1) & 7)) << 4);
    }
    auto issue_stage = [&](int ks, int buf) {
        unsigned char* dst = ring + buf * PC_STAGE + wave * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + (long)ks * 128),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
#pragma unroll
    for (int st = 0; st < PC_STAGES - 1; ++st) issue_stage(st, st);

    // ---- window: rows w = ts + 64 for ts in [-64, 320); valid ts in [0, T) come from X (fp32, split here), the rest are zeros ----
    for (int idx = tid; idx < PC_WIN_ROWS * 8; idx += 512) {
        const int w = idx >> 3, g8 = idx & 7;
        const int ts = w - PAD;
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
        if (ts >= 0 && ts < T) {
            const float* xp = g.A + (row0 + ts) * g.lda + grp * CG + g8 * 8;
            a = *reinterpret_cast<const f32x4*>(xp);
            b = *reinterpret_cast<const f32x4*>(xp + 4);
            p8_guard(g.status, a[0], a[1], a[2], a[3], p8_maxbits_of(g.a_exp));
            p8_guard(g.status, b[0], b[1], b[2], b[3], p8_maxbits_of(g.a_exp));
        }
        u32x2 h0, l0, h1, l1;
        split_f32x4(a, p8_scale_of(g.a_exp), h0, l0);
        split_f32x4(b, p8_scale_of(g.a_exp), h1, l1);
        const u32x4 hi = {h0[0], h0[1], h1[0], h1[1]}, lo = {l0[0], l0[1], l1[0], l1[1]};
        unsigned char* rowp = win + w * PC_PITCH + g8 * 32;
        *reinterpret_cast<u32x4*>(rowp) = hi;
        *reinterpret_cast<u32x4*>(rowp + 16) = lo;
    }

    // 8 waves = 4 (rows) x 2 (columns), wave tile 64 x 32 = 2 accumulators: two waves per SIMD, so one wave's barrier / wait / issue
    // gaps run under the other's MFMAs (with 4 waves of 128 x 32 the kernel took 1.41 ms against an MFMA floor of 0.67)
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_p8;
    const unsigned ring0 = lds0 + PC_WIN_BYTES;
    unsigned w_addr[2][2];              // [kb][hi/lo]: LDS address inside ring stage 0 (stage b = + b * PC_STAGE, an immediate)
    {
        const int wrow = wn * 32 + r, wkey = (wrow >> 1) & 7;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int lo = 0; lo < 2; ++lo) w_addr[kb][lo] = ring0 + wrow * 128 + ((((kb * 2 + h) * 2 + lo) ^ wkey) << 4);
    }
    // window address of this lane's activation row (tile 0) at tap 0, channel half 0, k block 0, hi: + tap * PC_PITCH per tap;
    // offsets: channel half 128, k block 64, lo 16, second row tile 32 * PC_PITCH
    const unsigned a_lane = lds0 + (wm * 64 + r) * PC_PITCH + h * 32;

    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    // Fragments are software-pipelined ACROSS the per-step barrier: the weight fragments of step ks+1 and the activation fragments
    // of its tile 0 are read during step ks.  Legal because the barrier at the top of step ks already guarantees that stage ks+1
    // has landed (every wave waited for its own piece of it first).  At most 12 LDS reads in flight (lgkmcnt is a 4-bit counter).
    // The loop is unrolled by 4 = ring depth, so ring buffer, channel half and register set are compile-time.
    f16x8 bh[2][2], bl[2][2];           // weights [register set][kb]
    f16x8 a0h[2][2], a0l[2][2];         // activations of tile 0 [register set][kb]
    f16x8 a1h[2], a1l[2];               // activations of tile 1 [kb]
    auto read_w = [&](auto buf_tag, auto set_tag) {
        constexpr int S = decltype(set_tag)::value, OFF = decltype(buf_tag)::value * PC_STAGE;
        bh[S][0] = lds_read128_big<OFF>(w_addr[0][0]); bl[S][0] = lds_read128_big<OFF>(w_addr[0][1]);
        bh[S][1] = lds_read128_big<OFF>(w_addr[1][0]); bl[S][1] = lds_read128_big<OFF>(w_addr[1][1]);
    };
    auto read_a0 = [&](unsigned abase, auto half_tag, auto set_tag) {
        constexpr int S = decltype(set_tag)::value, HO = decltype(half_tag)::value * 128;
        a0h[S][0] = lds_read128_big<HO>(abase);      a0l[S][0] = lds_read128_big<HO + 16>(abase);
        a0h[S][1] = lds_read128_big<HO + 64>(abase); a0l[S][1] = lds_read128_big<HO + 80>(abase);
    };
    auto read_a1 = [&](unsigned abase, auto half_tag) {
        constexpr int HO = decltype(half_tag)::value * 128 + 32 * PC_PITCH;
        a1h[0] = lds_read128_big<HO>(abase);      a1l[0] = lds_read128_big<HO + 16>(abase);
        a1h[1] = lds_read128_big<HO + 64>(abase); a1l[1] = lds_read128_big<HO + 80>(abase);
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;

    __syncthreads();                    // the window is complete (ds writes of every wave)
    wait_vmcnt<2>();                    // stage 0 landed (stages 1, 2: one piece each per wave, may be in flight)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    read_w(I0{}, I0{});
    read_a0(a_lane, I0{}, I0{});
    // one K step (J = ks & 3): on entry the reads of W and of tile 0 (register set J & 1) of this step are in flight, in that order
    auto kstep = [&](int ks, auto j_tag) {
        constexpr int J = decltype(j_tag)::value, S = J & 1;
        using SET = std::integral_constant<int, S>;
        using NSET = std::integral_constant<int, 1 - S>;
        using HALF = std::integral_constant<int, J & 1>;
        using NHALF = std::integral_constant<int, (J + 1) & 1>;
        using NBUF = std::integral_constant<int, (J + 1) & 3>;
        const unsigned abase = a_lane + (ks >> 1) * PC_PITCH;
        // stage ks+1 has landed when at most the youngest stage (one piece) is still in flight
        if (ks + 2 < NSTEP) wait_vmcnt<1>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();   // stage ks+1 landed for every wave; every wave is done with stage ks-1, whose buffer is refilled now
        __builtin_amdgcn_sched_barrier(0);
        if (ks + PC_STAGES - 1 < NSTEP) {
            issue_stage(ks + PC_STAGES - 1, (J + PC_STAGES - 1) & (PC_STAGES - 1));
            __builtin_amdgcn_sched_barrier(0);
        }
        read_a1(abase, HALF{});
        wait_lgkmcnt<4>();              // W and tile 0 are in registers
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[S][kb], a0h[S][kb], acc[0], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[S][kb], a0h[S][kb], acc[0], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[S][kb], a0l[S][kb], acc[0], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ks + 1 < NSTEP) {           // next step's weights and tile 0 into the other register set
            read_w(NBUF{}, NSET{});
            read_a0(a_lane + ((ks + 1) >> 1) * PC_PITCH, NHALF{}, NSET{});
            wait_lgkmcnt<8>();          // tile 1
        } else {
            wait_lgkmcnt<0>();
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[S][kb], a1h[kb], acc[1], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[S][kb], a1h[kb], acc[1], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[S][kb], a1l[kb], acc[1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int ks = 0; ks < NSTEP; ks += 4) {
        kstep(ks, I0{});
        kstep(ks + 1, I1{});
        kstep(ks + 2, I2{});
        kstep(ks + 3, I3{});
    }

    const EpiCtx epi = make_epi(g, g.bias, g.C, g.R);
#pragma nounroll
    for (int i = 0; i < 2; ++i) {                            // one copy of the epilogue code (common.h, epilogue_tiles)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[0][e] *= p8_out_scale_of(g.a_exp);
        const int t = wm * 64 + 32 * i + r;                  // frame inside the chunk; rows >= Ts belong to nobody
        const int row = t < Ts ? (int)(row0 + t) : g.M;
        epilogue_tile32(g, epi, row, grp * CG + wn * 32, h, acc[0]);
        acc[0] = acc[1];
    }
}
// g: A = input hidden states (fp32, row stride lda, rows chunk * Ts + t), Wp = packed weights [1024][128 * 64] (ldw = 8192),
// bias, C / R = output / residual (row stride ldc / ldr), act, M = n_chunks * Ts, N = 1024
void launch_posconv_p8(const GemmArgs& g, int n_chunks, int T, int Ts, hipStream_t s) {
    if (n_chunks <= 0) return;
    const size_t lds = PC_WIN_BYTES + PC_STAGES * PC_STAGE;
    static bool attr_set[64] = {};      // per device: more than the default 64 KB of dynamic LDS (this launch is never captured)
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&posconv_p8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set[dev] = true;
    }
    ARTALK_LAUNCH(posconv_p8_kernel, dim3(n_chunks * 16), dim3(512), lds, s, g, T, Ts);
}

template <int BM, int BN, int STAGES>
static void launch_p8_sm_cfg(const GemmArgs& g, hipStream_t s) {
    const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
    const size_t lds = STAGES * (BM + BN) * 128;
    if (g.graph_tag) ARTALK_LAUNCH((gemm_p8_sm_kernel<BM, BN, STAGES, 1>), dim3(tiles, g.splitk), dim3(256), lds, s, g);
    else ARTALK_LAUNCH((gemm_p8_sm_kernel<BM, BN, STAGES, 0>), dim3(tiles, g.splitk), dim3(256), lds, s, g);
}
bool gemm_p8_sm_eligible(const GemmArgs& g) {
    return g.Wp != nullptr && g.a_packed && g.amode == 0 && g.batch == 1 && g.K % 32 == 0 && (g.lda % 8) == 0;
}
// force_cfg 20: 64x64 x 4 stages (default); deep rings for the split-K launches of the small scale steps, where a workgroup's whole K
// slice should be in flight at once (the launch then costs one memory latency instead of one per K step): 23: 64x64 x 8 stages
// (128 KiB, one workgroup per CU), 24: 64x64 x 5 stages (80 KiB, two per CU).  (128x64, 64x128 and 128x128 tiles and a 3-stage ring
// were measured on the unsplit grids of the 50- / 100-token steps and never won: DESIGN.md section 6.)
void gemm_p8_prepare();
template <int STAGES>
static void launch_p8_mid(const GemmArgs& g, hipStream_t s) {
    const int tiles = ((g.M + 127) / 128) * ((g.N + 127) / 128);
    gemm_p8_prepare();
    if (g.graph_tag) ARTALK_LAUNCH((gemm_p8_mid_kernel<STAGES, 1>), dim3(tiles, g.splitk), dim3(512), STAGES * 256 * 128, s, g);
    else ARTALK_LAUNCH((gemm_p8_mid_kernel<STAGES, 0>), dim3(tiles, g.splitk), dim3(512), STAGES * 256 * 128, s, g);
}
template <int BM, int BN, int WMW, int WNW, int STAGES, int BAR2>
static void launch_p8_pp(const GemmArgs& g, hipStream_t s) {
    const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
    gemm_p8_prepare();
    const size_t lds = STAGES * (BM + BN) * 128;
    if (g.graph_tag) ARTALK_LAUNCH((gemm_p8_pp_kernel<BM, BN, WMW, WNW, STAGES, BAR2, 1>), dim3(tiles, g.splitk), dim3(512), lds, s, g);
    else ARTALK_LAUNCH((gemm_p8_pp_kernel<BM, BN, WMW, WNW, STAGES, BAR2, 0>), dim3(tiles, g.splitk), dim3(512), lds, s, g);
}
// what gemm_p8_pp_kernel's 32-bit DMA offsets need: operands below 4 GiB
bool gemm_p8_pp_ok(const GemmArgs& g) {
    return (double)g.M * g.lda * 4.0 < 4294967296.0 && (double)g.N * g.ldw * 4.0 < 4294967296.0;
}
void launch_gemm_p8_sm(const GemmArgs& g, hipStream_t s) {
    if (g.M <= 0 || g.N <= 0) return;
    int cfg = g.force_cfg;
    if (cfg >= 30 && cfg <= 33 && !gemm_p8_pp_ok(g)) cfg = 28;
    switch (cfg) {
        // ping-pong kernel (round 5): 30 / 31 = 256 x 128 tile, 3 stages, one / two barriers per K step; 33 = 128 x 128, 4 stages, two barriers
        // (one barrier per step and the 128 x 256 tile measured no better: profiles/r05_pp_gemm_sweep.log)
        case 30: launch_p8_pp<256, 128, 4, 2, 3, 0>(g, s); break;
        case 31: launch_p8_pp<256, 128, 4, 2, 3, 1>(g, s); break;
        case 33: launch_p8_pp<128, 128, 2, 4, 4, 1>(g, s); break;
        case 28: launch_p8_mid<4>(g, s); break;
        case 29: launch_p8_mid<5>(g, s); break;      // (experiment: the whole 160 KiB of LDS as a ring of 5 stages, four K tiles in flight)
        case 23: launch_p8_sm_cfg<64, 64, 8>(g, s); break;
        case 24: launch_p8_sm_cfg<64, 64, 5>(g, s); break;
        default: launch_p8_sm_cfg<64, 64, 4>(g, s); break;
    }
}

int gemm_p8_variant(const GemmArgs& g);
// host-side copy of make_epi()'s test: every row start and every run of 4 columns of C / bias / R / gate is 16-byte aligned
static bool epi_vec_host(const GemmArgs& g) {
    unsigned long long bits = (unsigned long long)(g.N | g.ldc) | ((unsigned long long)g.C >> 2);
    if (g.bias) bits |= (unsigned long long)g.bias >> 2;
    if (g.R) bits |= (unsigned long long)g.ldr | ((unsigned long long)g.R >> 2);
    if (g.gate) bits |= (unsigned long long)g.ldg | ((unsigned long long)g.gate >> 2);
    return (bits & 3) == 0;
}
void gemm_p8_prepare() {      // more than the default 64 KB of dynamic LDS (outside any capture)
    static bool done[64] = {};      // per device
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || done[dev]) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_p8_2wgp_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 256 * 128 + 16384);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_p8_2wgp_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 256 * 128 + 16384);
    const void* big4[] = {reinterpret_cast<const void*>(&gemm_p8_big_kernel<4, false, 0>), reinterpret_cast<const void*>(&gemm_p8_big_kernel<4, false, 1>),
                          reinterpret_cast<const void*>(&gemm_p8_big_kernel<4, true, 0>), reinterpret_cast<const void*>(&gemm_p8_big_kernel<4, true, 1>)};
    const void* big5[] = {reinterpret_cast<const void*>(&gemm_p8_big_kernel<5, false, 0>), reinterpret_cast<const void*>(&gemm_p8_big_kernel<5, false, 1>),
                          reinterpret_cast<const void*>(&gemm_p8_big_kernel<5, true, 0>), reinterpret_cast<const void*>(&gemm_p8_big_kernel<5, true, 1>)};
    for (const void* f : big4) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 512 * 128 + 4096);
    for (const void* f : big5) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 576 * 128 + 4096);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_p8_mid_kernel<4, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 256 * 128);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_p8_mid_kernel<4, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 256 * 128);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_p8_mid_kernel<5, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 5 * 256 * 128);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_p8_mid_kernel<5, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 5 * 256 * 128);
    {
        const void* pp[] = {reinterpret_cast<const void*>(&gemm_p8_pp_kernel<256, 128, 4, 2, 3, 0, 0>), reinterpret_cast<const void*>(&gemm_p8_pp_kernel<256, 128, 4, 2, 3, 0, 1>),
                            reinterpret_cast<const void*>(&gemm_p8_pp_kernel<256, 128, 4, 2, 3, 1, 0>), reinterpret_cast<const void*>(&gemm_p8_pp_kernel<256, 128, 4, 2, 3, 1, 1>)};
        for (const void* f : pp) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 384 * 128);
        const void* pq[] = {reinterpret_cast<const void*>(&gemm_p8_pp_kernel<128, 128, 2, 4, 4, 1, 0>), reinterpret_cast<const void*>(&gemm_p8_pp_kernel<128, 128, 2, 4, 4, 1, 1>)};
        for (const void* f : pq) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 256 * 128);
    }
    done[dev] = true;
}
// Production kernels: force_cfg 7 / 12 = gemm_p8_big_kernel with 256 x 256 / 320 x 256 tiles (persistent, one workgroup per CU),
// 8 = gemm_p8_2wgp_kernel (persistent 128 x 128, two workgroups per CU, deferred epilogue), -1 = gemm_p8_variant()'s choice;
// 13 = gemm_p8_256_kernel (the non-persistent 256 x 256 kernel, kept as the A/B baseline of the persistent one), 17 = 13 with
// wall-clock stamps (tools/gemm_p8_stamps.py).  cfg 8 needs the 16-byte epilogue path (epi_vec_host).
static int gemm_p8_cus() {      // workgroups of the one-per-CU persistent kernel
    static int n[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) return 256;
    if (!n[dev]) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        n[dev] = v - v % 8;      // a multiple of 8: tile t then runs on XCD t & 7
        if (n[dev] <= 0) n[dev] = 8;
    }
    return n[dev];
}
// CU count of a launch: the partition's (GemmArgs::cus, artalk_set_cu_mask) rounded down to a multiple of 8 and clamped to
// [8, the device's]: a count of 1..7 would otherwise size a grid of zero workgroups (an invalid launch, and a division by zero
// in the cost model), one above the device's would put more than one persistent workgroup on a CU.
static int p8_cus_of(const GemmArgs& g) {
    const int dev = gemm_p8_cus();
    if (g.cus <= 0) return dev;
    const int c = g.cus - g.cus % 8;
    return c < 8 ? 8 : (c > dev ? dev : c);
}
// what gemm_p8_big_kernel's epilogue and 32-bit source offsets need: whole 256-column tiles, the 16-byte epilogue path, no gate, a
// residual only with an fp32 result and no activation, no column groups, operands below 4 GiB
static bool p8_big_ok(const GemmArgs& g) {
    return g.N % 256 == 0 && g.ngrp == 0 && !g.gate && !(g.R && (g.c_p8 || g.act != ACT_NONE)) && epi_vec_host(g) && (double)g.M * g.lda * 4.0 < 4294967296.0 &&
           (double)g.N * g.ldw * 4.0 < 4294967296.0;
}
template <int TM>
static void launch_p8_big(const GemmArgs& g, hipStream_t s) {
    const int tiles = ((g.M + 64 * TM - 1) / (64 * TM)) * ((g.N + 255) / 256);
    const int cus = p8_cus_of(g);
    const size_t lds = 2 * (64 * TM + 256) * 128 + 4096;
    const dim3 grid(tiles < cus ? tiles : cus);
    if (g.R) {
        if (g.graph_tag) ARTALK_LAUNCH((gemm_p8_big_kernel<TM, true, 1>), grid, dim3(512), lds, s, g);
        else ARTALK_LAUNCH((gemm_p8_big_kernel<TM, true, 0>), grid, dim3(512), lds, s, g);
    } else {
        if (g.graph_tag) ARTALK_LAUNCH((gemm_p8_big_kernel<TM, false, 1>), grid, dim3(512), lds, s, g);
        else ARTALK_LAUNCH((gemm_p8_big_kernel<TM, false, 0>), grid, dim3(512), lds, s, g);
    }
}
void launch_gemm_p8(const GemmArgs& g0, hipStream_t s) {
    if (g0.M <= 0 || g0.N <= 0) return;
    // No range guard in the epilogues of the large-grid kernels (compiled out: epilogue_tile32<.., GUARD = false>; with it the dominant
    // kernel lost registers to it): their P8 results (q|k|v, FFN hidden) are consumed by the attention kernel and by the next GEMM +
    // LayerNorm, an out-of-range value turns into inf / NaN there, and those producers (attention output, LayerNorm) carry the
    // guard - one kernel later instead of in place.
    GemmArgs g = g0;
    g.status = nullptr;
    const int t128 = ((g.M + 127) / 128) * ((g.N + 127) / 128), t256sq = ((g.M + 255) / 256) * ((g.N + 255) / 256);
    int cfg = g.force_cfg;
    if (cfg != 7 && cfg != 8 && cfg != 12 && cfg != 13 && cfg != 17) {
        const int v = gemm_p8_variant(g);
        cfg = v == 1 ? 7 : (v == 2 ? 12 : 8);
    }
    if ((cfg == 7 || cfg == 12) && !p8_big_ok(g)) cfg = epi_vec_host(g) ? 8 : 13;      // (tuning / tests only: the dispatch checks p8_big_ok itself)
    if (cfg == 8 && !epi_vec_host(g)) cfg = 13;     // (no launch of the path gets here: every large-grid result is 16-byte aligned)
    if (cfg == 7 || cfg == 12) {
        gemm_p8_prepare();
        // Round quantisation, the dispatch-only answer (round 5): a launch of 320-row tiles whose LAST round is partial
        // goes out as two launches when the rows of that round fit ONE round of 256-row tiles - whole rounds of 320 x 256 first, then the
        // remaining rows as 256 x 256 tiles (0.8 of a round's time).  With the shapes of batch 32 that is the encoder's FFN-in GEMM only
        // (960 tiles = 3.75 rounds -> 768 tiles + 240 tiles of 256 rows: 3.8 rounds of time instead of 4); q|k|v (2.81 rounds) has no such
        // split: 504 + 276 tiles, the second launch would spill into a second round.  Row ranges are independent and an element's K order
        // does not depend on its tile: bit-identical.  MEASURED, NO GAIN (same box, three alternating pairs, profiles/r05_split2_ab.log:
        // encoder 36.08 / 36.01 / 36.00 ms without, 36.03 / 36.09 / 35.98 ms with): in the partial last round of the single launch a quarter
        // of the CUs are idle and the power-managed clock of the others rises, which already recovers most of what
        // the quantisation costs on paper (5 % of the launch).  Off by default; ARTALK_P8_SPLIT2=1 switches it on (A/B).
        static const int split2 = getenv("ARTALK_P8_SPLIT2") ? atoi(getenv("ARTALK_P8_SPLIT2")) : 0;
        if (cfg == 12 && split2 && g0.force_cfg < 0 && g.cmap.rpb == INT_MAX && !g.gate) {
            const int cus = p8_cus_of(g), tn = (g.N + 255) / 256, tm5 = (g.M + 319) / 320;
            const int tiles5 = tm5 * tn, full = tiles5 / cus * cus;
            if (full > 0 && tiles5 > full) {
                const int rows1 = full / tn;                       // row tiles of the first launch (whole rounds or a little less)
                const int M1 = rows1 * 320, M2 = g.M - M1;
                const int tiles2 = ((M2 + 255) / 256) * tn;
                const double t_one = std::ceil((double)tiles5 / cus), t_two = std::ceil((double)rows1 * tn / cus) + 0.8 * std::ceil((double)tiles2 / cus) + 0.04;
                if (rows1 > 0 && M2 > 0 && tiles2 <= cus && t_two < t_one) {
                    GemmArgs a = g, b = g;
                    a.M = M1;
                    b.M = M2; b.A = g.A + (long)M1 * g.lda; b.C = g.C + (long)M1 * g.ldc;
                    if (g.R) b.R = g.R + (long)M1 * g.ldr;
                    launch_p8_big<5>(a, s);
                    launch_p8_big<4>(b, s);
                    return;
                }
            }
        }
        if (cfg == 7) launch_p8_big<4>(g, s); else launch_p8_big<5>(g, s);
    } else if (cfg == 13) {
        ARTALK_LAUNCH((gemm_p8_256_kernel<0>), dim3(t256sq), dim3(512), 8 * 64 * 68 * 4, s, g);
    } else if (cfg == 17) {
        ARTALK_LAUNCH((gemm_p8_256_kernel<6>), dim3(t256sq), dim3(512), 8 * 64 * 68 * 4, s, g);
    } else {
        // residual tiles deferred too (ARTALK_P8_RES_DEFER=0 finishes them at once; tests compare both): 16 KiB more LDS, still two
        // workgroups per CU
        static const int res_defer = getenv("ARTALK_P8_RES_DEFER") ? atoi(getenv("ARTALK_P8_RES_DEFER")) : 1;
        g.res_lds = (res_defer && g.R && !g.gate && !g.c_p8) ? 1 : 0;
        if (g.res_lds) gemm_p8_prepare();
        const size_t lds = 2 * 256 * 128 + (g.res_lds ? 16384 : 0);
        const int slots = 2 * p8_cus_of(g);      // two workgroups per CU
        if (g.graph_tag) ARTALK_LAUNCH((gemm_p8_2wgp_kernel<1>), dim3(t128 < slots ? t128 : slots), dim3(256), lds, s, g);
        else ARTALK_LAUNCH((gemm_p8_2wgp_kernel<0>), dim3(t128 < slots ? t128 : slots), dim3(256), lds, s, g);
    }
}
// 0: persistent two-workgroup 128x128 kernel, 1 / 2: persistent big-tile kernel with 256x256 / 320x256 tiles.  A cost model picks the
// configuration with the shortest critical path: rounds x (K steps x time per step + epilogue), calibrated on the encoder shapes of
// batch 32 (profiles/r03_gemm_f16s_bench.log): a 64-row slab of a 256-column tile takes 0.58 us per 32-deep K step and ~3 us of
// epilogue, a pair of co-resident 128x128 tiles 1.1 - 1.5 us per K step (1.3 taken).  At M = 19200 this gives 320x256 everywhere: N = 1024 (out-projection,
// FFN-out) is ONE round of 240 tiles on 256 CUs where the 128x128 kernel's 512 workgroups need 2.34 rounds (FFN-out 482 -> 406 us),
// q|k|v 2.81 rounds instead of the 256x256 tile's 3.52 (403 -> 328 us).  ARTALK_P8_BIG = 0 / 1 / 2 forces a variant (A/B runs).
int gemm_p8_variant(const GemmArgs& g) {
    static const int forced = getenv("ARTALK_P8_BIG") ? atoi(getenv("ARTALK_P8_BIG")) : -1;
    const bool big_ok = p8_big_ok(g);
    if (forced >= 0) return big_ok ? forced : 0;
    if (!big_ok) return 0;
    const double nk = g.K / 32, cus = p8_cus_of(g);
    auto big = [&](int TM) {
        const double tiles = (double)((g.M + 64 * TM - 1) / (64 * TM)) * ((g.N + 255) / 256);
        return std::ceil(tiles / cus) * (TM * nk * 0.58 + 3.0 * TM + 2.0);
    };
    const double t128 = (double)((g.M + 127) / 128) * ((g.N + 127) / 128);
    const double c0 = std::ceil(t128 / (2 * cus)) * (nk * 1.30 + 4.0), c1 = big(4), c2 = big(5);
    if (c0 < c1 && c0 < c2) return 0;
    return c1 < c2 ? 1 : 2;
}
bool gemm_p8_eligible(const GemmArgs& g) {
    return g.Wp != nullptr && g.a_packed && g.amode == 0 && g.batch == 1 && g.splitk == 1 && g.K % 32 == 0 && (g.lda % 8) == 0 &&
           (long)((g.M + 127) / 128) * ((g.N + 127) / 128) >= 384 && (g.ngrp == 0 || (g.ngrp % 128 == 0 && g.N % g.ngrp == 0));
}

template <int BM, int BN, int WM, int WN>
static void launch_f16s_cfg(const GemmArgs& g, hipStream_t s) {
    const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
    const size_t lds = 2 * (BM + BN) * 144;
    if (g.amode == 1) {
        ARTALK_LAUNCH((gemm_f16s_kernel<BM, BN, WM, WN, 0, 0, 1>), dim3(tiles, 1, g.batch), dim3(256), lds, s, g);
    } else if (g.a_packed) {
        if (g.graph_tag) ARTALK_LAUNCH((gemm_f16s_kernel<BM, BN, WM, WN, 1, 1>), dim3(tiles, g.splitk), dim3(256), lds, s, g);
        else ARTALK_LAUNCH((gemm_f16s_kernel<BM, BN, WM, WN, 1, 0>), dim3(tiles, g.splitk), dim3(256), lds, s, g);
    } else {
        if (g.graph_tag) ARTALK_LAUNCH((gemm_f16s_kernel<BM, BN, WM, WN, 0, 1>), dim3(tiles, g.splitk), dim3(256), lds, s, g);
        else ARTALK_LAUNCH((gemm_f16s_kernel<BM, BN, WM, WN, 0, 0>), dim3(tiles, g.splitk), dim3(256), lds, s, g);
    }
}

// 0: 128x128 (dominant kernel of the split mode), 1: 64x64
int gemm_f16s_config(const GemmArgs& g) {
    if (g.force_cfg >= 0) return g.force_cfg;
    if (g.amode == 1) return 1;   // N = 64 per group
    // 64x64 tiles (4-5 workgroups/CU) beat 128x128 (2/CU) until the grid is several waves deep: M=3200,N=3072,K=768 runs at
    // 210 vs 135 TF/s (profiles/r01_gemm_f16s_bench.log)
    const long t128 = (long)((g.M + 127) / 128) * ((g.N + 127) / 128);
    return t128 >= 1024 ? 0 : 1;
}

bool gemm_f16s_eligible(const GemmArgs& g) {
    if (g.amode == 1) return g.Wp != nullptr && !g.a_packed && g.splitk == 1 && g.K % 32 == 0 && g.pc_cin % 32 == 0;
    return g.Wp != nullptr && g.batch == 1 && g.K % 32 == 0;
}

void launch_gemm_f16s(const GemmArgs& g, hipStream_t s) {
    if (g.M <= 0 || g.N <= 0) return;
    if (gemm_f16s_config(g) == 0) launch_f16s_cfg<128, 128, 2, 2>(g, s);
    else launch_f16s_cfg<64, 64, 2, 2>(g, s);
}

}  // namespace artalk
