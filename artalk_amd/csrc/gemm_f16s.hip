// fp32-accurate GEMM on the fp16 matrix cores by operand splitting ("f16x3"): C = epilogue(A[M,K] * W[N,K]^T).
//
// Every fp32 operand x is represented as hi + lo/2048 with hi = f16(x) and lo = f16((x - hi) * 2048): 22 significand
// bits, the residual is scaled so it stays in fp16's normal range.  A product a*b then needs three exact fp16 products
//   a_hi*b_hi                (accumulator "main")
//   a_hi*b_lo + a_lo*b_hi    (accumulator "cross", scaled by 2^-11 once in the epilogue)
// accumulated in fp32 by v_mfma_f32_32x32x16_f16; the dropped a_lo*b_lo term is 2^-22 relative.  Measured on random
// data (K = 1024): max error 8e-8 of the output scale, below the 5e-7 accumulation-order noise of any fp32 GEMM, and
// every reference golden stays decision-exact (tests/test_e2e_gpu.py runs both precision modes).  Three fp16 MFMAs
// cover 16 k in 96 cycles where v_mfma_f32_32x32x2_f32 needs 512: the MFMA time drops 5.3x and the kernel becomes
// staging-bound.  fp16 range: |x| must stay below 65504 (all GEMM inputs of the path are LayerNorm/GELU/attention
// outputs or weights); operands are never denormal-flushed (residuals are rescaled).
//
// Memory layout is unchanged: activations stay fp32 in HBM and are split while they are staged into LDS; weights are
// pre-split once at load into the "P8" format (every 8 elements -> 16 bytes of hi halves + 16 bytes of lo halves), so
// a packed matrix has the same size, pitch and 16-byte loads as the fp32 one and a 16-byte chunk is an MFMA fragment.  LDS rows hold the hi plane (BK halves) followed by the
// lo plane, padded to 144 bytes (conflict-free ds_read_b128 for the 32x32x16 operand map: lane (r,h) holds k = 8h..8h+7).
#include "common.h"

namespace artalk {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr float kLoScale = 2048.0f, kLoInv = 1.0f / 2048.0f;

__device__ __forceinline__ void split_f32x4(const f32x4 x, u32x2& hi, u32x2& lo) {
    f16x4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        h[e] = (_Float16)x[e];
        l[e] = (_Float16)((x[e] - (float)h[e]) * kLoScale);
    }
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
__global__ __launch_bounds__(256) void pack_split_kernel(const float* __restrict__ w, unsigned int* __restrict__ out, long n) {
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += (long)gridDim.x * 2048) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(w + i), b = *reinterpret_cast<const f32x4*>(w + i + 4);
        u32x2 h0, l0, h1, l1;
        split_f32x4(a, h0, l0);
        split_f32x4(b, h1, l1);
        u32x4 hi = {h0[0], h0[1], h1[0], h1[1]}, lo = {l0[0], l0[1], l1[0], l1[1]};
        *reinterpret_cast<u32x4*>(out + i) = hi;
        *reinterpret_cast<u32x4*>(out + i + 4) = lo;
    }
}
void launch_pack_split(const float* w, unsigned int* out, long n, hipStream_t s) {
    if (n <= 0) return;
    const long blocks = (n / 8 + 255) / 256;   // n % 8 == 0 (every packed tensor has an inner dimension that is a multiple of 32)
    hipLaunchKernelGGL(pack_split_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, s, w, out, n);
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
                split_f32x4(__builtin_bit_cast(f32x4, ra[i]), hi, lo);
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

    f32x16 accm[TM][TN], accx[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { accm[i][j][e] = 0.f; accx[i][j][e] = 0.f; }

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
                    accm[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], ah[i], accm[i][j], 0, 0, 0);
                    accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[j], ah[i], accx[i][j], 0, 0, 0);
                    accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], al[i], accx[i][j], 0, 0, 0);
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
                for (int e = 0; e < 16; ++e) accm[i][j][e] += accx[i][j][e] * kLoInv;
                partial_tile32(g, P, m0 + wm * (BM / WM) + i * 32 + r, n0 + wn * (BN / WN) + j * 32, h, accm[i][j]);
            }
        return;
    }
    const EpiCtx epi = make_epi(g, g.bias ? g.bias + z * g.sBias : nullptr, g.C + z * g.sC, g.R ? g.R + z * g.sR : nullptr);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int e = 0; e < 16; ++e) accm[i][j][e] += accx[i][j][e] * kLoInv;
            epilogue_tile32(g, epi, m0 + wm * (BM / WM) + i * 32 + r, n0 + wn * (BN / WN) + j * 32, h, accm[i][j]);
        }
}

// ------------------------------------------------------------------------------------------------------------------
// LDS-DMA pipeline variant for the large GEMMs: BOTH operands already in P8 (A written in P8 by its producer kernel), so
// staging is a pure byte copy and goes global -> LDS directly (global_load_lds_dwordx4: no staging registers), STAGES-1
// K tiles in flight behind counted vmcnt waits and ONE raw s_barrier per K step.  The register-staged kernel above has a
// single K tile of prefetch and is load-latency bound; here 3 tiles (96 KiB) stay in flight per CU.
//   tile 128x128, BK = 32, 8 waves (2 x 4, each 64 x 32: 2 m-tiles x 1 n-tile, main + cross accumulators = 64 regs)
//   LDS stage: A 128 rows x 128 B, then W 128 rows x 128 B, rows unpadded (a DMA wave-instruction writes 1 KiB = 8 rows);
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
    if (units >= 3) wait_vmcnt<3 * UNIT>();
    else if (units == 2) wait_vmcnt<2 * UNIT>();
    else if (units == 1) wait_vmcnt<UNIT>();
    else wait_vmcnt<0>();
}

// BM = 128: waves 2 x 4, wave tile 64 x 32, 4 stages of 32 KiB.  BM = 256: waves 4 x 2, wave tile 64 x 64, 3 stages of 48 KiB
// (3/4 of the operand bytes per flop).
// ABL == 6 (tools/gemm_p8_stamps.py): the same kernel + wall-clock stamps (100 MHz) per workgroup in g.partial as 8 x u64:
// entry, first stage landed, main loop done, epilogue done, HW_ID, XCC_ID.
template <int BM, int STAGES, int ABL = 0>
__global__ __launch_bounds__(512, 1) void gemm_p8_kernel(const GemmArgs g) {
    constexpr int BN = 128, BK = 32;
    constexpr int WN_WAVES = (BM == 128) ? 4 : 2;          // waves along N
    constexpr int TN = BN / WN_WAVES / 32;                 // n-tiles per wave (1 or 2); m-tiles per wave = 2
    constexpr int APIECES = BM / 64, WPIECES = 2;          // 1-KiB DMA pieces per wave per stage
    constexpr int NDMA = APIECES + WPIECES;
    constexpr int STAGE_BYTES = (BM + BN) * 128;
    static_assert(STAGES * STAGE_BYTES <= 160 * 1024, "LDS");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_p8[];

    const int tid = threadIdx.x;
    unsigned long long* stamps = nullptr;
    if constexpr (ABL == 6) {
        stamps = reinterpret_cast<unsigned long long*>(g.partial) + (long)blockIdx.x * 8;
        if (tid == 0) {
            stamps[0] = wall_clock64();
            stamps[4] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));    // HW_REG_HW_ID, all 32 bits
            stamps[5] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));   // HW_REG_XCC_ID
        }
    }
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
        const long koff = (long)kt * (BK * 4);
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

    f32x16 accm[2][TN], accx[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { accm[i][j][e] = 0.f; accx[i][j][e] = 0.f; }

    const int nk = g.K / BK;
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
        bh[0] = lds_read128<0>(whp);
        ah[0] = lds_read128<0>(ahp);
        bl[0] = lds_read128<0>(wlp);
        al[0] = lds_read128<0>(alp);
        ah[1] = lds_read128<4096>(ahp);
        al[1] = lds_read128<4096>(alp);
        if constexpr (TN == 2) {
            bh[TN - 1] = lds_read128<4096>(whp);
            bl[TN - 1] = lds_read128<4096>(wlp);
        }
    };
    constexpr int NMF = 2 * TN * 3;    // MFMAs per half step
    auto mfma_slot = [&](int sidx, const f16x8 (&ah)[2], const f16x8 (&al)[2], const f16x8 (&bh)[TN], const f16x8 (&bl)[TN]) {
        const int i = sidx / (3 * TN), j = (sidx / 3) % TN, t = sidx % 3;     // weight fragment = A operand: C^T, see epilogue_tile32
        if (t == 0) accm[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], ah[i], accm[i][j], 0, 0, 0);
        else if (t == 1) accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[j], ah[i], accx[i][j], 0, 0, 0);
        else accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], al[i], accx[i][j], 0, 0, 0);
    };
    auto mfmas = [&](const f16x8 (&ah)[2], const f16x8 (&al)[2], const f16x8 (&bh)[TN], const f16x8 (&bl)[TN]) {
#pragma unroll
        for (int sidx = 0; sidx < NMF; ++sidx) mfma_slot(sidx, ah, al, bh, bl);
    };

    f16x8 ah0[2], al0[2], bh0[TN], bl0[TN], ah1[2], al1[2], bh1[TN], bl1[TN];
    wait_vmcnt_units<NDMA>(min(STAGES - 2, nk - 1));   // stage 0 landed (the prologue left up to STAGES-2 younger stages in flight)
    __builtin_amdgcn_s_barrier();
    if constexpr (ABL == 6) { if (tid == 0) stamps[1] = wall_clock64(); }
    read_frags(0, 0, ah0, al0, bh0, bl0);
    int kt = 0, buf = 0;                               // buf = kt % STAGES
    for (; kt + STAGES - 1 < nk; ++kt) {               // steady state: stage kt+S-1 still to be fetched
        const int nbuf = (buf + 1 == STAGES) ? 0 : buf + 1;
        const int fbuf = (buf == 0) ? STAGES - 1 : buf - 1;     // (kt + S - 1) % S
        read_frags(buf, 1, ah1, al1, bh1, bl1);
        wait_lgkmcnt<NRD>();                           // the kb=0 fragments (issued half a step ago) are in; the kb=1 reads fly on
        mfmas(ah0, al0, bh0, bl0);
        __builtin_amdgcn_sched_barrier(0);
        wait_vmcnt<(STAGES - 3) * NDMA>();             // stage kt+1 landed for this wave (stages kt+2 .. kt+S-2 may still fly)
        __builtin_amdgcn_s_barrier();                  // ... and for every wave; every wave is done with stage kt-1
        __builtin_amdgcn_sched_barrier(0);
        read_frags(nbuf, 0, ah0, al0, bh0, bl0);
        wait_lgkmcnt<NRD>();                           // kb=1 fragments of stage kt
#pragma unroll
        for (int sidx = 0; sidx < NMF; ++sidx) {
            mfma_slot(sidx, ah1, al1, bh1, bl1);
            if (sidx < NDMA) {
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
    if constexpr (ABL == 6) { if (tid == 0) stamps[2] = wall_clock64(); }

    const EpiCtx epi = make_epi(g, g.bias, g.C, g.R);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int e = 0; e < 16; ++e) accm[i][j][e] += accx[i][j][e] * kLoInv;
            epilogue_tile32(g, epi, m0 + wm * 64 + i * 32 + r, n0 + wn * (32 * TN) + j * 32, h, accm[i][j]);
        }
    if constexpr (ABL == 6) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (tid == 0) stamps[3] = wall_clock64();
    }
}

void launch_gemm_p8(const GemmArgs& g, hipStream_t s) {
    if (g.M <= 0 || g.N <= 0) return;
    const int t128 = ((g.M + 127) / 128) * ((g.N + 127) / 128), t256 = ((g.M + 255) / 256) * ((g.N + 127) / 128);
    switch (g.force_cfg) {   // tuning: 3 / 5 = pipeline depth of the 128x128 kernel, 6 = 256x128 tiles (3 stages of 48 KiB)
        case 3: hipLaunchKernelGGL((gemm_p8_kernel<128, 3>), dim3(t128), dim3(512), 3 * 256 * 128, s, g); break;
        case 5: hipLaunchKernelGGL((gemm_p8_kernel<128, 5>), dim3(t128), dim3(512), 5 * 256 * 128, s, g); break;
        case 6: hipLaunchKernelGGL((gemm_p8_kernel<256, 3>), dim3(t256), dim3(512), 3 * 384 * 128, s, g); break;
        case 16: hipLaunchKernelGGL((gemm_p8_kernel<128, 4, 6>), dim3(t128), dim3(512), 4 * 256 * 128, s, g); break;
        default: hipLaunchKernelGGL((gemm_p8_kernel<128, 4>), dim3(t128), dim3(512), 4 * 256 * 128, s, g); break;
    }
}
bool gemm_p8_eligible(const GemmArgs& g) {
    return g.Wp != nullptr && g.a_packed && g.amode == 0 && g.batch == 1 && g.splitk == 1 && g.K % 32 == 0 && (g.lda % 8) == 0 &&
           (long)((g.M + 127) / 128) * ((g.N + 127) / 128) >= 384;
}

template <int BM, int BN, int WM, int WN>
static void launch_f16s_cfg(const GemmArgs& g, hipStream_t s) {
    const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
    const size_t lds = 2 * (BM + BN) * 144;
    if (g.amode == 1) {
        hipLaunchKernelGGL((gemm_f16s_kernel<BM, BN, WM, WN, 0, 0, 1>), dim3(tiles, 1, g.batch), dim3(256), lds, s, g);
    } else if (g.a_packed) {
        if (g.graph_tag) hipLaunchKernelGGL((gemm_f16s_kernel<BM, BN, WM, WN, 1, 1>), dim3(tiles, g.splitk), dim3(256), lds, s, g);
        else hipLaunchKernelGGL((gemm_f16s_kernel<BM, BN, WM, WN, 1, 0>), dim3(tiles, g.splitk), dim3(256), lds, s, g);
    } else {
        if (g.graph_tag) hipLaunchKernelGGL((gemm_f16s_kernel<BM, BN, WM, WN, 0, 1>), dim3(tiles, g.splitk), dim3(256), lds, s, g);
        else hipLaunchKernelGGL((gemm_f16s_kernel<BM, BN, WM, WN, 0, 0>), dim3(tiles, g.splitk), dim3(256), lds, s, g);
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
