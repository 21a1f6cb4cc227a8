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
// pre-split once at load into a packed (hi | lo << 16) word per element, so a packed matrix has the same size, the
// same indexing and the same 16-byte loads as the fp32 one.  LDS rows hold the hi plane (BK halves) followed by the
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
__device__ __forceinline__ void unpack_x4(const u32x4 p, u32x2& hi, u32x2& lo) {
    hi[0] = __builtin_amdgcn_perm(p[1], p[0], 0x05040100u);   // low halves of p0,p1
    hi[1] = __builtin_amdgcn_perm(p[3], p[2], 0x05040100u);
    lo[0] = __builtin_amdgcn_perm(p[1], p[0], 0x07060302u);   // high halves
    lo[1] = __builtin_amdgcn_perm(p[3], p[2], 0x07060302u);
}

// fp32 -> packed (f16 hi | f16 lo << 16), elementwise
__global__ __launch_bounds__(256) void pack_split_kernel(const float* __restrict__ w, unsigned int* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float x = w[i];
        const _Float16 h = (_Float16)x;
        const _Float16 l = (_Float16)((x - (float)h) * kLoScale);
        out[i] = (unsigned int)__builtin_bit_cast(unsigned short, h) | ((unsigned int)__builtin_bit_cast(unsigned short, l) << 16);
    }
}
void launch_pack_split(const float* w, unsigned int* out, long n, hipStream_t s) {
    if (n <= 0) return;
    const long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(pack_split_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, s, w, out, n);
}

template <int BM, int BN, int WM, int WN, int APK, int TAG>
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
    const float* __restrict__ A = g.A;
    const unsigned int* __restrict__ Wp = g.Wp;

    u32x4 ra[A_LD];   // fp32 bits (APK=0: split while staging) or packed (hi | lo<<16) words (APK=1: producer already split)
    u32x4 rb[B_LD];
    const int lrow = tid >> 3, lc4 = (tid & 7) * 4;

    auto gload = [&](int kt) {
        const int k = kt * BK + lc4;
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            const int gm = m0 + lrow + i * 32;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (gm < g.M) v = *reinterpret_cast<const u32x4*>(A + (long)gm * g.lda + k);
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
            u32x2 hi, lo;
            if (APK) unpack_x4(ra[i], hi, lo); else split_f32x4(__builtin_bit_cast(f32x4, ra[i]), hi, lo);
            unsigned char* p = As + (buf * BM + lrow + i * 32) * ROWB + lc4 * 2;
            *reinterpret_cast<u32x2*>(p) = hi;
            *reinterpret_cast<u32x2*>(p + 64) = lo;
        }
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            u32x2 hi, lo;
            unpack_x4(rb[i], hi, lo);
            unsigned char* p = Bs + (buf * BN + lrow + i * 32) * ROWB + lc4 * 2;
            *reinterpret_cast<u32x2*>(p) = hi;
            *reinterpret_cast<u32x2*>(p + 64) = lo;
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
                    accm[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], accm[i][j], 0, 0, 0);
                    accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], accx[i][j], 0, 0, 0);
                    accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], accx[i][j], 0, 0, 0);
                }
        }
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }

    if (g.splitk > 1) {   // raw partial sums; the epilogue runs in splitk_reduce_kernel
        float* __restrict__ P = g.partial + (long)blockIdx.y * g.M * g.N;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * (BN / WN) + j * 32 + r;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = m0 + wm * (BM / WM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (row < g.M && col < g.N) P[(long)row * g.N + col] = accm[i][j][e] + accx[i][j][e] * kLoInv;
                }
        }
        return;
    }
    const float* __restrict__ bias = g.bias;
    float* __restrict__ C = g.C;
    const float* R = g.R;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * (BN / WN) + j * 32 + r;
        const bool cok = col < g.N;
        const float bv = (bias && cok) ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * (BM / WM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (row < g.M && cok) {
                    float v = (accm[i][j][e] + accx[i][j][e] * kLoInv) + bv;
                    v = apply_act_rt(v, g.act);
                    if (g.gate) v *= g.gate[(long)map_row(g.gmap, row) * g.ldg + col];
                    const long crow = map_row(g.cmap, row);
                    if (R) v += R[crow * g.ldr + col];
                    C[crow * g.ldc + col] = v;
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN>
static void launch_f16s_cfg(const GemmArgs& g, hipStream_t s) {
    const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
    const size_t lds = 2 * (BM + BN) * 144;
    if (g.a_packed) {
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
    const long t128 = (long)((g.M + 127) / 128) * ((g.N + 127) / 128);
    return t128 >= 512 ? 0 : 1;
}

bool gemm_f16s_eligible(const GemmArgs& g) {
    return g.Wp != nullptr && g.amode == 0 && g.batch == 1 && g.K % 32 == 0;
}

void launch_gemm_f16s(const GemmArgs& g, hipStream_t s) {
    if (g.M <= 0 || g.N <= 0) return;
    if (gemm_f16s_config(g) == 0) launch_f16s_cfg<128, 128, 2, 2>(g, s);
    else launch_f16s_cfg<64, 64, 2, 2>(g, s);
}

}  // namespace artalk
