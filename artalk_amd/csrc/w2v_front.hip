// wav2vec2 front-end kernels (HBM-bound byte movers) and the multi-scale audio pooling.
//   audio_normalize  app/modules/wav2vec.py:22-27   per-chunk (x-mean)/(std_unbiased+1e-6)
//   conv0            hf:275-299 layer 0             Conv1d(1->512,k=10,s=5)+bias -> LN(512) -> GELU(erf)
//   pool_silu        app/models.py:94-95            area pooling 199 -> {1,5,25,50,100}; SiLU of app/transformer.py:25
#include "common.h"

namespace artalk {

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void audio_normalize_kernel(const float* __restrict__ audio, const long* __restrict__ src_off,
                                                               float* __restrict__ xn, int n) {
    __shared__ double red[16];
    __shared__ float stat[2];
    const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* x = audio + src_off[c];
    double s = 0.0;
    for (int i = tid; i < n; i += 1024) s += (double)x[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) red[wv] = s;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int i = 0; i < 16; ++i) t += red[i];
        stat[0] = (float)(t / n);
    }
    __syncthreads();
    const double mean = (double)stat[0];
    double q = 0.0;
    for (int i = tid; i < n; i += 1024) { const double d = (double)x[i] - mean; q += d * d; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    __syncthreads();
    if (lane == 0) red[wv] = q;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int i = 0; i < 16; ++i) t += red[i];
        stat[1] = (float)sqrt(t / (n - 1));
    }
    __syncthreads();
    const float m = stat[0], den = stat[1] + 1e-6f;
    float* y = xn + (long)c * n;
    for (int i = tid; i < n; i += 1024) y[i] = (x[i] - m) / den;
}

void launch_audio_normalize(const float* audio, const long* src_off, float* xnorm, int n_chunks, int n, hipStream_t s) {
    if (n_chunks <= 0) return;
    ARTALK_LAUNCH(audio_normalize_kernel, dim3(n_chunks), dim3(1024), 0, s, audio, src_off, xnorm, n);
}

// ------------------------------------------------------------------------------------------------
// One wavefront per output frame: lane l owns channels 4l..4l+3 and 256+4l..256+4l+3 (two 1-KiB
// coalesced stores per row); the 80 filter taps live in registers; LN statistics by wave butterflies.
// 20 bytes of audio in, 2 KiB out per frame: store-bandwidth bound.
__global__ __launch_bounds__(256) void conv0_kernel(const float* __restrict__ xn, int n, const float* __restrict__ w,
                                                    const float* __restrict__ bias, const float* __restrict__ lnw,
                                                    const float* __restrict__ lnb, float* __restrict__ Y, int T, int row_stride, int* __restrict__ status, int out_p8, int p8_exp) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = blockIdx.y;
    // channel pairs (2 jj, 2 jj + 1) of this lane in packed-fp32 registers: the 80 taps, the LayerNorm and the GELU polynomial run on
    // v_pk_fma_f32 (two channels per instruction).  The kernel was bound by this arithmetic, not by its 2.5 GB of stores.
    f32x2 wt[4][10], bs[4], gw[4], gb[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int c0 = (jj >> 1) * 256 + lane * 4 + (jj & 1) * 2;
#pragma unroll
        for (int k = 0; k < 10; ++k) { wt[jj][k].x = w[c0 * 10 + k]; wt[jj][k].y = w[(c0 + 1) * 10 + k]; }
        bs[jj].x = bias[c0]; bs[jj].y = bias[c0 + 1];
        gw[jj].x = lnw[c0]; gw[jj].y = lnw[c0 + 1];
        gb[jj].x = lnb[c0]; gb[jj].y = lnb[c0 + 1];
    }
    const float* x = xn + (long)c * n;
    for (int t = blockIdx.x * 4 + wv; t < T; t += gridDim.x * 4) {
        const int idx = min(5 * t + (lane & 15), n - 1);
        const float xv = x[idx];
        float xs[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) xs[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xv), k));   // (v_readlane: a scalar, no LDS crossbar)
        f32x2 v[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            f32x2 acc = bs[jj];
#pragma unroll
            for (int k = 0; k < 10; ++k) acc = __builtin_elementwise_fma(wt[jj][k], (f32x2)(xs[k]), acc);      // same order per channel as the scalar form
            v[jj] = acc;
        }
        // (the sums keep the scalar kernel's order: channels 0..7 of the lane one after the other, then the wave butterfly)
        float s = 0.f;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) { s += v[jj].x; s += v[jj].y; }
        const float mean = wave_sum(s) * (1.0f / 512.0f);
        float q = 0.f;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const f32x2 d = v[jj] - mean;
            q = fmaf(d.x, d.x, q); q = fmaf(d.y, d.y, q);
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / 512.0f) + 1e-5f);
        f32x2 o[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) o[jj] = gelu_erf2(__builtin_elementwise_fma((v[jj] - mean) * rstd, gw[jj], gb[jj]));
        float* y = Y + ((long)c * row_stride + t) * 512;
        if (out_p8) {
            // adjacent lanes own the two halves of one 8-element P8 group: they trade halves and issue ONE 16-byte store each
            // (2 store instructions per row and lane instead of 4)
            store_p8x4_pair(y, lane * 4, o[0].x, o[0].y, o[1].x, o[1].y, status, p8_exp);
            store_p8x4_pair(y, 256 + lane * 4, o[2].x, o[2].y, o[3].x, o[3].y, status, p8_exp);
        } else {
            f32x4 lo = {o[0].x, o[0].y, o[1].x, o[1].y}, hi = {o[2].x, o[2].y, o[3].x, o[3].y};
            *reinterpret_cast<f32x4*>(y + lane * 4) = lo;
            *reinterpret_cast<f32x4*>(y + 256 + lane * 4) = hi;
        }
    }
}

void launch_conv0(const float* xnorm, int n, const float* w, const float* bias, const float* lnw, const float* lnb,
                  float* Y, int n_chunks, int T, int row_stride, hipStream_t s, int out_p8, int* status, int p8_exp) {
    if (n_chunks <= 0) return;
    ARTALK_LAUNCH(conv0_kernel, dim3(128, n_chunks), dim3(256), 0, s, xnorm, n, w, bias, lnw, lnb, Y, T, row_stride, status, out_p8, p8_exp);
}

// ------------------------------------------------------------------------------------------------
struct PoolLevels { int n; int pn[8]; };

__global__ __launch_bounds__(256) void pool_silu_kernel(const float* __restrict__ X, int x_tstride, int T, float* __restrict__ Y,
                                                        PoolLevels lv, int ntok, int D, int out_p8, int* __restrict__ status, int p8_exp) {
    const int tok = blockIdx.x, c = blockIdx.y;
    int p = 0, i = tok;
    while (i >= lv.pn[p]) { i -= lv.pn[p]; ++p; }
    const int pn = lv.pn[p];
    const int t0 = (i * T) / pn, t1 = ((i + 1) * T + pn - 1) / pn;   // adaptive_avg_pool1d bins
    const float inv = (float)(t1 - t0);
    for (int d = threadIdx.x * 4; d < D; d += 256 * 4) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int t = t0; t < t1; ++t) s += *reinterpret_cast<const f32x4*>(X + ((long)c * x_tstride + t) * D + d);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = silu(s[e] / inv);
        float* yrow = Y + ((long)c * ntok + tok) * D;
        if (out_p8) store_p8x4(yrow, d, o[0], o[1], o[2], o[3], status, p8_exp);
        else *reinterpret_cast<f32x4*>(yrow + d) = o;
    }
}

void launch_pool_silu(const float* X, int x_tstride, int T, float* Y, int n_chunks, const int* patch_nums, int n_lvls, int D,
                      hipStream_t s, int out_p8, int* status, int p8_exp) {
    if (n_chunks <= 0) return;
    PoolLevels lv;
    lv.n = n_lvls;
    int ntok = 0;
    for (int i = 0; i < 8; ++i) { lv.pn[i] = i < n_lvls ? patch_nums[i] : 1 << 30; if (i < n_lvls) ntok += patch_nums[i]; }
    ARTALK_LAUNCH(pool_silu_kernel, dim3(ntok, n_chunks), dim3(256), 0, s, X, x_tstride, T, Y, lv, ntok, D, out_p8, status, p8_exp);
}

}  // namespace artalk

// ------------------------------------------------------------------------------------------------
// torchaudio-style polyphase sinc resampler + channel mean (reference inference.py:230-231): output frame i = j*new + p
// is sum_k taps[p][k] * xpad[j*orig + k] with xpad = x zero-padded by `width` on the left; mean over channels afterwards.
// HBM-bound FIR (48k stereo -> 16k mono: 24 B in, 4 B out per output sample, 41 taps served from L1/L2).
namespace artalk {
__global__ __launch_bounds__(256) void resample_mean_kernel(const float* __restrict__ x, int nch, int n, const float* __restrict__ taps,
                                                            int orig, int nw, int width, float* __restrict__ out, int n_out) {
    const int ntaps = 2 * width + orig;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n_out; i += gridDim.x * 256) {
        const int j = i / nw, p = i - j * nw;
        const float* tp = taps + (long)p * ntaps;
        const long base = (long)j * orig - width;
        float acc = 0.f;
        for (int c = 0; c < nch; ++c) {
            const float* xc = x + (long)c * n;
            float s = 0.f;
            for (int k = 0; k < ntaps; ++k) {
                const long t = base + k;
                if (t >= 0 && t < n) s = fmaf(tp[k], xc[t], s);
            }
            acc += s;
        }
        out[i] = acc / (float)nch;
    }
}
void launch_resample_mean(const float* x, int nch, int n, const float* taps, int orig, int nw, int width, float* out, int n_out,
                          hipStream_t s) {
    if (n_out <= 0) return;
    const int blocks = (n_out + 255) / 256;
    ARTALK_LAUNCH(resample_mean_kernel, dim3(blocks < 4096 ? blocks : 4096), dim3(256), 0, s, x, nch, n, taps, orig, nw, width, out, n_out);
}
}  // namespace artalk
