// Small kernels between the GEMMs of the AR / VAE / style stages: bit decisions, the multi-scale
// binary-spherical-quantiser arithmetic and the embeddings around it.  All are tiny (<= 3200 floats per clip)
// and exist so that the whole scale step stays on the device (hipGraph-capturable, no host round trip).
//
//   ar_bits        app/models.py:104 (pairwise argmax) + bitwise_vae.py:291-305 (f_hat recurrence, area pooling)
//   vq_embed       app/models.py:19,100,106-107,113 (vqfeat_embed, style token, + level/position embedding)
//   dec_input      bitwise_vae.py:105-110,280-288
//   dec_finish     bitwise_vae.py:63-65,111-113 ; :59-61,87 for the re-encode input
//   bsq_history    bitwise_vae.py:227-242,316-334 (MultiScaleBSQ/BSQ forward) + :269-288 (features from bits)
//   style_*        app/modules/style_encoder.py:26-38,58-60 ; app/models.py:67-73
//   savgol         inference.py:89-95 (scipy.signal.savgol_filter, mode='interp')
#include "common.h"
#include <cmath>
#include <mutex>
#include <set>

namespace artalk {

constexpr int T100 = 100;   // frames per chunk = finest scale
constexpr int CD = 32;      // code dim
constexpr int NLV = 5;
__constant__ int c_pn[NLV];            // 1,5,25,50,100
__constant__ int c_off[NLV + 1];       // 0,1,6,31,81,181
__constant__ int c_up_i0[4][T100];     // linear upsample pn[p] -> 100 (align_corners=False)
__constant__ int c_up_i1[4][T100];
__constant__ float c_up_w1[4][T100];
__constant__ float c_hq;               // 1/sqrt(32)

static const int h_pn[NLV] = {1, 5, 25, 50, 100};

// __constant__ symbols exist once per DEVICE: the tables are uploaded the first time a model is created on each GPU
// (artalk_create calls this after hipSetDevice).  Returns 0, or -1 with nothing recorded if an upload failed.
int init_ms_tables() {
    static std::mutex mu;
    static std::set<int> ready;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    std::lock_guard<std::mutex> lock(mu);
    if (ready.count(dev)) return 0;
    int off[NLV + 1] = {0};
    for (int i = 0; i < NLV; ++i) off[i + 1] = off[i] + h_pn[i];
    int i0[4][T100], i1[4][T100];
    float w1[4][T100];
    for (int p = 0; p < 4; ++p) {
        const int in = h_pn[p];
        const float scale = (float)in / (float)T100;     // ATen area_pixel_compute_scale
        for (int t = 0; t < T100; ++t) {
            float src = scale * ((float)t + 0.5f) - 0.5f;   // area_pixel_compute_source_index
            if (src < 0.f) src = 0.f;
            const int a = (int)src;
            i0[p][t] = a;
            i1[p][t] = a + ((a < in - 1) ? 1 : 0);
            w1[p][t] = src - (float)a;
        }
    }
    const float hq = 1.0f / (float)std::sqrt(32.0);
    bool ok = hipMemcpyToSymbol(HIP_SYMBOL(c_pn), h_pn, sizeof(h_pn)) == hipSuccess;
    ok = ok && hipMemcpyToSymbol(HIP_SYMBOL(c_off), off, sizeof(off)) == hipSuccess;
    ok = ok && hipMemcpyToSymbol(HIP_SYMBOL(c_up_i0), i0, sizeof(i0)) == hipSuccess;
    ok = ok && hipMemcpyToSymbol(HIP_SYMBOL(c_up_i1), i1, sizeof(i1)) == hipSuccess;
    ok = ok && hipMemcpyToSymbol(HIP_SYMBOL(c_up_w1), w1, sizeof(w1)) == hipSuccess;
    ok = ok && hipMemcpyToSymbol(HIP_SYMBOL(c_hq), &hq, sizeof(hq)) == hipSuccess;
    if (!ok) return -1;
    ready.insert(dev);
    return 0;
}

__device__ __forceinline__ float up_lin(const float* src /*[pn][32] in LDS*/, int p, int t, int c) {
    const float w1 = c_up_w1[p][t], w0 = 1.0f - w1;
    return w0 * src[c_up_i0[p][t] * CD + c] + w1 * src[c_up_i1[p][t] * CD + c];
}
// adaptive average pooling of f[100][32] to pn bins (100 % pn == 0 for pn in {1,5,25,50,100})
__device__ __forceinline__ float area_pool(const float* f, int pn, int i, int c) {
    const int w = T100 / pn;
    float s = 0.f;
    for (int t = i * w; t < (i + 1) * w; ++t) s += f[t * CD + c];
    return s / (float)w;
}

// ------------------------------------------------------------------------------------------------
// Headroom audit (artalk_set_audit): max |x| over a P8 buffer (the hi halves are f16(2^e x): inv_scale = 2^-e) or over an fp32 buffer;
// positive floats order like their bit patterns, so the maximum is an atomicMax.
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ buf, int rows, int cols, long ld, int is_p8,
                                                     unsigned int* __restrict__ slot, int junk_period, int junk_from, float inv_scale) {
    float m = 0.f;
    const long groups = (long)rows * (cols / 8);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < groups; i += (long)gridDim.x * 256) {
        const long r = i / (cols / 8), g8 = i - r * (cols / 8);
        if (junk_period && r % junk_period >= junk_from) continue;      // layout padding rows (conv stack)
        const float* p = buf + r * ld + g8 * 8;
        if (is_p8) {
            const _Float16* h = reinterpret_cast<const _Float16*>(p);      // [8 x hi][8 x lo]
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = fabsf((float)h[e]) * inv_scale;
                m = (v != v) ? INFINITY : fmaxf(m, v);      // (a NaN reports as an overflow instead of vanishing in fmaxf)
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = fabsf(p[e]);
                m = (v != v) ? INFINITY : fmaxf(m, v);
            }
        }
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) atomicMax(slot, __float_as_uint(m));
}
void launch_absmax(const float* buf, int rows, int cols, long ld, int is_p8, unsigned int* slot, hipStream_t s, int junk_period, int junk_from, int p8_exp) {
    if (rows <= 0 || cols < 8) return;
    const long groups = (long)rows * (cols / 8);
    const long blocks = (groups + 255) / 256;
    ARTALK_LAUNCH(absmax_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, s, buf, rows, cols, ld, is_p8, slot, junk_period,
                       junk_from, p8_scale_of(-p8_exp));
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ar_begin_kernel(const float* __restrict__ style_cond, const float* __restrict__ lvlpos,
                                                       float* __restrict__ x0, float* __restrict__ fhat, int E) {
    const int b = blockIdx.x;
    for (int n = threadIdx.x; n < E; n += 256) x0[(long)b * E + n] = style_cond[(long)b * E + n] + lvlpos[n];
    for (int i = threadIdx.x; i < T100 * CD; i += 256) fhat[(long)b * T100 * CD + i] = 0.f;
}
void launch_ar_begin(const float* style_cond, const float* lvlpos, float* x0, float* fhat, int B, hipStream_t s) {
    ARTALK_LAUNCH(ar_begin_kernel, dim3(B), dim3(256), 0, s, style_cond, lvlpos, x0, fhat, 768);
}

// ------------------------------------------------------------------------------------------------
// level p: logits rows b*pn+i -> bits; f_hat += up(h_p); nextfeat[b, i', :] = area(f_hat -> pn[p+1])
constexpr int BITS_NT = 1024;      // passes over the clip's <= 3 200 values between barriers: 3 trips instead of 13
__global__ __launch_bounds__(BITS_NT) void ar_bits_kernel(const float* __restrict__ logits, uint8_t* __restrict__ bits,
                                                      float* __restrict__ fhat, float* __restrict__ nextfeat, int p, int* __restrict__ status) {
    __shared__ float hs[T100 * CD];
    __shared__ float fs[T100 * CD];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int pn = c_pn[p], off = c_off[p];
    for (int idx = tid; idx < pn * CD; idx += BITS_NT) {
        const int i = idx / CD, c = idx % CD;
        const float* l = logits + ((long)b * pn + i) * (2 * CD) + 2 * c;
        const int bit = l[1] > l[0];                       // argmax over the pair, ties -> 0
        // a NaN/Inf logit would be laundered into a 0 bit by the comparison: report it (fp16 overflow in f16x3 mode, bad weights)
        if (status && !(isfinite(l[0]) && isfinite(l[1]))) atomicOr(status, 1);
        bits[((long)b * 181 + off + i) * CD + c] = (uint8_t)bit;
        hs[idx] = bit ? c_hq : -c_hq;
    }
    if (p >= NLV - 1) return;
    __syncthreads();
    float* fg = fhat + (long)b * T100 * CD;
    for (int idx = tid; idx < T100 * CD; idx += BITS_NT) {
        const int t = idx / CD, c = idx % CD;
        const float v = fg[idx] + up_lin(hs, p, t, c);
        fg[idx] = v;
        fs[idx] = v;
    }
    __syncthreads();
    const int pn2 = c_pn[p + 1];
    for (int idx = tid; idx < pn2 * CD; idx += BITS_NT)
        nextfeat[(long)b * pn2 * CD + idx] = area_pool(fs, pn2, idx / CD, idx % CD);
}
void launch_ar_bits_next(const float* logits, uint8_t* bits, float* fhat, float* nextfeat, int B, int level, hipStream_t s, int* status) {
    ARTALK_LAUNCH(ar_bits_kernel, dim3(B), dim3(BITS_NT), 0, s, logits, bits, fhat, nextfeat, level, status);
}

// ------------------------------------------------------------------------------------------------
// X[b*xrows + xoff + i, :] = We * feat[b, i, :] + be + pos[i, :]      for i < n          (grid.x = i)
// X[b*xrows + 0, :]        = style_cond[b, :] + pos0[:]                for blockIdx.x == n (only if style_cond)
// A workgroup embeds VQ_TOK tokens of one clip: a thread keeps one weight row (32 floats) in registers and applies it to all the
// tokens (features in LDS), so the 98 KB weight matrix is read once per 16 tokens instead of once per token (69 -> ~10 us for
// the 180 history tokens of 16 clips).  The dot product runs over c in the same order as before: results are bit-identical.
constexpr int VQ_TOK = 16;
__global__ __launch_bounds__(256) void vq_embed_kernel(const float* __restrict__ feat, int n, const float* __restrict__ We,
                                                       const float* __restrict__ be, const float* __restrict__ pos,
                                                       float* __restrict__ X, int xrows, int xoff,
                                                       const float* __restrict__ style_cond, const float* __restrict__ pos0, int E) {
    __shared__ float f[VQ_TOK][CD];
    const int i0 = blockIdx.x * VQ_TOK, b = blockIdx.y, tid = threadIdx.x;
    if (blockIdx.x == 0 && style_cond) {
        for (int e = tid; e < E; e += 256) X[((long)b * xrows) * E + e] = style_cond[(long)b * E + e] + pos0[e];
    }
    const int nt = min(VQ_TOK, n - i0);
    for (int idx = tid; idx < VQ_TOK * CD; idx += 256) f[idx / CD][idx % CD] = idx < nt * CD ? feat[((long)b * n + i0) * CD + idx] : 0.f;
    // every global load of this thread - three weight rows, their biases and the position rows of all 16 tokens - is issued before
    // the first use: the position loads used to sit inside the token loop, twelve exposed latencies per workgroup (22 us per launch)
    constexpr int EPT = 3;                       // E = 768 = 3 x 256 threads (guarded for a smaller E)
    float w[EPT][CD], bias[EPT], pv[EPT][VQ_TOK];
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int e = tid + k * 256;
        const bool ok = e < E;
#pragma unroll
        for (int c4 = 0; c4 < CD / 4; ++c4) {
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            if (ok) t = *reinterpret_cast<const f32x4*>(We + (long)e * CD + 4 * c4);
            w[k][4 * c4] = t[0]; w[k][4 * c4 + 1] = t[1]; w[k][4 * c4 + 2] = t[2]; w[k][4 * c4 + 3] = t[3];
        }
        bias[k] = ok ? be[e] : 0.f;
#pragma unroll
        for (int t = 0; t < VQ_TOK; ++t) pv[k][t] = (ok && t < nt) ? pos[(long)(i0 + t) * E + e] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int e = tid + k * 256;
        if (e >= E) break;
        // four tokens at a time: four independent 32-long fma chains in flight (each token's own order is unchanged: bit-identical)
#pragma unroll
        for (int t0 = 0; t0 < VQ_TOK; t0 += 4) {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < CD; ++c)
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[u] = fmaf(w[k][c], f[t0 + u][c], acc[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (t0 + u < nt) X[((long)b * xrows + xoff + i0 + t0 + u) * E + e] = (acc[u] + bias[k]) + pv[k][t0 + u];
        }
    }
}
void launch_vq_embed(const float* feat, int n, const float* We, const float* be, const float* pos, float* X, int xrows,
                     int xoff, const float* style_cond, const float* pos0, int B, hipStream_t s) {
    ARTALK_LAUNCH(vq_embed_kernel, dim3((n + VQ_TOK - 1) / VQ_TOK, B), dim3(256), 0, s, feat, n, We, be, pos, X, xrows, xoff,
                       style_cond, pos0, 768);
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dec_input_kernel(const float* __restrict__ prev_fdec, const float* __restrict__ fhat,
                                                        const uint8_t* __restrict__ bits, const float* __restrict__ dpos,
                                                        float* __restrict__ X) {
    const int b = blockIdx.x;     // grid.y slices of the clip's 6 400 elements: one block per clip was a 25-iteration latency chain
    for (int idx = blockIdx.y * 256 + threadIdx.x; idx < 2 * T100 * CD; idx += gridDim.y * 256) {
        const int t = idx / CD, c = idx % CD;
        float v;
        if (t < T100) {
            v = prev_fdec[(long)b * T100 * CD + idx];
        } else {
            const int tt = t - T100;
            const float h = bits[((long)b * 181 + 81 + tt) * CD + c] ? c_hq : -c_hq;
            v = fhat[(long)b * T100 * CD + tt * CD + c] + h;
        }
        X[(long)b * 2 * T100 * CD + idx] = v + dpos[idx];
    }
}
void launch_dec_input(const float* prev_fdec, const float* fhat, const uint8_t* bits, const float* dpos, float* X, int B,
                      hipStream_t s) {
    ARTALK_LAUNCH(dec_input_kernel, dim3(B, 8), dim3(256), 0, s, prev_fdec, fhat, bits, dpos, X);
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dec_finish_kernel(const float* __restrict__ dec, const float* __restrict__ mean,
                                                         const float* __restrict__ stdv, const float* __restrict__ epos,
                                                         float* __restrict__ out, long out_bstride, int chunk,
                                                         float* __restrict__ E, int MD, int EK, int* __restrict__ status) {
    const int b = blockIdx.x;
    for (int idx = blockIdx.y * 256 + threadIdx.x; idx < T100 * EK; idx += gridDim.y * 256) {
        const int t = idx / EK, j = idx % EK;
        float e = 0.f;
        if (j < MD) {
            const float d = dec[((long)b * 2 * T100 + T100 + t) * MD + j];
            const float m = d * stdv[j] + mean[j];
            if (status && !isfinite(m)) atomicOr(status, 4);
            out[(long)b * out_bstride + ((long)chunk * T100 + t) * MD + j] = m;
            e = (m - mean[j]) / stdv[j] + epos[t * MD + j];
        }
        E[((long)b * T100 + t) * EK + j] = e;
    }
}
void launch_dec_finish(const float* dec, const float* mean, const float* stdv, const float* epos, float* out,
                       long out_bstride, int chunk, float* E, int B, hipStream_t s, int* status) {
    ARTALK_LAUNCH(dec_finish_kernel, dim3(B, 10), dim3(256), 0, s, dec, mean, stdv, epos, out, out_bstride, chunk, E, 106, 128, status);
}

__global__ __launch_bounds__(256) void enc_input_zero_kernel(const float* __restrict__ mean, const float* __restrict__ stdv,
                                                             const float* __restrict__ epos, float* __restrict__ E, int MD, int EK) {
    const int b = blockIdx.x;
    for (int idx = threadIdx.x; idx < T100 * EK; idx += 256) {
        const int t = idx / EK, j = idx % EK;
        E[((long)b * T100 + t) * EK + j] = (j < MD) ? ((0.f - mean[j]) / stdv[j] + epos[t * MD + j]) : 0.f;
    }
}
// dst[b][0 .. bytes) = src[0 .. bytes) for b < B (bytes % 16 == 0): the initial history of a clip is the same for every clip
__global__ __launch_bounds__(256) void broadcast16_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int n16) {
    uint4* d = dst + (long)blockIdx.y * n16;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) d[i] = src[i];
}
void launch_broadcast16(const void* src, void* dst, long bytes, int B, hipStream_t s) {
    if (B <= 0 || bytes <= 0) return;
    const int n16 = (int)(bytes / 16);
    ARTALK_LAUNCH(broadcast16_kernel, dim3((n16 + 255) / 256 < 8 ? (n16 + 255) / 256 : 8, B), dim3(256), 0, s,
                       reinterpret_cast<const uint4*>(src), reinterpret_cast<uint4*>(dst), n16);
}

void launch_enc_input_zero(const float* mean, const float* stdv, const float* epos, float* E, int B, hipStream_t s) {
    ARTALK_LAUNCH(enc_input_zero_kernel, dim3(B), dim3(256), 0, s, mean, stdv, epos, E, 106, 128);
}

// ------------------------------------------------------------------------------------------------
// One workgroup per clip.  Sequential over the 5 scales (each needs the residual left by the previous one); 1024 threads: the kernel is a
// chain of passes over the clip's 3 200 values between barriers, 3 trips per pass instead of 13 (41 -> ~20 us per launch).
constexpr int BSQ_NT = 1024;
__global__ __launch_bounds__(BSQ_NT) void bsq_history_kernel(const float* __restrict__ enc_out, uint8_t* __restrict__ hist_bits,
                                                          float* __restrict__ prev_fdec, float* __restrict__ msfeat, int* __restrict__ status) {
    __shared__ float resid[T100 * CD];
    __shared__ float fms[T100 * CD];     // sum of up(h_p), the feature recurrence from bits
    __shared__ float qs[T100 * CD];      // quantised values q = z + (zhat - z) of this scale
    __shared__ float hs[T100 * CD];      // +-1/sqrt(32) from the bits of this scale
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int idx = tid; idx < T100 * CD; idx += BSQ_NT) {
        resid[idx] = enc_out[(long)b * T100 * CD + idx];
        if (status && !isfinite(resid[idx])) atomicOr(status, 2);
        fms[idx] = 0.f;
    }
    __syncthreads();
    int msoff = 0;
    for (int p = 0; p < NLV; ++p) {
        const int pn = c_pn[p], off = c_off[p];
        // all threads run the loop body the same number of times (guard inside, shuffles outside)
        for (int base = 0; base < pn * CD; base += BSQ_NT) {
            const int idx = base + tid;
            const bool ok = idx < pn * CD;
            const int i = ok ? idx / CD : 0, c = idx % CD;
            float x = 0.f;
            if (ok) x = (pn == T100) ? resid[idx] : area_pool(resid, pn, i, c);
            float ss = x * x;                      // 32 consecutive lanes = one token
            ss = group_sum<32>(ss);
            const float z = x / fmaxf(sqrtf(ss), 1e-12f);
            const float zhat = (z > 0.f) ? c_hq : -c_hq;
            const float q = z + (zhat - z);
            const int bit = q > 0.f;
            if (ok) {
                qs[idx] = q;
                hs[idx] = bit ? c_hq : -c_hq;
                hist_bits[((long)b * 181 + off + i) * CD + c] = (uint8_t)bit;
            }
        }
        __syncthreads();
        if (p < NLV - 1) {
            for (int idx = tid; idx < T100 * CD; idx += BSQ_NT) {
                const int t = idx / CD, c = idx % CD;
                resid[idx] -= up_lin(qs, p, t, c);
                fms[idx] += up_lin(hs, p, t, c);
            }
            __syncthreads();
            const int pn2 = c_pn[p + 1];
            for (int idx = tid; idx < pn2 * CD; idx += BSQ_NT)
                msfeat[((long)b * 180 + msoff) * CD + idx] = area_pool(fms, pn2, idx / CD, idx % CD);
            msoff += pn2;
        } else {
            for (int idx = tid; idx < T100 * CD; idx += BSQ_NT) prev_fdec[(long)b * T100 * CD + idx] = fms[idx] + hs[idx];
        }
        __syncthreads();
    }
}
void launch_bsq_history(const float* enc_out, uint8_t* hist_bits, float* prev_fdec, float* msfeat, int B, hipStream_t s, int* status) {
    ARTALK_LAUNCH(bsq_history_kernel, dim3(B), dim3(BSQ_NT), 0, s, enc_out, hist_bits, prev_fdec, msfeat, status);
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void style_input_kernel(const float* __restrict__ motion, const float* __restrict__ mean,
                                                          const float* __restrict__ stdv, float* __restrict__ X, int rows, int MD, int EK) {
    const long n = (long)rows * EK;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
        const long row = idx / EK; const int j = idx % EK;
        X[idx] = (j < MD) ? (motion[row * MD + j] - mean[j]) / stdv[j] : 0.f;
    }
}
void launch_style_input(const float* motion, const float* mean, const float* stdv, float* X, int B, hipStream_t s) {
    const int rows = B * 50;
    ARTALK_LAUNCH(style_input_kernel, dim3((rows * 128 + 255) / 256), dim3(256), 0, s, motion, mean, stdv, X, rows, 106, 128);
}

__global__ __launch_bounds__(256) void add_row_kernel(float* __restrict__ X, const float* __restrict__ v, long n, int D) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) X[idx] += v[idx % D];
}
void launch_add_row(float* X, const float* v, int M, int D, hipStream_t s) {
    const long n = (long)M * D;
    ARTALK_LAUNCH(add_row_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, X, v, n, D);
}

__global__ __launch_bounds__(256) void style_finish_kernel(const float* __restrict__ feat, const float* __restrict__ Ws,
                                                           const float* __restrict__ bs, const float* __restrict__ null_cond,
                                                           const uint8_t* __restrict__ has_style, float* __restrict__ style_cond,
                                                           int L, int S, int E, const float* __restrict__ cached, long cached_stride) {
    __shared__ float m[128];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int has = has_style ? has_style[b] : 0;
    if (has != 1 && has != 2) {      // 0 = no style; the host rejects other values (artalk_infer), anything else would read stale rows
        for (int e = tid; e < E; e += 256) style_cond[(long)b * E + e] = null_cond[e];
        return;
    }
    if (has == 2) {     // row b of the style input carries a condition computed earlier by artalk_style_encode
        for (int e = tid; e < E; e += 256) style_cond[(long)b * E + e] = cached[(long)b * cached_stride + e];
        return;
    }
    if (tid < S) {
        float s = 0.f;
        for (int t = 0; t < L; ++t) s += feat[((long)b * L + t) * S + tid];
        m[tid] = s / (float)L;
    }
    __syncthreads();
    for (int e = tid; e < E; e += 256) {
        const float* w = Ws + (long)e * S;
        float acc = 0.f;
        for (int c = 0; c < S; ++c) acc = fmaf(w[c], m[c], acc);
        style_cond[(long)b * E + e] = (acc + bs[e]) * 1.1f - null_cond[e] * 0.1f;
    }
}
void launch_style_finish(const float* feat, const float* Ws, const float* bs, const float* null_cond,
                         const uint8_t* has_style, float* style_cond, int B, hipStream_t s, const float* cached, long cached_stride) {
    ARTALK_LAUNCH(style_finish_kernel, dim3(B), dim3(256), 0, s, feat, Ws, bs, null_cond, has_style, style_cond, 50, 128, 768,
                       cached, cached_stride);
}

// ------------------------------------------------------------------------------------------------
// Savitzky-Golay (5,2) on all dims, (9,3) on dims 100:103 of the unsmoothed signal; scipy mode='interp':
// FIR in the interior, least-squares polynomial of the first/last window evaluated at the edge samples.
// Coefficients are computed on the host in double; the kernel accumulates in double and rounds once,
// like scipy (float64 coefficients, float32 data).
struct SavgolCoef {
    double fir5[5], edge5[2][5];     // edge rows: output i (0,1) from first 5 samples; mirrored for the tail
    double fir9[9], edge9[4][9];
};

static void polyfit_edge(int window, int order, int n_edge, double* out /*[n_edge][window]*/, double* fir /*[window]*/) {
    // Least squares: y ~ sum_k a_k x^k over x = 0..window-1.  H = A (A^T A)^-1 A^T; row i of H evaluates the fit at x=i.
    const int m = order + 1;
    double ata[16] = {0}, inv[16];
    for (int r = 0; r < m; ++r)
        for (int c = 0; c < m; ++c) {
            double s = 0;
            for (int x = 0; x < window; ++x) s += std::pow((double)x, r + c);
            ata[r * m + c] = s;
        }
    // Gauss-Jordan inverse (m <= 4)
    double aug[4][8];
    for (int r = 0; r < m; ++r)
        for (int c = 0; c < 2 * m; ++c) aug[r][c] = c < m ? ata[r * m + c] : (c - m == r ? 1.0 : 0.0);
    for (int col = 0; col < m; ++col) {
        int piv = col;
        for (int r = col + 1; r < m; ++r) if (std::fabs(aug[r][col]) > std::fabs(aug[piv][col])) piv = r;
        for (int c = 0; c < 2 * m; ++c) std::swap(aug[col][c], aug[piv][c]);
        const double d = aug[col][col];
        for (int c = 0; c < 2 * m; ++c) aug[col][c] /= d;
        for (int r = 0; r < m; ++r)
            if (r != col) { const double f = aug[r][col]; for (int c = 0; c < 2 * m; ++c) aug[r][c] -= f * aug[col][c]; }
    }
    for (int r = 0; r < m; ++r) for (int c = 0; c < m; ++c) inv[r * m + c] = aug[r][m + c];
    auto H = [&](int i, int j) {
        double s = 0;
        for (int r = 0; r < m; ++r) for (int c = 0; c < m; ++c) s += std::pow((double)i, r) * inv[r * m + c] * std::pow((double)j, c);
        return s;
    };
    for (int i = 0; i < n_edge; ++i) for (int j = 0; j < window; ++j) out[i * window + j] = H(i, j);
    for (int j = 0; j < window; ++j) fir[j] = H(window / 2, j);
}

__global__ __launch_bounds__(128) void savgol_kernel(const float* __restrict__ in, float* __restrict__ out, int T, int D, SavgolCoef k) {
    const int t = blockIdx.x, d = threadIdx.x;
    if (d >= D) return;
    double acc = 0.0;
    if (d >= 100 && d < 103) {
        if (t < 4) { for (int j = 0; j < 9; ++j) acc += k.edge9[t][j] * (double)in[(long)j * D + d]; }
        else if (t >= T - 4) { const int i = T - 1 - t; for (int j = 0; j < 9; ++j) acc += k.edge9[i][j] * (double)in[(long)(T - 1 - j) * D + d]; }
        else { for (int j = 0; j < 9; ++j) acc += k.fir9[j] * (double)in[(long)(t - 4 + j) * D + d]; }
    } else {
        if (t < 2) { for (int j = 0; j < 5; ++j) acc += k.edge5[t][j] * (double)in[(long)j * D + d]; }
        else if (t >= T - 2) { const int i = T - 1 - t; for (int j = 0; j < 5; ++j) acc += k.edge5[i][j] * (double)in[(long)(T - 1 - j) * D + d]; }
        else { for (int j = 0; j < 5; ++j) acc += k.fir5[j] * (double)in[(long)(t - 2 + j) * D + d]; }
    }
    out[(long)t * D + d] = (float)acc;
}
void launch_savgol(const float* in, float* out, int T, int D, hipStream_t s) {
    static SavgolCoef k;
    static std::once_flag once;
    std::call_once(once, [] {
        polyfit_edge(5, 2, 2, &k.edge5[0][0], k.fir5);
        polyfit_edge(9, 3, 4, &k.edge9[0][0], k.fir9);
    });
    if (T < 9) abort();   // the Python host raises ValueError first (scipy does the same for mode='interp')
    ARTALK_LAUNCH(savgol_kernel, dim3(T), dim3(128), 0, s, in, out, T, D, k);
}

}  // namespace artalk
