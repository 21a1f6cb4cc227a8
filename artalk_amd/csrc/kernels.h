// Host-side launch API of the gfx950 kernels of the ARTalk audio->motion path.
// Every launcher enqueues on the given stream and returns immediately (no sync, no allocation),
// so a sequence of them can be captured into a hipGraph.
#pragma once
#include <hip/hip_runtime.h>
#include <climits>
#include <cstdint>

#include <hip/hip_ext.h>
#include <utility>
#include <vector>

namespace artalk {

// ---- per-kernel timing without a profiler (artalk_set_profiling level 3; bench.py `budget_ms`).  While a KernelTimer is active on the
// launching thread, every launch of the library goes through hipExtLaunchKernelGGL with a start / stop event pair: the events carry
// the dispatch's own begin / end timestamps (what rocprofv3 --kernel-trace reports), so their difference is the kernel's duration
// without the gap to its neighbours.  Durations are summed per bucket (a stage of the path, a scale step of the body).
struct KernelTimer {
    int bucket = 0;
    std::vector<hipEvent_t> pool; size_t used = 0;
    std::vector<std::pair<int, size_t>> recs;      // (bucket, index of the start event; the stop event is the next one)
};
extern thread_local KernelTimer* g_ktimer;          // engine.hip; null = plain launches
bool ktimer_events(hipEvent_t* e0, hipEvent_t* e1);  // next event pair of the active timer (false: none active)
#define ARTALK_LAUNCH(kernel, grid, block, lds, stream, ...)                                                          \
    do {                                                                                                              \
        hipEvent_t kt_e0_ = nullptr, kt_e1_ = nullptr;                                                                \
        if (artalk::g_ktimer && artalk::ktimer_events(&kt_e0_, &kt_e1_))                                              \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, kt_e0_, kt_e1_, 0, __VA_ARGS__);                  \
        else                                                                                                          \
            hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                        \
    } while (0)

// row(m) = (m / rpb) * bstride + off + (m % rpb).  Identity when rpb == INT_MAX.
// Lets one GEMM read/write compact [B*L] rows and per-clip strided storage (KV cache, AdaLN table).
struct RowMap {
    int rpb;
    int bstride;
    int off;
};
static inline RowMap rowmap_identity() { return RowMap{INT_MAX, 0, 0}; }
static inline RowMap rowmap(int rows_per_batch, int batch_stride_rows, int row_off) {
    return RowMap{rows_per_batch, batch_stride_rows, row_off};
}

enum Act { ACT_NONE = 0, ACT_GELU_ERF = 1, ACT_GELU_TANH = 2, ACT_LEAKY02 = 3 };
constexpr int kActExp = 4;      // default site exponent of the P8 split format: activations scaled by 2^4 (common.h)

// C[cmap(m), n] = R[cmap(m), n] + gate[gmap(m), n] * act(sum_k A[m,k] * W[n,k] + bias[n])
// A is [M,K] (row stride lda), W is [N,K] row-major (torch nn.Linear layout).  K % 32 == 0.
struct GemmArgs {
    const float* A = nullptr; long lda = 0;
    const float* W = nullptr; long ldw = 0;
    const unsigned int* Wp = nullptr;   // optional copy of W in the P8 split format (common.h), same ld: f16x3 split path
    int c_p8 = 0;                        // 1: write C in the P8 split format (the consumer is a split GEMM); every split-GEMM epilogue and the split-K reduce
    int a_packed = 0;                    // 1: A is already in the P8 split format (written so by its producer kernel), f16x3 path only
    // site exponents of the P8 format (common.h): A was written (a_packed) or is split while staging (fp32 A) with scale 2^a_exp; a P8
    // result (c_p8, c2) is written with 2^c_exp.  kActExp (16) unless the model's calibration lowered a site with outlier activations.
    int a_exp = 4, c_exp = 4;
    int exact = 0;                       // 1: decision-critical GEMM (logit / code heads), always on the fp32 MFMA path
    const float* bias = nullptr;
    float* C = nullptr; long ldc = 0; RowMap cmap = {INT_MAX, 0, 0};
    float* c2 = nullptr;                 // optional: the (fp32) result once more in the P8 split format, same pitch (engine gemm(): the small-grid kernel writes both, otherwise a split pass follows; needs ldc == N and an identity cmap)
    const float* gate = nullptr; long ldg = 0; RowMap gmap = {INT_MAX, 0, 0};
    const float* R = nullptr; long ldr = 0;     // residual, addressed with cmap; may alias C
    int M = 0, N = 0, K = 0;
    int act = ACT_NONE;
    // grid.z batching (element strides)
    int batch = 1; long sA = 0, sW = 0, sBias = 0, sC = 0, sR = 0;
    // amode 1: grouped positional conv window (wav2vec2 pos_conv_embed): row m = (c, t) with
    // t = m % pc_tstride, k = (tap, ci): A[m,k] = X[c*pc_tstride + t + tap - pc_pad, ci] if in [0,pc_T) else 0
    int amode = 0; int pc_T = 0, pc_tstride = 0, pc_pad = 0, pc_cin = 0;
    // split-K (small-M steps of the AR loop): grid.y = splitk workgroups share one output tile, each writes its raw partial
    // sums to partial[y][M][N]; launch_splitk_reduce adds them in a fixed order (deterministic) and applies the epilogue.
    int splitk = 1; float* partial = nullptr;
    // column groups (the persistent 128x128 LDS-DMA kernel only): columns [j*ngrp, (j+1)*ngrp) are an independent linear layer j that
    // shares A - weight rows at W + j*grpW, bias at bias + j*grpB, result columns at C + j*grpC (elements; ngrp % 128 == 0).
    // One launch then computes the same input through many layers' weights (the history K/V of all 12 AR blocks).
    int ngrp = 0; long grpW = 0, grpB = 0, grpC = 0;
    int res_lds = 0;     // persistent kernel: 16 KiB of LDS behind the two stages are there for the deferred residual tiles (launch_gemm_p8)
    int w_nt = 0;        // (tuning) small-grid kernel: weight pieces fetched with the non-temporal cache policy
    int cus = 0;         // compute units the launch may use (0 = the whole device): grid of the one-workgroup-per-CU persistent kernels when the model runs on a CU partition
    int force_cfg = -1;  // >= 0: tile configuration override (tuning/tests)
    int graph_tag = 0;   // 1 for launches inside the captured AR/VAE body (separate kernel symbol, same code)
    int* status = nullptr;   // model status word: bit 3 is raised when a value leaves the range of the P8 format (c_p8 results, fp32 A split while staging)
};
void launch_gemm(const GemmArgs& g, hipStream_t s);
void launch_splitk_reduce(const GemmArgs& g, hipStream_t s);   // epilogue pass of a split-K GEMM (g.splitk > 1)
struct LnArgs;
// split-K epilogue pass fused with the AdaLN-modulated LayerNorm that consumes the 768-wide result (AR residual stream)
bool splitk_reduce_ln_eligible(const GemmArgs& g, const LnArgs& ln);
void launch_splitk_reduce_ln(const GemmArgs& g, const LnArgs& ln, hipStream_t s);
int gemm_tile_count(const GemmArgs& g, bool f16s);             // output tiles of the configuration launch_gemm[_f16s] would pick
// fp32-accurate GEMM on the fp16 matrix cores by operand splitting (gemm_f16s.hip)
void launch_pack_split(const float* w, unsigned int* out, long n, bool is_weight, hipStream_t s, int* status = nullptr, int p8_exp = kActExp);
bool gemm_f16s_eligible(const GemmArgs& g);
int gemm_f16s_config(const GemmArgs& g);     // register-staged kernel: 0: 128x128, 1: 64x64
void launch_gemm_f16s(const GemmArgs& g, hipStream_t s);
bool gemm_p8_eligible(const GemmArgs& g);      // both operands in P8 and a large grid: the LDS-DMA kernels (gemm_p8_2wgp / _256)
void launch_gemm_p8(const GemmArgs& g, hipStream_t s);
void gemm_p8_prepare();      // one-time kernel attributes (call once per process before the first captured launch)
int gemm_p8_variant(const GemmArgs& g);       // 0: gemm_p8_2wgp_kernel (persistent 128x128, two workgroups per CU), 1: gemm_p8_256_kernel
bool gemm_p8_sm_eligible(const GemmArgs& g);   // both operands in P8, any grid (split-K capable): small-tile LDS-DMA kernel
void launch_gemm_p8_sm(const GemmArgs& g, hipStream_t s);
bool gemm_p8_pp_ok(const GemmArgs& g);      // operands within the 32-bit DMA offsets of the ping-pong kernel (launch_gemm_p8_sm cfg 30 / 31 / 33)
// wav2vec2 positional convolution (16 groups of 64 channels, 128 taps) with the chunk's input window resident in LDS (gemm_f16s.hip)
void launch_posconv_p8(const GemmArgs& g, int n_chunks, int T, int Ts, hipStream_t s);
int gemm_config(const GemmArgs& g);   // 4: 128x128 BK16 (dominant kernel), 2: 64x64, 1: 128x64, 3: 32x128; 0,5,6,7 tuning variants
// Average kernel time helper for benches: FLOPs of one launch
static inline double gemm_flops(const GemmArgs& g) { return 2.0 * g.M * (double)g.N * g.K * g.batch; }

// LayerNorm over the last dim D of M rows, one wavefront per row.
//   y = LN(x) [* w + b]  [-> * (1 + scale[mmap(m)]) + shift[mmap(m)]]  [-> act]
struct LnArgs {
    const float* X = nullptr; long ldx = 0;
    float* Y = nullptr; long ldy = 0;
    const float* w = nullptr; const float* b = nullptr;          // affine (nullable)
    const float* scale = nullptr; const float* shift = nullptr;  // AdaLN modulation (nullable), row stride ldm
    long ldm = 0; RowMap mmap = {INT_MAX, 0, 0};
    int M = 0, D = 0; float eps = 1e-5f; int act = ACT_NONE;
    int out_p8 = 0;   // 1: write Y in the P8 split format (consumer is a split GEMM); same row pitch as fp32
    int p8_exp = 4;   // out_p8: site exponent (scale 2^p8_exp, common.h)
    int* status = nullptr;   // out_p8: range guard (see GemmArgs::status)
    // rows r with r % junk_period >= junk_from are layout padding (the per-chunk row stride of the wav2vec2 buffers exceeds the
    // valid frames; their input is whatever an earlier launch left there and no valid row ever reads them): they are stored as
    // zeros, which keeps everything computed from them finite and inside the P8 range, and are exempt from the range guard.
    int junk_period = 0, junk_from = 0;
};
void launch_layernorm(const LnArgs& a, hipStream_t s);
// softmax(scale * Q K^T [+mask]) V, fp32 MFMA, online softmax over 64-key blocks staged in LDS.
struct AttnArgs {
    const float* Q = nullptr; long ldq = 0, q_bstride = 0;
    const float* K = nullptr; long ldk = 0, k_bstride = 0;
    const float* V = nullptr; long ldv = 0, v_bstride = 0;
    float* O = nullptr; long ldo = 0, o_bstride = 0;
    int B = 0, H = 0, HD = 64, Lq = 0, Lk = 0;
    float scale = 1.f;
    int l2norm = 0; const float* qscale = nullptr;   // AR: q,k L2-normalised, q *= qscale[h]
    int split_q = 0, split_k = 0;                    // queries < split_q see keys < split_k only (VAE mask)
    int out_p8 = 0;                                  // 1: write O in the P8 split format
    int split16 = 0;                                 // 1: fp16 operand-split MFMAs (f16x3 mode), 0: exact fp32 MFMAs
    int qkv_p8 = 0;                                  // 1 (with split16, no l2norm): Q, K, V rows are in the P8 split format (written so by the qkv GEMM)
    int qkv_exp = 4, o_exp = 4;                      // site exponents of the P8 format: Q / K / V rows (qkv_p8), O rows (out_p8)
    int* status = nullptr;                           // out_p8: range guard (see GemmArgs::status)
    int cus = 0;                                     // compute units of the model's partition (0 = the whole device): grid of the persistent kernel
    // short-query kernel, Lq <= 16: the q | k | v rows of the NEW tokens (the last Lq keys) are still split-K slabs [n_slabs][B * Lq][slab_ld]
    // (row b * Lq + t; columns q | k | v x heads x 64): the workgroup sums its head's rows (slabs ascending, then slab_bias), writes them to
    // Q / K / V and reads them back from there
    const float* slabs = nullptr; int n_slabs = 0; long slab_stride = 0; int slab_ld = 0; const float* slab_bias = nullptr;
};
void launch_attention(const AttnArgs& a, hipStream_t s);
void attention_prepare();      // one-time kernel attributes (call once per process before the first captured launch)

// ---- wav2vec2 front-end ----
// per-chunk mean / unbiased std, writes (x-mean)/(std+1e-6); chunk c is read at audio + src_off[c]
void launch_audio_normalize(const float* audio, const long* src_off, float* xnorm, int n_chunks, int n, hipStream_t s);
// conv0 (Cin=1,k=10,s=5) + bias + LN(512, affine) + GELU(erf): xnorm [C, n] -> Y rows c*row_stride + t, t < T
void launch_conv0(const float* xnorm, int n, const float* w /*[512,10]*/, const float* bias, const float* lnw,
                  const float* lnb, float* Y, int n_chunks, int T, int row_stride, hipStream_t s, int out_p8 = 0, int* status = nullptr, int p8_exp = kActExp);
// multi-scale adaptive average pooling 199 -> {1,5,25,50,100} followed by SiLU: X rows c*x_tstride + t -> Y rows c*181 + tok
void launch_pool_silu(const float* X, int x_tstride, int T, float* Y, int n_chunks, const int* patch_nums, int n_lvls,
                      int D, hipStream_t s, int out_p8 = 0, int* status = nullptr, int p8_exp = kActExp);

// ---- AR / VAE glue ----
// level p: logits [B*pn, 64] -> bits[b, off..off+pn, 32]; fhat[b] += up(h_p); nextfeat[b, :pn[p+1], 32] = area(fhat) (p < 4)
void launch_ar_bits_next(const float* logits, uint8_t* bits, float* fhat, float* nextfeat, int B, int level, hipStream_t s,
                         int* status = nullptr);   // status |= 1 if a logit is not finite
// X[b*xrows + xoff + i, :] = We*feat[b,i,:] + be + pos[i,:] (i < n); if style_cond: X[b*xrows, :] = style_cond[b] + pos0
void launch_vq_embed(const float* feat, int n, const float* We, const float* be, const float* pos, float* X, int xrows,
                     int xoff, const float* style_cond, const float* pos0, int B, hipStream_t s);
// x0[b, :] = style_cond[b] + lvlpos[0]; also zeroes fhat
void launch_ar_begin(const float* style_cond, const float* lvlpos, float* x0, float* fhat, int B, hipStream_t s);
// decoder input rows [b*200 + t]: t<100: prev_fdec[b,t] + dpos[t]; t>=100: fhat[b,t-100] + h(bits level 4)[t-100] + dpos[t]
void launch_dec_input(const float* prev_fdec, const float* fhat, const uint8_t* bits, const float* dpos, float* X,
                      int B, hipStream_t s);
// motion = dec*std+mean (2nd half rows) -> out[b, chunk*100 + t, :106]; enc_in = (motion-mean)/std + epos[t] -> E [B*100,128] (cols>=106 zero)
void launch_dec_finish(const float* dec /*[B*200,106]*/, const float* mean, const float* stdv, const float* epos,
                       float* out, long out_bstride, int chunk, float* E, int B, hipStream_t s, int* status = nullptr);   // status |= 4: non-finite code
// zero motion encoder input (initial history): E[b*100+t] = (0-mean)/std + epos[t]
void launch_enc_input_zero(const float* mean, const float* stdv, const float* epos, float* E, int B, hipStream_t s);
void launch_broadcast16(const void* src, void* dst, long bytes, int B, hipStream_t s);      // dst[b] = src for b < B (bytes % 16 == 0)
// multi-scale BSQ of enc_out [B*100,32] -> hist bits [B,181,32], prev_fdec [B,100,32], ms feats [B,180,32]
void launch_bsq_history(const float* enc_out, uint8_t* hist_bits, float* prev_fdec, float* msfeat, int B, hipStream_t s,
                        int* status = nullptr);    // status |= 2 if an encoder output is not finite
// style: X[b*50+t, 0:128] = (m - mean)/std (cols >= 106 zero)
void launch_style_input(const float* motion, const float* mean, const float* stdv, float* X, int B, hipStream_t s);
// style_cond[b] = has_style[b] == 1 ? 1.1*(Ws*mean_t(feat[b]) + bs) - 0.1*null : has_style[b] == 2 ? cached[b*cached_stride ..] : null
void launch_style_finish(const float* feat /*[B*50,128]*/, const float* Ws, const float* bs, const float* null_cond,
                         const uint8_t* has_style, float* style_cond, int B, hipStream_t s, const float* cached = nullptr,
                         long cached_stride = 0);
// y[m, :] = x[m, :] + v[:]   (broadcast add of one row; used for the style PE quirk)
void launch_add_row(float* X, const float* v, int M, int D, hipStream_t s);
// Savitzky-Golay post filter of reference inference.py:89-95 on device: in [T,106] -> out [T,106]
void launch_savgol(const float* in, float* out, int T, int D, hipStream_t s);

double launch_mfma_f32_peak(float* out, int blocks, int iters, int nacc, hipStream_t s);   // calibration kernel, returns FLOPs
// sinc resampler + channel mean: x [nch][n] -> out [n_out], taps [new][2*width+orig]
void launch_resample_mean(const float* x, int nch, int n, const float* taps, int orig, int nw, int width, float* out, int n_out,
                          hipStream_t s);
// FLAME linear blend skinning (flame.hip)
void launch_flame_pose(const float* pose, float* rot, float* feat, int T, int ldf, hipStream_t s);
void launch_flame_joints(const float* vs, const float* jreg, float* J, int T, int V, hipStream_t s);
void launch_flame_skin(const float* vposed, const float* rot, const float* J, const int* parents, const float* weights, float* out,
                       int T, int V, float scale, hipStream_t s);
// headroom audit: *slot = max(*slot, bits of max |x|) over rows x cols (cols % 8 == 0) of a P8 (is_p8: written with scale 2^p8_exp) or fp32 buffer
void launch_absmax(const float* buf, int rows, int cols, long ld, int is_p8, unsigned int* slot, hipStream_t s, int junk_period = 0,
                   int junk_from = 0, int p8_exp = kActExp);
int init_ms_tables();    // uploads the (tiny) interpolation tables to the CURRENT device's __constant__ memory, once per device; 0 = ok

}  // namespace artalk
