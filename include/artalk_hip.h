/* C ABI of libartalk_hip.so: the MI355X (gfx950) implementation of ARTalk's audio->motion path.
 *
 * The reference has no FFI layer; its operator boundary is the Python object stored in
 * ARTAvatarInferEngine.ARTalk (reference inference.py:27).  The entry points below are what a binding
 * for that object needs, one per call the reference makes across the boundary:
 *
 *   artalk_create            <- BitwiseARModel(configs).eval().to(device)        inference.py:27, app/models.py:14-56
 *   artalk_set_tensor        <- one state_dict entry of load_state_dict(strict)  inference.py:28
 *   artalk_finalize_weights  <- end of load_state_dict(strict=True): missing keys are an error
 *   artalk_infer             <- BitwiseARModel.inference(batch)                  inference.py:50, app/models.py:62-121
 *                               (batched: B independent batch-1 runs; the reference asserts B==1, app/models.py:65)
 *   artalk_savgol            <- ARTAvatarInferEngine.smooth_motion_savgol        inference.py:89-95
 *   artalk_destroy           <- object lifetime
 *
 * Conventions: plain pointers and sizes only; all *_dev pointers are device memory on the model's GPU;
 * every call returns 0 on success or a negative ARTALK_E* code (text via artalk_last_error); the caller
 * owns every buffer it passes; the library owns weights and workspace; one model per GPU, calls on one
 * model are not thread-safe; work is enqueued on the hipStream_t passed as `stream` (NULL = a stream the
 * library owns) and artalk_infer / artalk_stream_* return without synchronising: the small host tables of a
 * call (chunk offsets, style flags) go through a ring of pinned staging slots, so a call blocks only when
 * four earlier calls are still in flight, or when it has to grow the workspace (artalk_reserve avoids that).
 *
 * The artalk_op_* entry points expose single kernels so that tests can check each one against the CPU
 * oracle through this same ABI; they are not needed by a binding.
 */
#ifndef ARTALK_HIP_H
#define ARTALK_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ARTALK_OK 0
#define ARTALK_EINVAL (-1)      /* bad argument / shape mismatch                      */
#define ARTALK_EKEY (-2)        /* unknown state_dict key (strict=True)               */
#define ARTALK_EMISSING (-3)    /* finalize with keys missing (strict=True)           */
#define ARTALK_EHIP (-4)        /* HIP runtime error                                  */
#define ARTALK_ESTATE (-5)      /* call order (infer before finalize, ...)            */
#define ARTALK_ECAPACITY (-6)   /* batch larger than the reserved workspace           */
#define ARTALK_EBUSY (-7)       /* artalk_poll_status: the call is still running      */

#define ARTALK_DTYPE_F32 0
#define ARTALK_DTYPE_I64 1

typedef struct artalk_model artalk_model;

/* assets/config.json + the XLS-R-300M hyper-parameters (reference app/models.py:25) */
typedef struct artalk_config {
    int32_t ar_depth, ar_heads;                 /* AR_CONFIG.T_DEPTH, T_NUM_HEADS (PREV_RATIO must be 1)  */
    int32_t vae_depth, vae_heads, vae_hidden;   /* VAE_CONFIG.T_DEPTH, T_NUM_HEADS, T_HIDDEN_DIM          */
    int32_t code_dim, motion_dim;               /* 32, 106                                                */
    int32_t n_levels; int32_t patch_nums[8];    /* 5; 1,5,25,50,100                                       */
    int32_t w2v_layers, w2v_hidden, w2v_heads, w2v_ffn;
    int32_t w2v_n_conv; int32_t w2v_conv_kernel[8]; int32_t w2v_conv_stride[8]; int32_t w2v_conv_dim;
    int32_t w2v_pos_kernel, w2v_pos_groups;
    float w2v_ln_eps;
    int32_t style_dim, style_heads, style_layers, style_ffn, style_len;
} artalk_config;

int artalk_create(int device_id, const artalk_config* cfg, artalk_model** out);
void artalk_destroy(artalk_model* m);
const char* artalk_last_error(const artalk_model* m);   /* m may be NULL: error of the last failed create */

/* One call per state_dict entry, host memory, reference key names. */
int artalk_set_tensor(artalk_model* m, const char* key, const void* host_ptr, int dtype, int ndim, const int64_t* shape);
/* Folds weight-norm, re-lays out conv weights, builds fused tables.  ARTALK_EMISSING lists missing keys in last_error. */
int artalk_finalize_weights(artalk_model* m);

/* Allocate workspace for up to max_batch clips and max_total_chunks 4-second chunks per artalk_infer call. */
int artalk_reserve(artalk_model* m, int max_batch, int max_total_chunks);
int64_t artalk_workspace_bytes(const artalk_model* m);
int64_t artalk_weight_bytes(const artalk_model* m);

/* Audio -> FLAME codes for B independent clips.
 *   audio_dev        [B][audio_clip_stride] f32, 16 kHz mono; clip b holds n_chunks[b]*64000 samples, zero padded
 *                    by the caller exactly as app/models.py:78-85 pads.
 *   n_chunks         host, [B], must be non-increasing (the host sorts clips; ragged batches stay dense prefixes).
 *   style_motion_dev [B][50][106] f32 or NULL; has_style host [B] (NULL = none): app/models.py:67-73.  has_style[b] = 1: row b
 *                    is a style clip; 2: the first 768 floats of row b are a condition computed by artalk_style_encode
 *                    (the style-clip cache: the encoder is skipped); 0: no style (null_style_cond).
 *   out_motion_dev   [B][out_clip_stride] f32, receives n_chunks[b]*100 rows of 106 per clip (caller truncates
 *                    to ceil(N/640) rows, app/models.py:115).
 *   out_bits_dev     optional [B][max_chunks][181][32] u8: the 0/1 decisions of app/models.py:104 (last scale step).
 *   out_hist_bits_dev optional [B][max_chunks+1][181][32] u8: history bits of bitwise_vae.py:78-93 (index 0 = initial).
 *   out_w2v_dev      optional [total_chunks][199][1024] f32 wav2vec2 features, chunk order (chunk index major, clip minor).
 */
int artalk_infer(artalk_model* m, const float* audio_dev, int64_t audio_clip_stride, const int64_t* n_chunks, int B,
                 const float* style_motion_dev, const uint8_t* has_style, float* out_motion_dev, int64_t out_clip_stride,
                 uint8_t* out_bits_dev, uint8_t* out_hist_bits_dev, float* out_w2v_dev, void* stream);

/* Style condition of n style clips (app/models.py:67-73: StyleEncoder -> style_cond_embed -> 1.1 c - 0.1 null), for callers that
 * keep one style across many calls as the reference's engine does (inference.py:41-45): style_motion_dev [n][50][106] ->
 * out_cond_dev [n][768].  Pass a condition back through artalk_infer / artalk_stream_begin with has_style[b] = 2. */
int artalk_style_encode(artalk_model* m, const float* style_motion_dev, int n, float* out_cond_dev, void* stream);

/* Streaming form of the same path (chunk-at-a-time, history kept in the model between calls): the reference
 * loop body of app/models.py:92-114 for B parallel streams.  artalk_stream_begin computes the style condition and the initial
 * history (app/models.py:67-73,86-89); every artalk_stream_chunk consumes the next 64000 samples of each stream
 * (audio_dev [B][chunk_stride], zero padded by the caller at the end of a clip) and writes 100 x 106 codes per stream to
 * out_motion_dev [B][out_stride].  artalk_stream_end, a new artalk_stream_begin or a call to artalk_infer ends the session
 * (shared workspace); growing the workspace (artalk_reserve) ends it too, and artalk_stream_chunk then fails with ARTALK_ESTATE. */
int artalk_stream_end(artalk_model* m);
int artalk_stream_begin(artalk_model* m, int B, const float* style_motion_dev, const uint8_t* has_style, void* stream);
int artalk_stream_chunk(artalk_model* m, const float* audio_dev, int64_t chunk_stride, float* out_motion_dev, int64_t out_stride,
                        void* stream);

/* FLAME linear blend skinning: the consumer behind BITWISE_VAE.get_flame_verts (bitwise_vae.py:43-57) ->
 * FLAMEModel.forward(no_lmks=True) (app/flame_model/FLAME.py:117-142) -> lbs (app/flame_model/lbs.py:142-233).  Host arrays in
 * the layouts of the reference's buffers: v_template [V][3], shapedirs [V][3][NB], posedirs_t [V*3][36] (the reference buffer
 * transposed), J_regressor [5][V], parents [5], lbs_weights [V][5].  artalk_flame_verts: betas_dev [T][NB] (shape ++ expression),
 * full_pose_dev [T][15] (global, neck, jaw, eyes) -> out_dev [T][V][3], already multiplied by `scale`. */
typedef struct artalk_flame artalk_flame;
int artalk_flame_create(int device_id, int V, int NB, int P, const float* v_template, const float* shapedirs, const float* posedirs_t,
                        const float* J_regressor, const int32_t* parents, const float* lbs_weights, float scale, artalk_flame** out);
int artalk_flame_verts(artalk_flame* f, const float* betas_dev, const float* full_pose_dev, int T, float* out_dev, void* stream);
void artalk_flame_destroy(artalk_flame* f);
const char* artalk_flame_last_error(const artalk_flame* f);

/* Numerical health of the work enqueued since the last artalk_infer / artalk_stream_begin started.  Every call ends with an
 * asynchronous copy of the device status word to pinned host memory.  artalk_get_status WAITS for that copy (an event wait:
 * the one synchronisation on this boundary, paid only by callers that ask; `stream` is ignored); artalk_poll_status never
 * blocks and returns ARTALK_EBUSY while the call is still running.  Bits: 0 a logit was NaN/Inf (the pairwise argmax of
 * app/models.py:104 would silently turn it into a 0 bit), 1 a re-encoder output was NaN/Inf, 2 a FLAME code was NaN/Inf,
 * 3 an activation exceeded the range of the P8 split format (|x| >= 4094) where it was produced.  Non-zero in f16x3 mode
 * means an activation left fp16's range: redo the call in f32 mode (the Python host does, and stays in f32). */
int artalk_get_status(artalk_model* m, int* flags, void* stream);
int artalk_poll_status(artalk_model* m, int* flags);
/* Several calls in flight (a serving loop that enqueues batch i+1 before it looks at batch i): every call that publishes a status
 * word gets a ticket (1, 2, ...; artalk_last_ticket right after the call returns it), and artalk_get_status_of waits for THAT call
 * and returns its flags.  The words of the last 4 calls are kept: ARTALK_EINVAL for an older ticket. */
long long artalk_last_ticket(artalk_model* m);
int artalk_get_status_of(artalk_model* m, long long ticket, int* flags);

/* Savitzky-Golay smoothing of inference.py:89-95 on the device: in/out [T][106] f32, T >= 9. */
int artalk_savgol(artalk_model* m, const float* in_dev, float* out_dev, int T, void* stream);

/* Per-call stage timing (HIP events on the call's stream).  level 0 = off; 1 = light: hipGraphs stay on, events
 * bracket the eager launches only (wav2vec2, AdaLN table, graph replays) - cheap enough for a timed region;
 * 2 = full: graphs off, events also inside the AR/VAE body.  artalk_get_profile synchronises and returns the
 * LAST artalk_infer's numbers (milliseconds / counters):
 *   out[0] style  out[1] wav2vec2 conv stack  out[2] wav2vec2 encoder  out[3] AdaLN table GEMM
 *   out[4] AR scale steps (level 1: whole captured body)  out[5] VAE decode+re-encode (level 2 only; + initial history)
 *   out[6] total  out[7] bracketed launches of the dominant kernel (f16x3 mode: gemm_p8_big_kernel, the persistent 320x256 / 256x256-tile
 *   LDS-DMA split GEMM that takes every large GEMM of a step: wav2vec2 q|k|v, out-projection, FFN-in, FFN-out, the feature projection,
 *   conv1-6 and the AdaLN tables; f32 mode: the 128x128-tile fp32 MFMA GEMM)  out[8] their summed ms  out[9] their summed FLOP */
int artalk_set_profiling(artalk_model* m, int level);
int artalk_get_profile(artalk_model* m, double* out, int n);
/* Kernel-time budget without a profiler: after an artalk_infer at profiling level 3 (graphs off, one clip group, every launch carries its
 * own start / stop events), out[b] = summed kernel durations (ms) per bucket - 0 style, 1 conv stack, 2 encoder, 3 AdaLN tables,
 * 4 history K/V + glue, 5..9 scale steps 0..4, 10 VAE decode, 11 re-encode, 12 other - and out[13] = kernels timed (n >= 14).
 * bench.py reports it next to the stages' event times as `budget_ms`. */
int artalk_get_kernel_sums(artalk_model* m, double* out, int n);
/* GEMM arithmetic: 0 (default) = exact fp32 on v_mfma_f32_32x32x2_f32; 1 = "f16x3": every fp32 operand split into two
 * fp16 values (22 significand bits), three fp16 MFMA products accumulated in fp32 - fp32-class accuracy (parity tests run in
 * both modes) at 5.3x the matrix-core rate; the logit / code heads stay on the fp32 path in both modes. */
int artalk_set_precision(artalk_model* m, int mode);
/* Replay the AR/VAE part from hipGraphs captured per active-batch size (default 1 = on).  The batch is cut into 1/2/4 clip
 * groups whose graphs run concurrently on separate streams (automatic; enable | (groups << 8) forces a count, for tuning).
 * Bits 16-23 / 24-31, when non-zero, override the split-K policy in units of 16 tiles (split when a GEMM has fewer output
 * tiles than the first, aim for the second; defaults 192 / 384 measured best). */
int artalk_set_graphs(artalk_model* m, int enable);
/* Number of captured body graphs the model holds; *captures (nullable) = captures since the model was created.  The cache is bounded:
 * a group's re-encode count is rounded up to a multiple of 8 and at most 96 executables are kept (least recently used evicted), so a
 * serving loop with ever-changing ragged batches (reference: one clip per call, inference.py:47-57) cannot grow it without limit. */
int artalk_graph_count(const artalk_model* m, long long* captures);
/* Headroom audit of the f16x3 operand format: while enabled every artalk_infer runs without graphs and records, for each
 * producer of a P8 operand (LayerNorm outputs, GEMM results written in P8, attention outputs, ...), max |x| - times the site's scale
 * (16 by default, artalk_get_scales) the value that must stay below fp16's 65504.  Works in both precision modes (in f32 mode the same
 * activations are fp32 buffers).  artalk_get_audit synchronises and returns the number of sites; names_buf receives the site
 * names NUL-separated, values[i] the maximum seen at site i since the audit was switched on (tools/p8_headroom.py). */
int artalk_set_audit(artalk_model* m, int enable);
int artalk_get_audit(artalk_model* m, char* names_buf, int buf_len, float* values, int max_n);
/* Per-site operand scales of the f16x3 format.  Every producer of a P8 operand writes it with a power-of-two scale 2^e and its consumers
 * remove the same scale; e = 4 (x16, range |x| < 4094) by default.  A checkpoint with outlier activations (XLS-R-class encoders: FFN
 * hidden channels of 1e4 and more; the reference loads such a checkpoint, inference.py:24-28) needs more range at a few sites:
 * artalk_calibrate reads the maxima of an audit pass (artalk_set_audit(1) + artalk_infer in EXACT-F32 mode on representative clips)
 * and lowers e at every site where max|x| * 2^e * headroom would exceed fp16's 65504, per site, down to e = -8.  Only those sites change
 * (everything else stays bit-identical); exponents never go up again until artalk_reset_scales.  Returns the number of sites changed,
 * or a negative ARTALK_E* code.  artalk_get_scales: exps[i] = exponent of audit site i (the order of artalk_get_audit).
 * The Python host calls this by itself when a call trips the range guard (status bit 3) and re-runs the call in f16x3 mode. */
int artalk_calibrate(artalk_model* m, float headroom);
int artalk_reset_scales(artalk_model* m);
int artalk_get_scales(artalk_model* m, int* exps, int max_n);
/* Intermediate taps: device buffers that correspond to intermediates of the reference, for parity tests that localise a difference to
 * a kernel group (tests/test_taps_gpu.py against tests/golden/taps_*.npz, captured from the reference by oracle/make_golden_taps.py).
 * While a tap buffer is set, artalk_infer runs the AR/VAE body eagerly as one clip group and copies, for chunk index j and clip
 * position b (clips in the call's sorted order), into slot (j * max_batch + b), as fp32 (artalk_tap_layout gives the field offsets):
 *   blk0_in  [181][768]  attn_feat entering attn_blocks[0], rows of scale step p written at step p        (app/models.py:100)
 *   blk0_out [181][768]  output of attn_blocks[0]                                                          (app/transformer.py:30-43)
 *   blkL_out [181][768]  output of the last block                                                          (app/models.py:101-102)
 *   prev_in  [181][768]  prev_attn_feat + prev_lvl_pos_embed as chunk j uses it                            (app/models.py:101,111-114)
 *   logits   [181][64]   pred_motion_logits, fp32                                                          (app/models.py:103)
 *   dec_out  [200][106]  VAE decoder output before unnorm_with_stats                                       (app/modules/bitwise_vae.py:110-111)
 * With the KV cache only a scale step's NEW tokens are computed, so row t holds the value of the step that introduced token t -
 * which the reference recomputes, bit-identically, in every later step.  tap_dev = NULL switches the taps off. */
int artalk_set_tap(artalk_model* m, float* tap_dev, int max_batch, int max_chunks);
int artalk_tap_layout(int64_t* out, int n);
/* Restrict the model to a set of compute units: bit i of mask[0 .. n_words) = CU i / 8 of XCD i % 8 on MI355X.  The library's own
 * streams get the mask and its persistent kernels size their grids to it; the caller passes a stream created with the same mask
 * (artalk_op_create_masked_stream) to artalk_infer - e.g. to leave compute units to a renderer running beside the path.  (Two
 * replicas of the path on the two halves of the chip measured 18 % BELOW one replica on the whole chip: tools/dual_partition_probe.py,
 * DESIGN.md section 6.)  n_words = 0 clears it.  A stream created with a CU mask (hipExtStreamCreateWithCUMask) is a BLOCKING stream:
 * unlike the library's unmasked side streams (hipStreamNonBlocking) it is implicitly ordered against work on the legacy null stream
 * of the process, and a null-stream call made while the library captures a graph on it is a capture error - keep other GPU work of
 * the process (a renderer) on streams of its own while a partitioned model runs.  The mask count is clamped to the device's CUs. */
int artalk_set_cu_mask(artalk_model* m, const uint32_t* mask, int n_words);

/* ---- single-kernel entry points for the parity tests (device pointers, row-major f32) ---- */
/* C[M,N] = R + gate * act(A[M,K] W[N,K]^T + bias); act: 0 none, 1 gelu(erf), 2 gelu(tanh), 3 leaky_relu(0.2). K % 32 == 0 */
int artalk_op_gemm(const float* A, int64_t lda, const float* W, const float* bias, const float* gate, const float* R,
                   float* C, int M, int N, int K, int act, void* stream);
/* same, with an explicit tile configuration (4: 128x128 BK16, 2: 64x64, 1: 128x64, 3: 32x128, 0/5/6/7: tuning variants; -1: heuristic) for tuning */
int artalk_op_gemm_ex(const float* A, int64_t lda, const float* W, const float* bias, float* C, int M, int N, int K, int act,
                      int force_cfg, void* stream);
/* calibration: register-only fp32 MFMA loop (blocks x 256 threads, 32*iters MFMAs per wave, nacc = 1 or 4 independent accumulators); *flops = FLOPs of the launch */
int artalk_op_mfma_f32_peak(float* out_dev, int blocks, int iters, int nacc, double* flops, void* stream);
/* f16x3 split GEMM (mode 1 above) on fp32 inputs; cfg 0: 128x128 tiles, 1: 64x64, -1: heuristic.  M > 32, K % 32 == 0 */
int artalk_op_gemm_f16s(const float* A, int64_t lda, const float* W, const float* bias, float* C, int M, int N, int K, int act,
                        int force_cfg, void* stream);
/* tuning helpers: fp32 -> packed split words; split GEMM on pre-packed W (and optionally pre-packed A) */
int artalk_op_pack_split(const float* in, void* out_u32, int64_t n, int is_weight, void* stream);   /* operand scale: 0 activation, 1 weight */
int artalk_op_gemm_f16s_packed(const void* A, int a_packed, int64_t lda, const void* Wp, const float* bias, float* C, int M, int N,
                               int K, int act, int force_cfg, void* stream);
/* frees the split-K scratch artalk_op_gemm_f16s_packed grows on demand (the buffers it outgrew are kept until this call, because
 * graphs captured from earlier launches still write to them): call it when no such graph will be replayed again */
int artalk_op_release_scratch(void);
/* which production LDS-DMA kernel launch_gemm_p8 picks for an M x N x K product (dense rows, with or without a residual) with both
 * operands in P8 and no forced configuration (what the model path calls): 7 / 12 = gemm_p8_big_kernel with 256x256 / 320x256 tiles,
 * 8 = gemm_p8_2wgp_kernel; force_cfg 99 of artalk_op_gemm_f16s_packed launches exactly that choice */
int artalk_op_gemm_p8_plan(int M, int N, int K, int residual);
/* a HIP stream restricted to the compute units whose bits are set in mask[0 .. n_words) (hipExtStreamCreateWithCUMask; on MI355X bit i =
 * CU i / 8 of XCD i % 8): the stream a model with artalk_set_cu_mask is driven on */
int artalk_op_create_masked_stream(const uint32_t* mask, int n_words, void** out_stream);
int artalk_op_destroy_stream(void* stream);
/* y = LN(x)[*w+b][*(1+scale)+shift][act], D in {128,512,768,1024}; act | 0x100 writes y in the P8 split format */
int artalk_op_layernorm(const float* X, float* Y, const float* w, const float* b, const float* scale, const float* shift,
                        int M, int D, float eps, int act, void* stream);
/* Q,K,V,O: [B][L][H*HD] contiguous; l2norm bit 0 -> q,k normalised, q *= qscale[h]; bit 1 -> the fp16 operand-split kernel of
 * f16x3 mode (HD = 64); bit 2 (with bit 1, without bit 0) -> Q, K, V are given in the P8 split format; split: queries<split see keys<split */
int artalk_op_attention(const float* Q, const float* K, const float* V, float* O, int B, int H, int HD, int Lq, int Lk,
                        float scale, int l2norm, const float* qscale, int split, void* stream);
/* audio [C][n] -> normalised -> conv0+LN+GELU: Y [C][T][512], T = (n-10)/5+1 */
int artalk_op_w2v_front(const float* audio, int C, int n, const float* w, const float* bias, const float* lnw,
                        const float* lnb, float* xnorm_out, float* Y, void* stream);
/* audio front-end of inference.py:230-231: polyphase sinc FIR (torchaudio Resample defaults) + mean over channels.
 * x [nch][n] f32, taps [nw][2*width+orig] f32 (host-built, see artalk_amd/audio.py), out [n_out], n_out = ceil(n*nw/orig) */
int artalk_op_resample_mean(const float* x, int nch, int n, const float* taps, int orig, int nw, int width, float* out, int n_out,
                            void* stream);
/* X [C][T][D] -> Y [C][181][D]: area pooling to {1,5,25,50,100} then SiLU */
int artalk_op_pool_silu(const float* X, int C, int T, int D, float* Y, void* stream);
/* enc_out [B][100][32] -> hist_bits [B][181][32] u8, prev_fdec [B][100][32], msfeat [B][180][32] */
int artalk_op_bsq_history(const float* enc_out, uint8_t* hist_bits, float* prev_fdec, float* msfeat, int B, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ARTALK_HIP_H */
