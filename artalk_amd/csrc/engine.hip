// Model object behind the C ABI (include/artalk_hip.h): weight table in the reference's key names, derived
// layouts, workspace, and the launch sequence of the audio->motion path.
//
// Execution plan (differs from the reference's loop order, app/models.py:92-114, same arithmetic per token):
//   1. style encoder for all clips                                   (app/models.py:67-73)
//   2. wav2vec2 for ALL 4-second chunks of ALL clips in one batched pass - chunk i of the audio does not depend on
//      the AR state (app/models.py:93), so M = chunks*200 rows per GEMM instead of 199
//   3. for each chunk index j (sequential: chunk j+1 needs the re-encoded motion of chunk j), all clips that
//      still have a chunk j batched:  AdaLN table for all 12 blocks + head in ONE GEMM (the conditioning is
//      layer-independent, app/transformer.py:32), K/V of the 181 history tokens once per layer, then the 5
//      scale steps with a per-layer KV cache (only the new level's tokens run), VAE decode,
//      re-encode + quantise to the next history.
#include "kernels.h"
#include "../../include/artalk_hip.h"

#include <hip/hip_ext.h>
#include <rocprofiler-sdk-roctx/roctx.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

using namespace artalk;

namespace artalk {
thread_local KernelTimer* g_ktimer = nullptr;
bool ktimer_events(hipEvent_t* e0, hipEvent_t* e1) {
    KernelTimer* t = g_ktimer;
    if (!t) return false;
    while (t->pool.size() < t->used + 2) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return false;
        t->pool.push_back(e);
    }
    *e0 = t->pool[t->used]; *e1 = t->pool[t->used + 1];
    t->recs.emplace_back(t->bucket, t->used);
    t->used += 2;
    return true;
}
}  // namespace artalk

namespace {
// buckets of the kernel timer (artalk_get_kernel_sums): stages of the path, the scale steps of the body
enum { KB_STYLE = 0, KB_CONV = 1, KB_ENC = 2, KB_ADA = 3, KB_AR_MISC = 4, KB_LEVEL0 = 5, KB_VAE_DEC = 10, KB_REENC = 11, KB_OTHER = 12, KB_N = 13 };

thread_local std::string g_create_error;

constexpr int kSamplesPerChunk = 64000;
constexpr int kE = 768;      // embed dim (app/models.py:19)
constexpr int kCond = 1024;  // audio feature dim (app/models.py:27)
constexpr int kNTok = 181;
constexpr int kMaxLv = 5;

// roctx range around one kernel group of the path (SURVEY.md section 5, K-groups): visible with `rocprofv3 --marker-trace`.
// Host-side ranges: they bracket the ENQUEUE of the group (the path is asynchronous); inside a hipGraph capture nothing is
// enqueued at replay time, so the captured body shows up as one "body.graph" range per clip group and its inner ranges
// appear only when graphs are off (artalk_set_graphs(0) / profiling level 2).
struct Range {
    explicit Range(const char* name) { roctxRangePushA(name); }
    ~Range() { roctxRangePop(); }
    Range(const Range&) = delete;
    Range& operator=(const Range&) = delete;
};

enum SlotKind { SK_DIRECT, SK_PADK, SK_CONV, SK_HOST, SK_IGNORE };

struct Slot {
    std::vector<int64_t> shape;
    int dtype = ARTALK_DTYPE_F32;
    SlotKind kind = SK_DIRECT;
    float* dst = nullptr;      // device destination (DIRECT/PADK/CONV)
    int64_t pad_to = 0;        // PADK: padded inner size
    bool set = false;
    std::vector<float> host;   // SK_HOST (and every small tensor) keeps a host copy
    std::vector<int64_t> host_i64;
    int64_t numel() const { int64_t n = 1; for (auto d : shape) n *= d; return n; }
};

struct W2VLayer { float *qkv_w, *qkv_b, *out_w, *out_b, *ln1w, *ln1b, *ln2w, *ln2b, *ff1_w, *ff1_b, *ff2_w, *ff2_b; };
struct ARLayer { float *qkv_w, *qkv_b, *proj_w, *proj_b, *ffn1_w, *ffn1_b, *ffn2_w, *ffn2_b, *qscale; };
struct VAELayer { float *lnw, *lnb, *qkv_w, *out_w, *out_b, *m1_w, *m1_b, *m2_w, *m2_b; };
struct VAESide { float *in_w, *in_b, *out_w, *out_b; std::vector<VAELayer> layers; };
struct StyleLayer { float *in_w, *in_b, *out_w, *out_b, *l1_w, *l1_b, *l2_w, *l2_b, *n1w, *n1b, *n2w, *n2b; };

// Site exponents of the P8 operand format (common.h): every producer of a P8 operand writes it with scale 2^e and every consumer removes
// the same scale, e = kActExp (16) by default.  artalk_calibrate lowers the exponent of a site whose activations need the range
// (a real XLS-R-class checkpoint: FFN hidden channels of 1e4 and more) - per SITE, so that one outlier channel in a few layers
// does not cost the whole model its fast mode.  Indexed like the audit's site names.
struct SiteExps {
    int conv[8]; int fp_ln, posconv_in;
    struct W2V { int ln1, qkv, attn, ln2, ffn; } w2v[64];
    int silu_cond, hist_tok;
    struct AR { int ln1, attn, ln2, ffn; } ar[32];
    struct VAE { int ln, qkv, attn, resid, mlp; } vae[2][16];
    int vae_enc_in, vae_dec_in, vae_dec_head;
    int style_in; struct ST { int h, attn, h1, ffn; } style[8];      // style encoder: fp32 A rows split while staging
    SiteExps() { int* p = reinterpret_cast<int*>(this); for (size_t i = 0; i < sizeof(SiteExps) / sizeof(int); ++i) p[i] = kActExp; }
};

struct Workspace {
    int maxB = 0, maxC = 0, G = 0;   // G = wav2vec2 chunk-group size
    // wav2vec2
    long* src_off = nullptr; float *xnorm = nullptr, *convA = nullptr, *convB = nullptr;
    float *h0 = nullptr, *h1 = nullptr, *xln = nullptr, *qkv = nullptr, *att = nullptr, *ffn = nullptr;
    float* silu_cond = nullptr;     // [maxC*181, 1024]
    // AR
    float *ada = nullptr, *style_cond = nullptr, *prev_in = nullptr, *prev_in_p8 = nullptr, *cache = nullptr;
    float *x = nullptr, *xmod = nullptr, *attn_out = nullptr, *ffn_h = nullptr, *logits = nullptr;
    float *fhat = nullptr, *nextfeat = nullptr;
    float* splitk = nullptr; int64_t splitk_floats = 0;   // partial sums of split-K GEMMs
    float* splitk_b[4] = {nullptr, nullptr, nullptr, nullptr};   // one scratch per concurrent branch of the AR body (splitk_b[0] == splitk)
    uint8_t *bits = nullptr, *hist_bits = nullptr, *has_style = nullptr;
    int* status = nullptr;           // device word: bit 0 non-finite logit, bit 1 non-finite re-encoder output (see artalk_get_status)
    // VAE
    float *prev_fdec = nullptr, *msfeat = nullptr, *dec_x = nullptr, *vh = nullptr, *vln = nullptr, *vqkv = nullptr;
    float *vatt = nullptr, *vmlp = nullptr, *dec_out = nullptr, *enc_in = nullptr, *enc_out = nullptr, *motion_chunk = nullptr;
    // style
    float *s_in = nullptr, *s_h = nullptr, *s_qkv = nullptr, *s_att = nullptr, *s_ffn = nullptr, *s_tmp = nullptr;
    int64_t bytes = 0;
};

}  // namespace

struct artalk_model {
    artalk_config cfg{};
    int device = 0;
    std::string err;
    bool finalized = false;
    std::map<std::string, Slot> slots;
    std::vector<void*> allocs;      // weights (model lifetime)
    std::vector<void*> ws_allocs;   // workspace (re-allocated when a larger batch arrives)
    int64_t weight_bytes = 0;
    struct PackRange { const float* base; int64_t n; unsigned int* packed; };
    std::vector<PackRange> wranges;   // every weight allocation and its packed f16x3 copy (built at finalize)
    int precision = 0;                // 0: fp32 MFMA everywhere, 1: f16x3 split GEMMs (heads stay fp32)
    int splitk_tiles = 192, splitk_target = 384;   // split-K when the grid has fewer tiles than splitk_tiles; aim at splitk_target workgroups
    // headroom audit (artalk_set_audit): max |x| * 16 at every producer of a P8 operand, by site name
    bool audit = false;
    unsigned int* audit_vals = nullptr;            // device, kAuditSlots floats (as bits)
    std::vector<std::string> audit_names;
    std::map<std::string, int> audit_index;
    std::vector<int*> audit_exp;                   // the site's exponent (SiteExps member) per audit slot: what artalk_calibrate adjusts
    SiteExps ex;                                   // site exponents of the P8 format (all kActExp until a calibration lowers some)
    int scales_changed = 0;                        // sites whose exponent differs from kActExp
    // The INITIAL history of a clip (app/models.py:86-89: encode + quantise an all-zero motion) is a function of the weights only - the
    // same bits, decoder features and history tokens for every clip of every call.  It is computed once per precision mode (for ONE
    // clip, through the same run_reencode as every later history) and broadcast to the batch afterwards; only the style token in
    // front of the history tokens differs per clip (launch_vq_embed).  [0] exact-f32 mode, [1] f16x3 mode.
    struct InitHistory { uint8_t* bits = nullptr; float *fdec = nullptr, *msfeat = nullptr; bool valid = false; } init_hist[2];
    // intermediate taps (artalk_set_tap): parity tests compare these device buffers with intermediates captured from the reference
    float* tap = nullptr; int tap_B = 0, tap_maxch = 0, tap_chunk = 0;
    std::vector<uint32_t> cu_mask;    // artalk_set_cu_mask: the library's own streams are restricted to these compute units
    int n_cus = 0;                    // their number (0 = the whole device): grid of the persistent one-workgroup-per-CU kernels
    int stream_B = 0;                 // streams opened by artalk_stream_begin (history lives in the workspace)
    Workspace* view = nullptr;        // workspace view (clip sub-range) the body launchers currently work on; null = m->ws
    bool sticky_error = false;        // set by internal consistency checks inside the launch sequence; reported by artalk_infer
    // derived sizes
    int n_conv = 0; int conv_T[8]{}; int conv_S[8]{};   // valid frames / padded row stride per conv layer output
    int Tw = 0, Ts = 0;                                 // 199, 200
    int ada_n = 0;                                      // 12*4608 + 1536
    int pn[kMaxLv]{}, off[kMaxLv + 1]{};
    // weights
    float *conv0_w = nullptr, *conv_b[8]{}, *conv_lnw[8]{}, *conv_lnb[8]{}, *conv_w[8]{};
    float *fp_lnw = nullptr, *fp_lnb = nullptr, *fp_w = nullptr, *fp_b = nullptr;
    float *pos_w = nullptr, *pos_b = nullptr, *enc_lnw = nullptr, *enc_lnb = nullptr;
    std::vector<W2VLayer> w2v;
    float *ada_w = nullptr, *ada_b = nullptr;
    float *ar_qkv_w = nullptr, *ar_qkv_b = nullptr;   // [depth][3E][E] / [depth][3E]: every block's q|k|v weights, contiguous
    std::vector<ARLayer> ar;
    float *logits_w = nullptr, *logits_b = nullptr, *vq_w = nullptr, *vq_b = nullptr, *sc_w = nullptr, *sc_b = nullptr;
    float *null_style = nullptr, *lvl_pos = nullptr, *prev_lvl_pos = nullptr;
    float *enc_pos = nullptr, *dec_pos = nullptr, *vae_mean = nullptr, *vae_std = nullptr;
    VAESide enc, dec;
    float *st_mean = nullptr, *st_std = nullptr, *st_pe = nullptr, *st_proj_w = nullptr, *st_proj_b = nullptr;
    std::vector<StyleLayer> style;
    Workspace ws;
    // profiling
    int profiling = 0;        // 0 off, 1 light (graphs stay on; events around eager launches only), 2 full (graphs off), 3 kernel timer (graphs off, one clip group, every launch timed)
    KernelTimer ktimer;       // level 3: per-kernel durations by bucket (artalk_get_kernel_sums)
    bool use_graphs = true;
    bool in_graph_body = false;   // capturing: no events
    bool in_body = false;         // inside run_chunk_body (captured or eager)
    std::vector<hipEvent_t> ev_pool; size_t ev_used = 0;
    std::vector<std::pair<size_t, double>> dom_events;   // (event index of start, flops)
    std::vector<std::pair<int, size_t>> marks;          // (bucket of the interval ending here, event index), caller's stream
    hipStream_t prof_stream = nullptr;
    hipStream_t side_stream[3] = {nullptr, nullptr, nullptr};   // extra branches of the AR body (run_chunk_body_graphs)
    // The split-K reduce of the 1- / 5-token q|k|v GEMMs inside the short-query attention kernel (run_chunk_body): 12 launches fewer per
    // scale step (113 -> 101), bit-identical (tests/test_edge_cases_gpu.py::test_fused_qkv_reduce_is_bit_identical) - and NOT faster: same
    // box, alternating runs, body time 31.33 / 31.51 / 31.33 ms fused against 31.00 / 31.18 / 31.12 ms with the separate reduce pass
    // (round 4).  The attention launch grows by what the reduce launch took: inside a graph a 4.8 us kernel costs less than its
    // duration (the front end has the next dispatch ready), while the slab loads now sit on the attention kernel's own critical path.
    // Off by default (ARTALK_FUSE_QKV_REDUCE=1 for an A/B run).
    bool fuse_qkv_reduce = false;
    hipEvent_t fork_ev = nullptr, join_ev[3] = {nullptr, nullptr, nullptr};
    int branches = 0;                 // 0 = automatic (2 for B >= 8), else forced 1/2/4
    hipStream_t own_stream = nullptr;   // used when the caller passes stream == NULL (graph capture needs a real stream)
    // graphs: one per (active batch size, clip group, precision, re-encoded clips), LRU-bounded (run_chunk_body_graphs)
    struct GraphEntry { hipGraphExec_t exec = nullptr; long long last_use = 0; };
    std::map<long long, GraphEntry> graphs;
    long long graph_clock = 0, graph_captures = 0;
    // Pinned host staging for the small per-call tables (chunk offsets, style flags): a ring of slots, each guarded by an event
    // recorded after its H2D copies, so artalk_infer never has to wait for its own copies (it blocks only if kStageSlots calls
    // are still in flight).  h_status receives the device status word at the end of every call (async), status_ev marks it.
    static constexpr int kStageSlots = 4;
    struct Stage { long* src = nullptr; uint8_t* has = nullptr; hipEvent_t done = nullptr; bool used = false; };
    Stage stage[kStageSlots];
    int stage_cap_c = 0, stage_cap_b = 0, stage_next = 0;
    // one slot per call in flight (ring of kStatusSlots, indexed by the call's ticket): a caller that keeps several calls queued reads
    // each call's own word (artalk_get_status_of)
    static constexpr int kStatusSlots = 4;
    int* h_status = nullptr; hipEvent_t status_ev[kStatusSlots] = {nullptr, nullptr, nullptr, nullptr}; bool status_pending = false;
    long long ticket = 0;          // number of calls that have published a status word; the last one's ticket is `ticket`, slot (ticket - 1) % kStatusSlots
};

namespace {

#define HIPCHK(m, call)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            (m)->err = std::string(#call) + ": " + hipGetErrorString(e_);                       \
            return ARTALK_EHIP;                                                                 \
        }                                                                                       \
    } while (0)

int fail(artalk_model* m, int code, const std::string& msg) { m->err = msg; return code; }
// the streaming / style-encode entry points run without stage events; the level comes back on every exit path
struct ProfilingOff {
    artalk_model* m; int saved;
    explicit ProfilingOff(artalk_model* x) : m(x), saved(x->profiling) { m->profiling = 0; }
    ~ProfilingOff() { m->profiling = saved; }
    ProfilingOff(const ProfilingOff&) = delete;
    ProfilingOff& operator=(const ProfilingOff&) = delete;
};

template <typename T>
T* dalloc_in(std::vector<void*>& pool, int64_t n) {
    void* p = nullptr;
    if (n <= 0) n = 1;
    if (hipMalloc(&p, n * sizeof(T)) != hipSuccess) { pool.push_back(nullptr); return nullptr; }
    (void)hipMemset(p, 0, n * sizeof(T));
    pool.push_back(p);
    return reinterpret_cast<T*>(p);
}
template <typename T>
T* dalloc(artalk_model* m, int64_t n) { return dalloc_in<T>(m->allocs, n); }

float* walloc(artalk_model* m, int64_t n) {
    m->weight_bytes += n * 4;
    float* p = dalloc<float>(m, n);
    m->wranges.push_back({p, n, nullptr});
    return p;
}

const unsigned int* packed_of(const artalk_model* m, const float* w) {
    for (const auto& r : m->wranges)
        if (r.packed && w >= r.base && w < r.base + r.n) return r.packed + (w - r.base);
    return nullptr;
}

void add_slot(artalk_model* m, const std::string& key, std::vector<int64_t> shape, SlotKind kind, float* dst,
              int64_t pad_to = 0, int dtype = ARTALK_DTYPE_F32) {
    Slot s;
    s.shape = std::move(shape); s.kind = kind; s.dst = dst; s.pad_to = pad_to; s.dtype = dtype;
    m->slots[key] = std::move(s);
}

// Linear: weight [out,in] (+bias) copied as is.
void add_linear(artalk_model* m, const std::string& p, int out_f, int in_f, float*& w, float*& b, bool bias = true) {
    w = walloc(m, (int64_t)out_f * in_f);
    add_slot(m, p + ".weight", {out_f, in_f}, SK_DIRECT, w);
    if (bias) { b = walloc(m, out_f); add_slot(m, p + ".bias", {out_f}, SK_DIRECT, b); } else b = nullptr;
}
void add_vec(artalk_model* m, const std::string& key, int n, float*& v) {
    v = walloc(m, n);
    add_slot(m, key, {n}, SK_DIRECT, v);
}
void add_ln(artalk_model* m, const std::string& p, int n, float*& w, float*& b) {
    add_vec(m, p + ".weight", n, w);
    add_vec(m, p + ".bias", n, b);
}

int build_registry(artalk_model* m) {
    const artalk_config& c = m->cfg;
    const int NT = kNTok;
    // ---- top level (app/models.py:19-56) ----
    add_slot(m, "null_style_cond", {1, 1, kE}, SK_DIRECT, m->null_style = walloc(m, kE));
    add_slot(m, "pos_embed", {1, NT, kE}, SK_HOST, nullptr);
    add_slot(m, "prev_pos_embed", {1, NT, kE}, SK_HOST, nullptr);
    add_slot(m, "attn_bias_for_masking", {1, 1, NT, 2 * NT}, SK_HOST, nullptr);
    add_slot(m, "lvl_idx", {1, NT}, SK_HOST, nullptr, 0, ARTALK_DTYPE_I64);
    add_slot(m, "lvl_embed.weight", {c.n_levels, kE}, SK_HOST, nullptr);
    m->lvl_pos = walloc(m, (int64_t)NT * kE);
    m->prev_lvl_pos = walloc(m, (int64_t)NT * kE);
    add_linear(m, "vqfeat_embed", kE, c.code_dim, m->vq_w, m->vq_b);
    add_linear(m, "style_cond_embed", kE, c.style_dim, m->sc_w, m->sc_b);
    add_linear(m, "logits_head", 2 * c.code_dim, kE, m->logits_w, m->logits_b);
    // fused AdaLN table: rows [l*6E, (l+1)*6E) = block l, rows [depth*6E, depth*6E+2E) = head
    m->ada_n = c.ar_depth * 6 * kE + 2 * kE;
    m->ada_w = walloc(m, (int64_t)m->ada_n * kCond);
    m->ada_b = walloc(m, m->ada_n);
    add_slot(m, "cond_logits_head.ada_lin.1.weight", {2 * kE, kCond}, SK_DIRECT, m->ada_w + (int64_t)c.ar_depth * 6 * kE * kCond);
    add_slot(m, "cond_logits_head.ada_lin.1.bias", {2 * kE}, SK_DIRECT, m->ada_b + c.ar_depth * 6 * kE);
    m->ar.resize(c.ar_depth);
    // q|k|v weights of all blocks in ONE allocation ([depth][3E][E]): the K/V projections of the 181 history tokens are the same
    // input through every block's key / value weights, and run as one GEMM over column groups (run_chunk_body)
    m->ar_qkv_w = walloc(m, (int64_t)c.ar_depth * 3 * kE * kE);
    m->ar_qkv_b = walloc(m, (int64_t)c.ar_depth * 3 * kE);
    for (int i = 0; i < c.ar_depth; ++i) {
        ARLayer& L = m->ar[i];
        const std::string p = "attn_blocks." + std::to_string(i);
        add_slot(m, p + ".attn.scale_mul_1H11", {1, c.ar_heads, 1, 1}, SK_HOST, nullptr);
        L.qscale = walloc(m, c.ar_heads);
        L.qkv_w = m->ar_qkv_w + (int64_t)i * 3 * kE * kE;
        L.qkv_b = m->ar_qkv_b + (int64_t)i * 3 * kE;     // key has no bias (app/transformer.py:61): stays zero
        add_slot(m, p + ".attn.query.weight", {kE, kE}, SK_DIRECT, L.qkv_w);
        add_slot(m, p + ".attn.query.bias", {kE}, SK_DIRECT, L.qkv_b);
        add_slot(m, p + ".attn.key.weight", {kE, kE}, SK_DIRECT, L.qkv_w + (int64_t)kE * kE);
        add_slot(m, p + ".attn.value.weight", {kE, kE}, SK_DIRECT, L.qkv_w + (int64_t)2 * kE * kE);
        add_slot(m, p + ".attn.value.bias", {kE}, SK_DIRECT, L.qkv_b + 2 * kE);
        add_linear(m, p + ".attn.proj", kE, kE, L.proj_w, L.proj_b);
        add_linear(m, p + ".ffn.0", 4 * kE, kE, L.ffn1_w, L.ffn1_b);
        add_linear(m, p + ".ffn.2", kE, 4 * kE, L.ffn2_w, L.ffn2_b);
        add_slot(m, p + ".ada_lin.1.weight", {6 * kE, kCond}, SK_DIRECT, m->ada_w + (int64_t)i * 6 * kE * kCond);
        add_slot(m, p + ".ada_lin.1.bias", {6 * kE}, SK_DIRECT, m->ada_b + i * 6 * kE);
    }
    // ---- VAE (app/modules/bitwise_vae.py) ----
    const int H = c.vae_hidden, T2 = 2 * c.patch_nums[c.n_levels - 1], MD = c.motion_dim;
    add_slot(m, "basic_vae.enc_pos_embed", {1, T2, MD}, SK_DIRECT, m->enc_pos = walloc(m, (int64_t)T2 * MD));
    add_slot(m, "basic_vae.dec_pos_embed", {1, T2, c.code_dim}, SK_DIRECT, m->dec_pos = walloc(m, (int64_t)T2 * c.code_dim));
    add_slot(m, "basic_vae.attn_mask", {1, 1, T2, T2}, SK_HOST, nullptr);
    add_vec(m, "basic_vae.motion_mean", MD, m->vae_mean);
    add_vec(m, "basic_vae.motion_std", MD, m->vae_std);
    m->enc.in_w = walloc(m, (int64_t)H * 128);
    add_slot(m, "basic_vae.encoder.inp_mapping.0.weight", {H, MD}, SK_PADK, m->enc.in_w, 128);
    add_vec(m, "basic_vae.encoder.inp_mapping.0.bias", H, m->enc.in_b);
    add_linear(m, "basic_vae.encoder.code_mapping", c.code_dim, H, m->enc.out_w, m->enc.out_b);
    add_linear(m, "basic_vae.decoder.inp_mapping.0", H, c.code_dim, m->dec.in_w, m->dec.in_b);
    add_linear(m, "basic_vae.decoder.out_mapping", MD, H, m->dec.out_w, m->dec.out_b);
    for (int side = 0; side < 2; ++side) {
        VAESide& S = side == 0 ? m->enc : m->dec;
        const std::string base = side == 0 ? "basic_vae.encoder.encoder_transformer." : "basic_vae.decoder.decoder_transformer.";
        S.layers.resize(c.vae_depth);
        for (int i = 0; i < c.vae_depth; ++i) {
            VAELayer& L = S.layers[i];
            const std::string a = base + std::to_string(2 * i), f = base + std::to_string(2 * i + 1);
            add_ln(m, a + ".norm", H, L.lnw, L.lnb);
            float* nb = nullptr;
            add_linear(m, a + ".to_qkv", 3 * H, H, L.qkv_w, nb, false);
            add_linear(m, a + ".to_out", H, H, L.out_w, L.out_b);
            add_linear(m, f + ".0", H * 3 / 2, H, L.m1_w, L.m1_b);
            add_linear(m, f + ".2", H, H * 3 / 2, L.m2_w, L.m2_b);
        }
    }
    // ---- style encoder (app/modules/style_encoder.py) ----
    const int S = c.style_dim;
    add_vec(m, "style_encoder.motion_mean", MD, m->st_mean);
    add_vec(m, "style_encoder.motion_std", MD, m->st_std);
    add_slot(m, "style_encoder.PE.pe", {1, 600, S}, SK_HOST, nullptr);
    m->st_pe = walloc(m, S);
    m->st_proj_w = walloc(m, (int64_t)S * 128);
    add_slot(m, "style_encoder.encoder.motion_proj.weight", {S, MD}, SK_PADK, m->st_proj_w, 128);
    add_vec(m, "style_encoder.encoder.motion_proj.bias", S, m->st_proj_b);
    m->style.resize(c.style_layers);
    for (int i = 0; i < c.style_layers; ++i) {
        StyleLayer& L = m->style[i];
        const std::string p = "style_encoder.encoder.transformer.layers." + std::to_string(i);
        L.in_w = walloc(m, (int64_t)3 * S * S); L.in_b = walloc(m, 3 * S);
        add_slot(m, p + ".self_attn.in_proj_weight", {3 * S, S}, SK_DIRECT, L.in_w);
        add_slot(m, p + ".self_attn.in_proj_bias", {3 * S}, SK_DIRECT, L.in_b);
        add_linear(m, p + ".self_attn.out_proj", S, S, L.out_w, L.out_b);
        add_linear(m, p + ".linear1", c.style_ffn, S, L.l1_w, L.l1_b);
        add_linear(m, p + ".linear2", S, c.style_ffn, L.l2_w, L.l2_b);
        add_ln(m, p + ".norm1", S, L.n1w, L.n1b);
        add_ln(m, p + ".norm2", S, L.n2w, L.n2b);
    }
    // ---- wav2vec2 ----
    const int Hs = c.w2v_hidden, Fi = c.w2v_ffn, CD = c.w2v_conv_dim;
    add_slot(m, "audio_encoder.masked_spec_embed", {Hs}, SK_IGNORE, nullptr);   // dead in inference
    int cin = 1;
    for (int i = 0; i < c.w2v_n_conv; ++i) {
        const std::string p = "audio_encoder.feature_extractor.conv_layers." + std::to_string(i);
        const int k = c.w2v_conv_kernel[i];
        m->conv_w[i] = walloc(m, (int64_t)CD * cin * k);
        add_slot(m, p + ".conv.weight", {CD, cin, k}, i == 0 ? SK_DIRECT : SK_CONV, m->conv_w[i]);
        add_vec(m, p + ".conv.bias", CD, m->conv_b[i]);
        add_ln(m, p + ".layer_norm", CD, m->conv_lnw[i], m->conv_lnb[i]);
        cin = CD;
    }
    m->conv0_w = m->conv_w[0];
    add_ln(m, "audio_encoder.feature_projection.layer_norm", CD, m->fp_lnw, m->fp_lnb);
    add_linear(m, "audio_encoder.feature_projection.projection", Hs, CD, m->fp_w, m->fp_b);
    const std::string pc = "audio_encoder.encoder.pos_conv_embed.conv";
    add_vec(m, pc + ".bias", Hs, m->pos_b);
    add_slot(m, pc + ".parametrizations.weight.original0", {1, 1, c.w2v_pos_kernel}, SK_HOST, nullptr);
    add_slot(m, pc + ".parametrizations.weight.original1", {Hs, Hs / c.w2v_pos_groups, c.w2v_pos_kernel}, SK_HOST, nullptr);
    m->pos_w = walloc(m, (int64_t)Hs * (Hs / c.w2v_pos_groups) * c.w2v_pos_kernel);
    add_ln(m, "audio_encoder.encoder.layer_norm", Hs, m->enc_lnw, m->enc_lnb);
    m->w2v.resize(c.w2v_layers);
    for (int i = 0; i < c.w2v_layers; ++i) {
        W2VLayer& L = m->w2v[i];
        const std::string p = "audio_encoder.encoder.layers." + std::to_string(i);
        L.qkv_w = walloc(m, (int64_t)3 * Hs * Hs); L.qkv_b = walloc(m, 3 * Hs);
        const char* names[3] = {"q_proj", "k_proj", "v_proj"};
        for (int j = 0; j < 3; ++j) {
            add_slot(m, p + ".attention." + names[j] + ".weight", {Hs, Hs}, SK_DIRECT, L.qkv_w + (int64_t)j * Hs * Hs);
            add_slot(m, p + ".attention." + names[j] + ".bias", {Hs}, SK_DIRECT, L.qkv_b + j * Hs);
        }
        add_linear(m, p + ".attention.out_proj", Hs, Hs, L.out_w, L.out_b);
        add_ln(m, p + ".layer_norm", Hs, L.ln1w, L.ln1b);
        add_ln(m, p + ".final_layer_norm", Hs, L.ln2w, L.ln2b);
        add_linear(m, p + ".feed_forward.intermediate_dense", Fi, Hs, L.ff1_w, L.ff1_b);
        add_linear(m, p + ".feed_forward.output_dense", Hs, Fi, L.ff2_w, L.ff2_b);
    }
    for (void* p : m->allocs) if (!p) return fail(m, ARTALK_EHIP, "hipMalloc failed while allocating weights");
    return ARTALK_OK;
}

int upload(artalk_model* m, float* dst, const float* src, int64_t n) {
    HIPCHK(m, hipMemcpy(dst, src, n * sizeof(float), hipMemcpyHostToDevice));
    return ARTALK_OK;
}

// ------------------------------------------------------------------------------------------------ profiling
hipEvent_t next_event(artalk_model* m, hipStream_t s, size_t* idx) {
    if (m->ev_used == m->ev_pool.size()) {
        hipEvent_t e;
        (void)hipEventCreate(&e);
        m->ev_pool.push_back(e);
    }
    *idx = m->ev_used++;
    (void)hipEventRecord(m->ev_pool[*idx], s);
    return m->ev_pool[*idx];
}
// profile buckets (artalk_get_profile): the interval that ENDS at a mark is charged to the mark's bucket
enum { PB_STYLE = 0, PB_CONV = 1, PB_ENC = 2, PB_ADA = 3, PB_AR = 4, PB_VAE = 5, PB_OTHER = 6 };
void kbucket(artalk_model* m, int b) { m->ktimer.bucket = b; }
void stage_mark(artalk_model* m, hipStream_t s, int bucket) {
    if (!m->profiling || m->in_graph_body) return;
    size_t i;
    next_event(m, s, &i);
    m->marks.emplace_back(bucket, i);
}

// fuse_ln: the AdaLN-modulated LayerNorm that consumes this GEMM's (768-wide, residual-stream) result.  If the GEMM is split
// over K, its reduce pass also writes that LayerNorm's output (one launch instead of two) and gemm() returns true; otherwise
// the caller launches the LayerNorm itself.
constexpr int kAuditSlots = 1024;
// audit hook: records max |x| of a just-produced P8 buffer (or of the fp32 buffer that holds the same activation in f32 mode, or that a
// register-staged GEMM will split).  ex = the site's exponent (a SiteExps member of this model): what the buffer was written with, and
// what artalk_calibrate adjusts from the recorded maximum.  Works in both precision modes: the calibration pass runs in exact-f32
// mode, where nothing can overflow on the way to a later site.
void audit(artalk_model* m, const char* site, int idx0, const char* sub, const float* buf, int rows, int cols, long ld, bool is_p8, hipStream_t s,
           int* ex, int junk_period = 0, int junk_from = 0) {
    if (!m->audit || !m->audit_vals) return;
    std::string name = site;
    if (idx0 >= 0) name += std::to_string(idx0);
    if (sub) name += sub;
    auto it = m->audit_index.find(name);
    int idx;
    if (it == m->audit_index.end()) {
        if ((int)m->audit_names.size() >= kAuditSlots) return;
        idx = (int)m->audit_names.size();
        m->audit_names.push_back(name);
        m->audit_exp.push_back(ex);
        m->audit_index.emplace(name, idx);
    } else idx = it->second;
    launch_absmax(buf, rows, cols & ~7, ld, is_p8 ? 1 : 0, m->audit_vals + idx, s, junk_period, junk_from, ex ? *ex : kActExp);
}

// defer_reduce: the caller's next kernel sums the split-K slabs itself (the short-query attention kernel does, for the q|k|v rows of
// its own head): if this GEMM is split, its reduce pass is NOT launched and *defer_reduce receives the slab count (else 0).
bool gemm(artalk_model* m, const GemmArgs& g0, hipStream_t s, const LnArgs* fuse_ln = nullptr, int* defer_reduce = nullptr) {
    GemmArgs g = g0;
    g.graph_tag = m->in_body ? 1 : 0;
    g.cus = m->n_cus;
    g.status = m->precision == 1 ? (m->view ? m->view->status : m->ws.status) : nullptr;     // P8 range guard at the producers
    bool split = false;
    if (m->precision == 1 && !g.exact) {
        g.Wp = packed_of(m, g.W);
        split = gemm_f16s_eligible(g);
    }
    // split-K for grids that would leave most CUs idle (small-M scale steps): S workgroups per output tile
    const Workspace& cw = m->view ? *m->view : m->ws;
    const bool sm_path = split && g.a_packed && gemm_p8_sm_eligible(g) && !gemm_p8_eligible(g);
    if (g.batch == 1 && g.amode == 0 && cw.splitk && g.K >= 256) {
        const int tiles = gemm_tile_count(g, split);
        const int lim = m->splitk_tiles, tgt = m->splitk_target;
        if (sm_path) {
            // Small-grid LDS-DMA kernel (64x64 tiles).  These launches are latency-bound: measured with cold weights, replayed from a
            // graph, GEMM + reduce (profiles/r02_tiny_gemm_sweep.log): K = 3072 gains from a split below ~192 tiles (M = 400: 19.1 us
            // split in 6 vs 27 unsplit); K = 768 gains only on the smallest grids, where the split workgroups' whole K slice fits the
            // ring and is in flight at once (deep-ring configurations 23 / 24): proj at M = 80 7.8 us split in 6 vs 9.8, qkv 9.9 us
            // split in 3 vs 13.2.  A split 768-wide result also gets its LayerNorm for free (fuse_ln).
            int S = 1, cfg = -1;
            if (g.K >= 2048) {
                if (tiles <= 24) { S = 8; cfg = 23; }
                else if (tiles < 192) S = tiles < 48 ? 6 : 3;
            } else {
                if (tiles <= 24) { S = 6; cfg = 24; }
                else if (tiles <= 36) { S = 4; cfg = 24; }
                else if (tiles <= 72) { S = 3; cfg = 23; }
                else if (tiles <= 108) S = 2;
            }
            while (S > 1 && (int64_t)S * g.M * g.N > cw.splitk_floats) { --S; cfg = -1; }
            if (S == 5 || S == 7) { --S; }        // the unrolled reduce kernels exist for 2, 3, 4, 6, 8 slabs
            // Mid-grid kernel (128x128 tiles, one 8-wave workgroup per CU, 4-stage ring: gemm_p8_mid_kernel, cfg 28) where its grid
            // fills most of the chip: these launches are bound by what an XCD pulls in per K step, and a 128x128 tile has four times
            // the matrix work per fetched byte of the 64x64 tile (profiles/r03_mid_gemm_sweep.log, cold weights, graph replay:
            // q|k|v at M = 1600 30.7 -> 24.6 us, FFN-in at M = 800 28.8 -> 22.2, FFN-out at M = 1600 split in 3 42.6 -> 35.0; below
            // ~150 tiles the 64x64 kernel wins: projection at M = 1600 14.0 vs 19.0).  It has no second P8 copy of the result (c2).
            if (S == 1 && g.force_cfg < 0 && !g.c2) {
                const int t128 = ((g.M + 127) / 128) * ((g.N + 127) / 128);
                // Ping-pong kernel (gemm_p8_pp_kernel, cfg 31: 256 x 128 tiles, the two waves of a SIMD one phase apart) where its grid is
                // 120 .. 256 tiles: q|k|v and FFN-in of the 100-token step, q|k|v of the VAE decoder stack.  Measured (round 5,
                // profiles/r05_pp_gemm_sweep.log, r05_pp_model_ab.log): FFN-in at M = 1600 39.0 -> 32.6 us (168 tiles in ONE round instead
                // of 312 in two), VAE q|k|v at M = 3200 27.1 -> 24.2, q|k|v at M = 1600 slower alone (24.3 -> 30.8: half the chip) but not
                // beside the other clip group's launches; body 31.75 / 31.92 -> 31.58 / 31.79 ms per step in same-box pairs.  The
                // 128 x 128 form (cfg 33) equals the mid-grid kernel on every shape.  ARTALK_PP=0 switches it off (A/B), 2 / 3 add cfg 33.
                static const int pp_mode = getenv("ARTALK_PP") ? atoi(getenv("ARTALK_PP")) : 1;
                static const int pp_min = getenv("ARTALK_PP_MIN") ? atoi(getenv("ARTALK_PP_MIN")) : 120;
                const int t256 = ((g.M + 255) / 256) * ((g.N + 127) / 128);
                if ((pp_mode & 1) && t256 >= pp_min && t256 <= 256 && gemm_p8_pp_ok(g)) cfg = 31;
                else if (t128 >= 150) cfg = ((pp_mode & 2) && gemm_p8_pp_ok(g)) ? 33 : 28;
                if (cfg == 31 || cfg == 33) g.force_cfg = cfg;
                else if (t128 < 150 && g.K >= 2048 && 3 * t128 >= 150 && (int64_t)3 * g.M * g.N <= cw.splitk_floats) { S = 3; cfg = 28; }
                if (cfg == 28 && S == 1) g.force_cfg = 28;
            }
            if (S > 1) { g.splitk = S; g.partial = cw.splitk; if (g.force_cfg < 0) g.force_cfg = cfg; }
            // weights of the narrowest steps (levels 0 and 1) are read once per launch by a handful of workgroups: fetched with the
            // non-temporal policy they do not displace the resident history (A/B, cold weights: -0.2..-0.5 us of 8..16 us at M <= 80,
            // +0.4 us at M = 400, where several row tiles re-read each weight tile through L2)
            if (g.M <= 160) g.w_nt = 1;
        } else if (tiles < lim) {
            int S = std::min(std::min(g.K / 64, (tgt + tiles - 1) / tiles), 16);
            while (S > 1 && (int64_t)S * g.M * g.N > cw.splitk_floats) --S;
            if (S > 1) { g.splitk = S; g.partial = cw.splitk; }
        }
    }
    const bool dominant = !m->in_body && g.M > 0 &&
                          (split ? (g.a_packed ? (gemm_p8_eligible(g) && gemm_p8_variant(g) >= 1) : gemm_f16s_config(g) == 0) : gemm_config(g) == 4);
    size_t i0 = 0, i1 = 0;
    if (m->profiling && dominant) next_event(m, s, &i0);
    if (g.a_packed && !split) { m->err = "internal: P8 activation handed to an fp32 GEMM"; m->sticky_error = true; return false; }
    const bool dma = split && g.splitk == 1 && gemm_p8_eligible(g);
    // second copy of the result in P8 (GemmArgs::c2): written by the small-grid kernel's epilogue when that kernel finishes the tiles
    // itself, otherwise by a split pass over the fp32 result
    float* const c2 = g.c2;
    const bool c2_fused = c2 && !dma && split && gemm_p8_sm_eligible(g) && g.splitk == 1 && !g.c_p8 && (g.N % 8) == 0 && (g.ldc % 8) == 0 &&
                          !g.gate && (g.ldr % 4) == 0 &&      // (the epilogue's 16-byte path: every pointer and row start aligned)
                          (((unsigned long long)g.C | (unsigned long long)c2 | (unsigned long long)g.bias | (unsigned long long)g.R) & 15) == 0;
    if (!c2_fused) g.c2 = nullptr;
    if (dma) launch_gemm_p8(g, s);
    else if (split && gemm_p8_sm_eligible(g)) launch_gemm_p8_sm(g, s);     // P8 activation, small grid (AR/VAE scale steps)
    else if (split) launch_gemm_f16s(g, s);
    else launch_gemm(g, s);
    bool fused = false;
    if (defer_reduce) *defer_reduce = 0;
    if (g.splitk > 1) {
        if (fuse_ln && splitk_reduce_ln_eligible(g, *fuse_ln)) { launch_splitk_reduce_ln(g, *fuse_ln, s); fused = true; }
        else if (defer_reduce && !c2) *defer_reduce = g.splitk;
        else launch_splitk_reduce(g, s);
    }
    if (c2 && !c2_fused) launch_pack_split(g.C, reinterpret_cast<unsigned int*>(c2), (long)g.M * g.N, false, s, g.status, g.c_exp);
    if (m->profiling && dominant) {
        next_event(m, s, &i1);
        m->dom_events.emplace_back(i0, gemm_flops(g));
    }
    return fused;
}

// ---- intermediate taps (include/artalk_hip.h: artalk_set_tap) ----
// slot (chunk index j, clip position b) of the tap buffer, fields in floats:
enum { TAP_BLK0_IN = 0, TAP_BLK0_OUT = kNTok * kE, TAP_BLKL_OUT = 2 * kNTok * kE, TAP_PREV_IN = 3 * kNTok * kE,
       TAP_LOGITS = 4 * kNTok * kE, TAP_DEC_OUT = 4 * kNTok * kE + kNTok * 64, TAP_SLOT = 4 * kNTok * kE + kNTok * 64 + 200 * 106 };
// copies `rows` x `cols` fp32 values per clip (compact source [B][rows][cols]) to row `row0` of field `field` of the chunk's slots
void tap_copy(artalk_model* m, int field, int row0, const float* src, int rows, int cols, int B, hipStream_t s) {
    if (!m->tap || m->in_graph_body || m->tap_chunk >= m->tap_maxch || B > m->tap_B) return;
    float* dst = m->tap + ((long)m->tap_chunk * m->tap_B) * TAP_SLOT + field + (long)row0 * cols;
    (void)hipMemcpy2DAsync(dst, (size_t)TAP_SLOT * 4, src, (size_t)rows * cols * 4, (size_t)rows * cols * 4, B, hipMemcpyDeviceToDevice, s);
}

enum { LF_EXACT = 1, LF_A_P8 = 2, LF_C_P8 = 4 };   // linear() flags: decision-critical (fp32 path) / A is in P8 / write C in P8
// plain y = act(x W^T + b) [+ R]
void linear(artalk_model* m, const float* A, long lda, const float* W, const float* bias, float* C, long ldc, int M, int N, int K,
            int act, const float* R, hipStream_t s, int flags = 0, float* c2 = nullptr, int a_exp = kActExp, int c_exp = kActExp) {
    GemmArgs g;
    g.c2 = c2; g.a_exp = a_exp; g.c_exp = c_exp;
    g.exact = flags & LF_EXACT; g.a_packed = (flags & LF_A_P8) ? 1 : 0; g.c_p8 = (flags & LF_C_P8) ? 1 : 0;
    g.A = A; g.lda = lda; g.W = W; g.ldw = K; g.bias = bias; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.act = act;
    g.R = R; g.ldr = ldc;
    gemm(m, g, s);
}

void layernorm(const float* X, float* Y, const float* w, const float* b, int M, int D, float eps, int act, hipStream_t s, int out_p8 = 0,
               int* status = nullptr, int junk_period = 0, int junk_from = 0, int p8_exp = kActExp) {
    LnArgs a;
    a.out_p8 = out_p8; a.p8_exp = p8_exp; a.status = out_p8 ? status : nullptr; a.junk_period = junk_period; a.junk_from = junk_from;
    a.X = X; a.ldx = D; a.Y = Y; a.ldy = D; a.w = w; a.b = b; a.M = M; a.D = D; a.eps = eps; a.act = act;
    launch_layernorm(a, s);
}

// ------------------------------------------------------------------------------------------------ stages
// wav2vec2 on chunks [c0, c0+n) of the chunk list (app/modules/wav2vec.py:11-20); writes SiLU(pooled cond) rows c0*181..
void run_wav2vec(artalk_model* m, const float* audio, int c0, int n, float* out_w2v, hipStream_t s) {
    const artalk_config& c = m->cfg;
    Workspace& w = m->ws;
    const int CD = c.w2v_conv_dim, Hs = c.w2v_hidden;
    // f16x3 mode: every activation whose only consumer is a GEMM is written by its producer directly in the P8 split
    // format (same bytes), so the big GEMMs can stage both operands with LDS-DMA (gemm_p8_2wgp_kernel / gemm_p8_256_kernel).
    const int p8 = m->precision == 1 ? 1 : 0;
    const int AP = p8 ? LF_A_P8 : 0;
    Range r_w2v("artalk.wav2vec2");
    roctxRangePushA("artalk.wav2vec2.conv_stack");      // K1-K3: normalise, conv0+LN+GELU, conv1-6 as GEMMs + LN + GELU
    kbucket(m, KB_CONV);
    launch_audio_normalize(audio, w.src_off + c0, w.xnorm, n, kSamplesPerChunk, s);
    SiteExps& ex = m->ex;
    launch_conv0(w.xnorm, kSamplesPerChunk, m->conv0_w, m->conv_b[0], m->conv_lnw[0], m->conv_lnb[0], w.convA, n, m->conv_T[0],
                 m->conv_S[0], s, p8, w.status, ex.conv[0]);
    audit(m, "w2v.conv", 0, ".ln_gelu", w.convA, n * m->conv_S[0], CD, CD, p8, s, &ex.conv[0], m->conv_S[0], m->conv_T[0]);
    float* src = w.convA; float* dst = w.convB;
    for (int i = 1; i < c.w2v_n_conv; ++i) {
        // stride-2 conv as a GEMM: output row r reads input rows 2r..2r+k-1 (contiguous K = k*512 floats)
        const int M = n * m->conv_S[i];
        linear(m, src, (long)c.w2v_conv_stride[i] * CD, m->conv_w[i], m->conv_b[i], dst, CD, M, CD, c.w2v_conv_kernel[i] * CD,
               ACT_NONE, nullptr, s, AP, nullptr, ex.conv[i - 1]);
        // the last conv output feeds a LayerNorm (feature projection), not a GEMM: it stays fp32
        // rows t >= conv_T[i] of every chunk are layout padding (computed from the padding rows below them): no range guard there
        layernorm(dst, dst, m->conv_lnw[i], m->conv_lnb[i], M, CD, 1e-5f, ACT_GELU_ERF, s, (p8 && i + 1 < c.w2v_n_conv) ? 1 : 0, w.status,
                  m->conv_S[i], m->conv_T[i], ex.conv[i]);
        if (i + 1 < c.w2v_n_conv) audit(m, "w2v.conv", i, ".ln_gelu", dst, M, CD, CD, p8, s, &ex.conv[i], m->conv_S[i], m->conv_T[i]);
        std::swap(src, dst);
    }
    stage_mark(m, s, PB_CONV);
    roctxRangePop();
    Range r_enc("artalk.wav2vec2.encoder");             // K4-K7: projection, pos-conv, 24 layers, final LN, pooling + SiLU
    kbucket(m, KB_ENC);
    const int M = n * m->Ts;
    // hidden states use a 200-row stride per chunk, 199 valid.  The padding row is never read by a valid row; the LayerNorms store
    // it as zeros (LnArgs::junk_period), and the attention output's padding row is never written (zero since allocation), so
    // everything computed in it stays finite and inside the P8 range without the GEMM epilogues having to know about it.
    const int JP = m->Ts, JF = m->Tw;
    // feature projection (hf:429-434)
    layernorm(src, dst, m->fp_lnw, m->fp_lnb, M, CD, c.w2v_ln_eps, ACT_NONE, s, p8, w.status, JP, JF, ex.fp_ln);
    audit(m, "w2v.feature_projection.ln", -1, nullptr, dst, M, CD, CD, p8, s, &ex.fp_ln, JP, JF);
    linear(m, dst, CD, m->fp_w, m->fp_b, w.h0, Hs, M, Hs, CD, ACT_NONE, nullptr, s, AP, nullptr, ex.fp_ln);
    // positional conv embedding (hf:360-368): h1 = h0 + gelu(groupconv(h0) + b)
    audit(m, "w2v.posconv.input(fp32 A)", -1, nullptr, w.h0, M, Hs, Hs, false, s, &ex.posconv_in, JP, JF);
    {
        const int cg = Hs / c.w2v_pos_groups;
        GemmArgs g;
        g.A = w.h0; g.lda = Hs; g.W = m->pos_w; g.ldw = (long)cg * c.w2v_pos_kernel; g.bias = m->pos_b; g.C = w.h1; g.ldc = Hs; g.R = w.h0; g.ldr = Hs;
        g.M = M; g.act = ACT_GELU_ERF; g.a_exp = ex.posconv_in;
        if (p8 && cg == 64 && c.w2v_pos_kernel == 128 && c.w2v_pos_groups == 16 && m->Ts <= 256) {
            // f16x3 mode: one workgroup per (chunk, group) with the chunk's input window resident in LDS (posconv_p8_kernel)
            g.N = Hs; g.K = cg * c.w2v_pos_kernel; g.Wp = packed_of(m, m->pos_w); g.status = w.status;
            launch_posconv_p8(g, n, m->Tw, m->Ts, s);
        } else {
            // grouped GEMM over grid.z whose A operand gathers the zero-padded window (amode 1)
            g.amode = 1; g.pc_T = m->Tw; g.pc_tstride = m->Ts; g.pc_pad = c.w2v_pos_kernel / 2; g.pc_cin = cg;
            g.N = cg; g.K = cg * c.w2v_pos_kernel;
            g.batch = c.w2v_pos_groups; g.sA = cg; g.sW = (long)cg * cg * c.w2v_pos_kernel; g.sBias = cg; g.sC = cg; g.sR = cg;
            gemm(m, g, s);
        }
    }
    float* h = w.h1;
    const int nh = c.w2v_heads, hd = Hs / nh;
    for (int i = 0; i < c.w2v_layers; ++i) {
        const W2VLayer& L = m->w2v[i];
        SiteExps::W2V& E = ex.w2v[i];
        layernorm(h, w.xln, L.ln1w, L.ln1b, M, Hs, c.w2v_ln_eps, ACT_NONE, s, p8, w.status, JP, JF, E.ln1);
        audit(m, "w2v.layer", i, ".ln1", w.xln, M, Hs, Hs, p8, s, &E.ln1, JP, JF);
        linear(m, w.xln, Hs, L.qkv_w, L.qkv_b, w.qkv, 3 * Hs, M, 3 * Hs, Hs, ACT_NONE, nullptr, s, AP | (p8 ? LF_C_P8 : 0), nullptr, E.ln1, E.qkv);   // q, k, v leave in P8
        AttnArgs a;
        a.qkv_p8 = p8; a.qkv_exp = E.qkv; a.o_exp = E.attn;
        a.Q = w.qkv; a.K = w.qkv + Hs; a.V = w.qkv + 2 * Hs;
        a.ldq = a.ldk = a.ldv = 3 * Hs; a.q_bstride = a.k_bstride = a.v_bstride = (long)m->Ts * 3 * Hs;
        a.O = w.att; a.ldo = Hs; a.o_bstride = (long)m->Ts * Hs;
        a.B = n; a.H = nh; a.HD = hd; a.Lq = m->Tw; a.Lk = m->Tw; a.scale = 1.0f / std::sqrt((float)hd);
        audit(m, "w2v.layer", i, ".qkv", w.qkv, M, 3 * Hs, 3 * Hs, p8, s, &E.qkv, JP, JF);
        a.out_p8 = p8; a.split16 = p8; a.status = p8 ? w.status : nullptr; a.cus = m->n_cus;
        launch_attention(a, s);
        audit(m, "w2v.layer", i, ".attn_out", w.att, M, Hs, Hs, p8, s, &E.attn, JP, JF);
        linear(m, w.att, Hs, L.out_w, L.out_b, h, Hs, M, Hs, Hs, ACT_NONE, h, s, AP, nullptr, E.attn);
        layernorm(h, w.xln, L.ln2w, L.ln2b, M, Hs, c.w2v_ln_eps, ACT_NONE, s, p8, w.status, JP, JF, E.ln2);
        audit(m, "w2v.layer", i, ".ln2", w.xln, M, Hs, Hs, p8, s, &E.ln2, JP, JF);
        linear(m, w.xln, Hs, L.ff1_w, L.ff1_b, w.ffn, c.w2v_ffn, M, c.w2v_ffn, Hs, ACT_GELU_ERF, nullptr, s, AP | (p8 ? LF_C_P8 : 0), nullptr, E.ln2, E.ffn);
        audit(m, "w2v.layer", i, ".ffn_hidden", w.ffn, M, c.w2v_ffn, c.w2v_ffn, p8, s, &E.ffn, JP, JF);
        linear(m, w.ffn, c.w2v_ffn, L.ff2_w, L.ff2_b, h, Hs, M, Hs, c.w2v_ffn, ACT_NONE, h, s, AP, nullptr, E.ffn);
    }
    layernorm(h, w.xln, m->enc_lnw, m->enc_lnb, M, Hs, c.w2v_ln_eps, ACT_NONE, s);
    if (out_w2v)
        (void)hipMemcpy2DAsync(out_w2v + (long)c0 * m->Tw * Hs, (size_t)m->Tw * Hs * 4, w.xln, (size_t)m->Ts * Hs * 4,
                               (size_t)m->Tw * Hs * 4, n, hipMemcpyDeviceToDevice, s);
    launch_pool_silu(w.xln, m->Ts, m->Tw, w.silu_cond + (long)c0 * kNTok * kCond, n, m->pn, c.n_levels, kCond, s, p8, w.status, ex.silu_cond);
    audit(m, "ar.silu_cond", -1, nullptr, w.silu_cond + (long)c0 * kNTok * kCond, n * kNTok, kCond, kCond, p8, s, &ex.silu_cond);
    stage_mark(m, s, PB_ENC);
}

// style_motion rows: a (50,106) clip where the host flag is 1, a cached 768-float condition where it is 2 (artalk_style_encode);
// `encode` = some row has flag 1 (the host knows: has_style is a host array), otherwise the encoder stack is skipped altogether
void run_style(artalk_model* m, const float* style_motion, int B, hipStream_t s, bool encode = true) {
    const artalk_config& c = m->cfg;
    Workspace& w = m->ws;
    const int S = c.style_dim, L = c.style_len, M = B * L;
    Range r_style("artalk.style_encoder");
    kbucket(m, KB_STYLE);
    if (style_motion && encode) {
        launch_style_input(style_motion, m->st_mean, m->st_std, w.s_in, B, s);
        SiteExps& ex = m->ex;
        audit(m, "style.input(fp32 A)", -1, nullptr, w.s_in, M, 128, 128, false, s, &ex.style_in);
        linear(m, w.s_in, 128, m->st_proj_w, m->st_proj_b, w.s_h, S, M, S, 128, ACT_NONE, nullptr, s, 0, nullptr, ex.style_in);
        launch_add_row(w.s_h, m->st_pe, M, S, s);   // PositionalEncoding quirk: pe[:, seq_len] added to every token
        for (int i = 0; i < c.style_layers; ++i) {
            const StyleLayer& Ly = m->style[i];
            SiteExps::ST& E = ex.style[i];
            audit(m, "style.layer", i, ".in(fp32 A)", w.s_h, M, S, S, false, s, &E.h);
            linear(m, w.s_h, S, Ly.in_w, Ly.in_b, w.s_qkv, 3 * S, M, 3 * S, S, ACT_NONE, nullptr, s, 0, nullptr, E.h);
            AttnArgs a;
            a.Q = w.s_qkv; a.K = w.s_qkv + S; a.V = w.s_qkv + 2 * S; a.ldq = a.ldk = a.ldv = 3 * S;
            a.q_bstride = a.k_bstride = a.v_bstride = (long)L * 3 * S;
            a.O = w.s_att; a.ldo = S; a.o_bstride = (long)L * S;
            a.B = B; a.H = c.style_heads; a.HD = S / c.style_heads; a.Lq = L; a.Lk = L;
            a.scale = 1.0f / std::sqrt((float)(S / c.style_heads));
            launch_attention(a, s);
            audit(m, "style.layer", i, ".attn_out(fp32 A)", w.s_att, M, S, S, false, s, &E.attn);
            linear(m, w.s_att, S, Ly.out_w, Ly.out_b, w.s_tmp, S, M, S, S, ACT_NONE, w.s_h, s, 0, nullptr, E.attn);
            layernorm(w.s_tmp, w.s_h, Ly.n1w, Ly.n1b, M, S, 1e-5f, ACT_NONE, s);
            audit(m, "style.layer", i, ".norm1(fp32 A)", w.s_h, M, S, S, false, s, &E.h1);
            linear(m, w.s_h, S, Ly.l1_w, Ly.l1_b, w.s_ffn, c.style_ffn, M, c.style_ffn, S, ACT_GELU_ERF, nullptr, s, 0, nullptr, E.h1);
            audit(m, "style.layer", i, ".ffn_hidden(fp32 A)", w.s_ffn, M, c.style_ffn, c.style_ffn, false, s, &E.ffn);
            linear(m, w.s_ffn, c.style_ffn, Ly.l2_w, Ly.l2_b, w.s_tmp, S, M, S, c.style_ffn, ACT_NONE, w.s_h, s, 0, nullptr, E.ffn);
            layernorm(w.s_tmp, w.s_h, Ly.n2w, Ly.n2b, M, S, 1e-5f, ACT_NONE, s);
        }
    }
    launch_style_finish(w.s_h, m->sc_w, m->sc_b, m->null_style, style_motion ? w.has_style : nullptr, w.style_cond, B, s,
                        style_motion, (long)L * c.motion_dim);
}

// one VAE transformer stack (app/modules/bitwise_vae.py:149-157 / :183-191) on B sequences of T tokens, in place on ws.vh
void run_vae_stack(artalk_model* m, const VAESide& S, int B, int T, int split, hipStream_t s) {
    const artalk_config& c = m->cfg;
    Workspace& w = m->view ? *m->view : m->ws;
    const int H = c.vae_hidden, M = B * T, F = H * 3 / 2;
    const int p8 = m->precision == 1 ? 1 : 0;      // GEMM-only activations in the P8 split format (see run_chunk_body)
    const int AP = p8 ? LF_A_P8 : 0;
    const int sd = &S == &m->enc ? 0 : 1;
    const char* const an = sd == 0 ? "vae.encoder.layer" : "vae.decoder.layer";
    for (int i = 0; i < c.vae_depth; ++i) {
        const VAELayer& L = S.layers[i];
        SiteExps::VAE& E = m->ex.vae[sd][i];
        layernorm(w.vh, w.vln, L.lnw, L.lnb, M, H, 1e-5f, ACT_NONE, s, p8, w.status, 0, 0, E.ln);
        audit(m, an, i, ".ln", w.vln, M, H, H, p8, s, &E.ln);
        linear(m, w.vln, H, L.qkv_w, nullptr, w.vqkv, 3 * H, M, 3 * H, H, ACT_NONE, nullptr, s, AP | (p8 ? LF_C_P8 : 0), nullptr, E.ln, E.qkv);   // q, k, v leave in P8
        AttnArgs a;
        a.qkv_p8 = p8; a.qkv_exp = E.qkv; a.o_exp = E.attn;
        a.Q = w.vqkv; a.K = w.vqkv + H; a.V = w.vqkv + 2 * H; a.ldq = a.ldk = a.ldv = 3 * H;
        a.q_bstride = a.k_bstride = a.v_bstride = (long)T * 3 * H;
        a.O = w.vatt; a.ldo = H; a.o_bstride = (long)T * H;
        a.B = B; a.H = c.vae_heads; a.HD = H / c.vae_heads; a.Lq = T; a.Lk = T;
        a.scale = 1.0f / std::sqrt((float)H);      // hidden_dim**-0.5, NOT head_dim (bitwise_vae.py:198)
        audit(m, an, i, ".qkv", w.vqkv, M, 3 * H, 3 * H, p8, s, &E.qkv);
        a.split_q = split; a.split_k = split; a.out_p8 = p8; a.split16 = p8; a.status = p8 ? w.status : nullptr; a.cus = m->n_cus;
        launch_attention(a, s);
        audit(m, an, i, ".attn_out", w.vatt, M, H, H, p8, s, &E.attn);
        // the MLP reads the residual stream itself (no LayerNorm in front of it, bitwise_vae.py:139-145).  In f16x3 mode the
        // out-projection writes it a second time in P8 (w.vln) so that the MLP GEMM can stage it by LDS-DMA (the register-staged
        // kernel on fp32 rows took 36 us at M = 6400, a split pass + the small-grid kernel 4 + 15, the second copy from the epilogue 15)
        linear(m, w.vatt, H, L.out_w, L.out_b, w.vh, H, M, H, H, ACT_NONE, w.vh, s, AP, p8 ? w.vln : nullptr, E.attn, E.resid);
        audit(m, an, i, ".residual", p8 ? w.vln : w.vh, M, H, H, p8, s, &E.resid);
        linear(m, p8 ? w.vln : w.vh, H, L.m1_w, L.m1_b, w.vmlp, F, M, F, H, ACT_GELU_TANH, nullptr, s, p8 ? (LF_A_P8 | LF_C_P8) : 0, nullptr, E.resid, E.mlp);
        audit(m, an, i, ".mlp_hidden", w.vmlp, M, F, F, p8, s, &E.mlp);
        linear(m, w.vmlp, F, L.m2_w, L.m2_b, w.vh, H, M, H, F, ACT_NONE, w.vh, s, AP, nullptr, E.mlp);
    }
}

// motion (already in ws.enc_in as normalised + pos-embedded rows) -> history bits, decoder features, prev tokens
void run_reencode(artalk_model* m, int B, hipStream_t s) {
    const artalk_config& c = m->cfg;
    Workspace& w = m->view ? *m->view : m->ws;
    const int H = c.vae_hidden, T = 100;
    Range r_re("artalk.vae.reencode_bsq");              // K16-K17: encoder, multi-scale BSQ, history features
    kbucket(m, KB_REENC);
    audit(m, "vae.encoder.input(fp32 A)", -1, nullptr, w.enc_in, B * T, 128, 128, false, s, &m->ex.vae_enc_in);
    linear(m, w.enc_in, 128, m->enc.in_w, m->enc.in_b, w.vh, H, B * T, H, 128, ACT_LEAKY02, nullptr, s, 0, nullptr, m->ex.vae_enc_in);
    run_vae_stack(m, m->enc, B, T, 0, s);
    linear(m, w.vh, H, m->enc.out_w, m->enc.out_b, w.enc_out, c.code_dim, B * T, c.code_dim, H, ACT_NONE, nullptr, s, LF_EXACT);
    launch_bsq_history(w.enc_out, w.hist_bits, w.prev_fdec, w.msfeat, B, s, w.status);
    launch_vq_embed(w.msfeat, kNTok - 1, m->vq_w, m->vq_b, m->prev_lvl_pos + kE, w.prev_in, kNTok, 1, w.style_cond,
                    m->prev_lvl_pos, B, s);
}

// Initial history of B clips (app/models.py:86-89) from the per-model cache (artalk_model::init_hist).
int run_init_history(artalk_model* m, int B, hipStream_t s) {
    const artalk_config& c = m->cfg;
    struct Restore { artalk_model* m; ~Restore() { kbucket(m, KB_OTHER); } } restore{m};
    Workspace& w = m->ws;
    artalk_model::InitHistory& ih = m->init_hist[m->precision == 1 ? 1 : 0];
    const long nb = (long)kNTok * c.code_dim, nf = 100L * c.code_dim * 4, nm = 180L * c.code_dim * 4;
    if (!ih.valid) {
        if (!ih.bits) {
            ih.bits = dalloc<uint8_t>(m, nb); ih.fdec = dalloc<float>(m, nf / 4); ih.msfeat = dalloc<float>(m, nm / 4);
            if (!ih.bits || !ih.fdec || !ih.msfeat) return fail(m, ARTALK_EHIP, "hipMalloc failed for the initial-history cache");
            HIPCHK(m, hipDeviceSynchronize());      // (dalloc's zero fill runs on the null stream)
        }
        launch_enc_input_zero(m->vae_mean, m->vae_std, m->enc_pos, w.enc_in, 1, s);
        run_reencode(m, 1, s);
        HIPCHK(m, hipMemcpyAsync(ih.bits, w.hist_bits, nb, hipMemcpyDeviceToDevice, s));
        HIPCHK(m, hipMemcpyAsync(ih.fdec, w.prev_fdec, nf, hipMemcpyDeviceToDevice, s));
        HIPCHK(m, hipMemcpyAsync(ih.msfeat, w.msfeat, nm, hipMemcpyDeviceToDevice, s));
        // The cache is complete before this call returns to its caller's enqueue loop: a later call on ANOTHER stream reads it with no
        // ordering against this fill otherwise.  Once per model and mode, beside the hipMalloc above.  A fill that raised a
        // health flag (the range guard of the P8 format, a non-finite value) is NOT kept: every later call would reuse the damaged
        // history with a clean status word; it is recomputed - and flagged again - by the next call instead.
        int flags[4] = {0, 0, 0, 0};
        HIPCHK(m, hipMemcpyAsync(flags, w.status, sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(m, hipStreamSynchronize(s));
        ih.valid = flags[0] == 0;
    }
    launch_broadcast16(ih.bits, w.hist_bits, nb, B, s);
    launch_broadcast16(ih.fdec, w.prev_fdec, nf, B, s);
    launch_broadcast16(ih.msfeat, w.msfeat, nm, B, s);
    launch_vq_embed(w.msfeat, kNTok - 1, m->vq_w, m->vq_b, m->prev_lvl_pos + kE, w.prev_in, kNTok, 1, w.style_cond, m->prev_lvl_pos, B, s);
    return ARTALK_OK;
}

// Everything of one chunk index that depends only on (B, fixed workspace pointers): capturable as one hipGraph.
// n_reencode: clips (a prefix of the B) whose generated motion is re-encoded into the next history - the clips that HAVE a next chunk.
// The reference re-encodes after every chunk (app/models.py:111-114), the last one included, and throws that result away.
void run_chunk_body(artalk_model* m, int B, hipStream_t s, int n_reencode = -1) {
    struct BodyScope { artalk_model* m; BodyScope(artalk_model* x) : m(x) { m->in_body = true; } ~BodyScope() { m->in_body = false; } } scope(m);
    const artalk_config& c = m->cfg;
    Workspace& w = m->view ? *m->view : m->ws;
    const long ldada = m->ada_n;
    const long cache_l = (long)w.maxB * 2 * kNTok * 3 * kE;   // floats per layer
    // f16x3 mode: the block's GEMM-only activations (modulated LN output, attention output, FFN hidden) are written in the P8 split
    // format by their producers, which lets every block GEMM use the LDS-DMA kernels (gemm_p8_sm_kernel at these grid sizes)
    const int p8 = m->precision == 1 ? 1 : 0;
    // K/V of the 181 history tokens, once per layer (raw prev tokens, not modulated: app/transformer.py:68-70)
    kbucket(m, KB_AR_MISC);
    tap_copy(m, TAP_PREV_IN, 0, w.prev_in, kNTok, kE, B, s);      // prev_attn_feat + prev_lvl_pos_embed (app/models.py:101, 2nd argument)
    roctxRangePushA("artalk.ar.history_kv");
    SiteExps& ex = m->ex;
    if (p8) launch_pack_split(w.prev_in, reinterpret_cast<unsigned int*>(w.prev_in_p8), (long)B * kNTok * kE, false, s, w.status, ex.hist_tok);   // one split for the 12 layers
    audit(m, "ar.history_tokens", -1, nullptr, p8 ? w.prev_in_p8 : w.prev_in, B * kNTok, kE, kE, p8, s, &ex.hist_tok);
    {
        // all blocks in one launch: N = depth x (E keys + E values), column group l = block l's weight rows / cache columns
        GemmArgs g;
        g.A = p8 ? w.prev_in_p8 : w.prev_in; g.a_packed = p8; g.a_exp = ex.hist_tok;
        g.lda = kE; g.W = m->ar_qkv_w + (long)kE * kE; g.ldw = kE; g.bias = m->ar_qkv_b + kE;
        g.C = w.cache + kE; g.ldc = 3 * kE; g.cmap = rowmap(kNTok, 2 * kNTok, 0);
        g.M = B * kNTok; g.N = c.ar_depth * 2 * kE; g.K = kE;
        g.ngrp = 2 * kE; g.grpW = (long)3 * kE * kE; g.grpB = 3 * kE; g.grpC = cache_l;
        g.Wp = packed_of(m, g.W);
        if (p8 && gemm_p8_eligible(g)) {
            gemm(m, g, s);
        } else {
            for (int l = 0; l < c.ar_depth; ++l) {          // exact-f32 mode / small batches: one launch per block
                const ARLayer& L = m->ar[l];
                GemmArgs q = g;
                q.Wp = nullptr; q.ngrp = 0; q.grpW = q.grpB = q.grpC = 0;
                q.W = L.qkv_w + (long)kE * kE; q.bias = L.qkv_b + kE; q.C = w.cache + l * cache_l + kE; q.N = 2 * kE;
                gemm(m, q, s);
            }
        }
    }
    roctxRangePop();
    launch_ar_begin(w.style_cond, m->lvl_pos, w.x, w.fhat, B, s);
    static const char* const kLevelName[kMaxLv] = {"artalk.ar.scale_step0", "artalk.ar.scale_step1", "artalk.ar.scale_step2",
                                                   "artalk.ar.scale_step3", "artalk.ar.scale_step4"};
    for (int p = 0; p < c.n_levels; ++p) {
        Range r_lv(kLevelName[p]);                       // K8-K14 of one scale step: 12 blocks, head, bits, next-scale features
        kbucket(m, KB_LEVEL0 + p);
        const int pn = m->pn[p], off = m->off[p], M = B * pn;
        const RowMap amap = rowmap(pn, kNTok, off);            // rows of the AdaLN table for this level's tokens
        // AdaLN-modulated LayerNorm into w.xmod: block l's first (which = 0) / second (1) norm, or the head's (l = depth).  Each is
        // either its own launch or - when the GEMM that produces x is split over K - written by that GEMM's reduce pass.
        auto ln_args = [&](int l, int which) {
            LnArgs n;
            n.X = w.x; n.ldx = kE; n.Y = w.xmod; n.ldy = kE; n.ldm = ldada; n.mmap = amap; n.M = M; n.D = kE; n.eps = 1e-6f;
            n.status = p8 ? w.status : nullptr;
            if (l < c.ar_depth) {
                const float* ada = w.ada + (long)l * 6 * kE;   // gamma1,gamma2,scale1,scale2,shift1,shift2 (app/transformer.py:32)
                n.scale = ada + (2 + which) * kE; n.shift = ada + (4 + which) * kE; n.out_p8 = p8;
                n.p8_exp = which == 0 ? ex.ar[l].ln1 : ex.ar[l].ln2;
            } else {                                           // head (app/models.py:145-148): scale, shift = split2; fp32 (exact logits GEMM)
                const float* hada = w.ada + (long)c.ar_depth * 6 * kE;
                n.scale = hada; n.shift = hada + kE; n.out_p8 = 0;
            }
            return n;
        };
        bool have_ln = false;                                  // w.xmod already holds the norm the next GEMM reads
        for (int l = 0; l < c.ar_depth; ++l) {
            const ARLayer& L = m->ar[l];
            const float* ada = w.ada + (long)l * 6 * kE;
            float* cache = w.cache + l * cache_l;
            SiteExps::AR& E = ex.ar[l];
            if (l == 0) tap_copy(m, TAP_BLK0_IN, off, w.x, pn, kE, B, s);        // attn_feat entering attn_blocks[0] (app/models.py:100)
            if (!have_ln) launch_layernorm(ln_args(l, 0), s);
            audit(m, "ar.block", l, ".ln1_mod", w.xmod, M, kE, kE, p8, s, &E.ln1);
            GemmArgs q;
            q.A = w.xmod; q.lda = kE; q.W = L.qkv_w; q.ldw = kE; q.bias = L.qkv_b; q.C = cache; q.ldc = 3 * kE;
            q.cmap = rowmap(pn, 2 * kNTok, kNTok + off); q.M = M; q.N = 3 * kE; q.K = kE; q.a_packed = p8; q.a_exp = E.ln1;
            // 1- and 5-token steps: a split q|k|v GEMM leaves its slabs to the attention kernel, whose workgroup (clip, head) sums the
            // rows of its own head (same order: slabs ascending, then the bias), writes them to the KV cache and goes on - one launch
            // less per block (ARTALK_FUSE_QKV_REDUCE=0: the separate reduce pass; a test compares both arms bit for bit)
            int qkv_slabs = 0;
            gemm(m, q, s, nullptr, (m->fuse_qkv_reduce && pn <= 16) ? &qkv_slabs : nullptr);
            AttnArgs a;
            if (qkv_slabs > 0) {
                a.slabs = w.splitk; a.n_slabs = qkv_slabs; a.slab_stride = (long)M * 3 * kE; a.slab_ld = 3 * kE; a.slab_bias = L.qkv_b;
            }
            a.Q = cache + (long)(kNTok + off) * 3 * kE; a.K = cache + kE; a.V = cache + 2 * kE;
            a.ldq = a.ldk = a.ldv = 3 * kE; a.q_bstride = a.k_bstride = a.v_bstride = (long)2 * kNTok * 3 * kE;
            a.O = w.attn_out; a.ldo = kE; a.o_bstride = (long)pn * kE;
            a.B = B; a.H = c.ar_heads; a.HD = kE / c.ar_heads; a.Lq = pn; a.Lk = kNTok + off + pn; a.scale = 1.0f;
            a.l2norm = 1; a.qscale = L.qscale; a.out_p8 = p8; a.split16 = p8; a.status = p8 ? w.status : nullptr; a.o_exp = E.attn;
            launch_attention(a, s);
            audit(m, "ar.block", l, ".attn_out", w.attn_out, M, kE, kE, p8, s, &E.attn);
            GemmArgs pj;
            pj.A = w.attn_out; pj.lda = kE; pj.W = L.proj_w; pj.ldw = kE; pj.bias = L.proj_b; pj.C = w.x; pj.ldc = kE;
            pj.gate = ada; pj.ldg = ldada; pj.gmap = amap; pj.R = w.x; pj.ldr = kE; pj.M = M; pj.N = kE; pj.K = kE; pj.a_packed = p8; pj.a_exp = E.attn;
            const LnArgs n2 = ln_args(l, 1);
            if (!gemm(m, pj, s, &n2)) launch_layernorm(n2, s);
            audit(m, "ar.block", l, ".ln2_mod", w.xmod, M, kE, kE, p8, s, &E.ln2);
            linear(m, w.xmod, kE, L.ffn1_w, L.ffn1_b, w.ffn_h, 4 * kE, M, 4 * kE, kE, ACT_GELU_TANH, nullptr, s, p8 ? (LF_A_P8 | LF_C_P8) : 0, nullptr, E.ln2, E.ffn);
            audit(m, "ar.block", l, ".ffn_hidden", w.ffn_h, M, 4 * kE, 4 * kE, p8, s, &E.ffn);
            GemmArgs f2;
            f2.A = w.ffn_h; f2.lda = 4 * kE; f2.W = L.ffn2_w; f2.ldw = 4 * kE; f2.bias = L.ffn2_b; f2.C = w.x; f2.ldc = kE;
            f2.gate = ada + kE; f2.ldg = ldada; f2.gmap = amap; f2.R = w.x; f2.ldr = kE; f2.M = M; f2.N = kE; f2.K = 4 * kE; f2.a_packed = p8; f2.a_exp = E.ffn;
            const LnArgs nn = ln_args(l + 1, 0);               // next block's first norm, or the head's after the last block
            have_ln = gemm(m, f2, s, &nn);
            if (l == 0) tap_copy(m, TAP_BLK0_OUT, off, w.x, pn, kE, B, s);        // output of attn_blocks[0] (app/transformer.py:43)
            if (l == c.ar_depth - 1) tap_copy(m, TAP_BLKL_OUT, off, w.x, pn, kE, B, s);   // output of the last block (app/models.py:102)
        }
        if (!have_ln) launch_layernorm(ln_args(c.ar_depth, 0), s);
        linear(m, w.xmod, kE, m->logits_w, m->logits_b, w.logits, 2 * c.code_dim, M, 2 * c.code_dim, kE, ACT_NONE, nullptr, s, LF_EXACT);
        tap_copy(m, TAP_LOGITS, off, w.logits, pn, 2 * c.code_dim, B, s);          // pred_motion_logits (app/models.py:103), fp32
        launch_ar_bits_next(w.logits, w.bits, w.fhat, w.nextfeat, B, p, s, w.status);
        if (p + 1 < c.n_levels)
            launch_vq_embed(w.nextfeat, m->pn[p + 1], m->vq_w, m->vq_b, m->lvl_pos + (long)m->off[p + 1] * kE, w.x, m->pn[p + 1], 0,
                            nullptr, nullptr, B, s);
    }
    stage_mark(m, s, PB_AR);
    // ---- VAE decode of [history | current] (bitwise_vae.py:105-113) ----
    const int H = c.vae_hidden;
    roctxRangePushA("artalk.vae.decode");               // K15
    kbucket(m, KB_VAE_DEC);
    launch_dec_input(w.prev_fdec, w.fhat, w.bits, m->dec_pos, w.dec_x, B, s);
    audit(m, "vae.decoder.input(fp32 A)", -1, nullptr, w.dec_x, B * 200, c.code_dim, c.code_dim, false, s, &ex.vae_dec_in);
    linear(m, w.dec_x, c.code_dim, m->dec.in_w, m->dec.in_b, w.vh, H, B * 200, H, c.code_dim, ACT_LEAKY02, nullptr, s, 0, nullptr, ex.vae_dec_in);
    run_vae_stack(m, m->dec, B, 200, 100, s);
    audit(m, "vae.decoder.output_head(fp32 A)", -1, nullptr, w.vh, B * 200, H, H, false, s, &ex.vae_dec_head);
    linear(m, w.vh, H, m->dec.out_w, m->dec.out_b, w.dec_out, c.motion_dim, B * 200, c.motion_dim, H, ACT_NONE, nullptr, s, 0, nullptr, ex.vae_dec_head);
    tap_copy(m, TAP_DEC_OUT, 0, w.dec_out, 200, c.motion_dim, B, s);               // dec_out before unnorm_with_stats (bitwise_vae.py:111)
    launch_dec_finish(w.dec_out, m->vae_mean, m->vae_std, m->enc_pos, w.motion_chunk, 100L * c.motion_dim, 0, w.enc_in, B, s, w.status);
    roctxRangePop();
    // ---- re-encode the generated motion into the next history (app/models.py:111-114) ----
    if (n_reencode < 0 || n_reencode > B) n_reencode = B;
    if (n_reencode > 0) run_reencode(m, n_reencode, s);
    stage_mark(m, s, PB_VAE);
}

// A view of the workspace for clips [b0, ...): every per-clip buffer pointer advanced by b0 clips (compact step buffers get a
// fixed 100-row region per clip so the two halves never overlap), its own split-K scratch.
Workspace clip_view(const artalk_model* m, int b0, int branch) {
    const artalk_config& c = m->cfg;
    Workspace v = m->ws;
    const long b = b0;
    v.ada += b * kNTok * m->ada_n; v.style_cond += b * kE; v.prev_in += b * kNTok * kE; v.prev_in_p8 += b * kNTok * kE;
    v.cache += b * 2 * kNTok * 3 * kE;
    v.x += b * 100 * kE; v.xmod += b * 100 * kE; v.attn_out += b * 100 * kE; v.ffn_h += b * 100 * 4 * kE;
    v.logits += b * 100 * 2 * c.code_dim; v.fhat += b * 100 * c.code_dim; v.nextfeat += b * 100 * c.code_dim;
    v.bits += b * kNTok * c.code_dim; v.hist_bits += b * kNTok * c.code_dim;
    v.prev_fdec += b * 100 * c.code_dim; v.msfeat += b * 180 * c.code_dim; v.dec_x += b * 200 * c.code_dim;
    const long H = c.vae_hidden;
    v.vh += b * 200 * H; v.vln += b * 200 * H; v.vqkv += b * 200 * 3 * H; v.vatt += b * 200 * H; v.vmlp += b * 200 * H * 3 / 2;
    v.dec_out += b * 200 * c.motion_dim; v.enc_in += b * 100 * 128; v.enc_out += b * 100 * c.code_dim;
    v.motion_chunk += b * 100 * c.motion_dim;
    v.splitk = m->ws.splitk_b[branch];
    return v;
}

// The AR/VAE body of one chunk index for B clips.  The small scale steps leave most CUs idle, and clips are independent, so
// for B >= 8 the batch is cut into two halves that run as parallel branches (stream s and m->side_stream; inside a capture this
// becomes a fork/join in the hipGraph): one half's GPU-filling step overlaps the other half's latency-bound ones.
int ensure_side_streams(artalk_model* m, int n);
int body_branches(const artalk_model* m, int B);
int run_chunk_body_split(artalk_model* m, int B, hipStream_t s, int n_next = -1) {
    if (n_next < 0 || n_next > B) n_next = B;
    if (m->profiling == 3 && B >= 8 && m->branches != 1 && !m->audit && !m->tap) {
        // kernel timer: the SAME clip groups (and therefore the same kernels, tiles and split-K factors) as the graphs replay, one
        // after the other on one stream - the sums are then those of the production launches, without their overlap
        const int NS = body_branches(m, B);
        for (int h = 0; h < NS; ++h) {
            const int lo = (int)((long)B * h / NS), hi = (int)((long)B * (h + 1) / NS);
            Workspace v = clip_view(m, lo, h);
            m->view = &v;
            run_chunk_body(m, hi - lo, s, std::max(0, std::min(n_next - lo, hi - lo)));
            m->view = nullptr;
        }
        return ARTALK_OK;
    }
    if (B < 8 || m->profiling >= 2 || m->branches == 1 || m->audit || m->tap) { run_chunk_body(m, B, s, n_next); return ARTALK_OK; }
    if (int rc = ensure_side_streams(m, 1)) return rc;
    const int B0 = (B + 1) / 2, B1 = B - B0;
    Workspace v0 = clip_view(m, 0, 0), v1 = clip_view(m, B0, 1);
    HIPCHK(m, hipEventRecord(m->fork_ev, s));
    HIPCHK(m, hipStreamWaitEvent(m->side_stream[0], m->fork_ev, 0));
    m->view = &v0; run_chunk_body(m, B0, s, std::min(n_next, B0));
    m->view = &v1; run_chunk_body(m, B1, m->side_stream[0], std::max(0, n_next - B0));
    m->view = nullptr;
    HIPCHK(m, hipEventRecord(m->join_ev[0], m->side_stream[0]));
    HIPCHK(m, hipStreamWaitEvent(s, m->join_ev[0], 0));
    return ARTALK_OK;
}

// Body of one chunk index through hipGraphs: the batch is cut into NS clip groups (1, 2 or 4), one graph per (active batch, group),
// launched on NS streams so that they really run concurrently (a single captured graph with a fork/join ran almost serially).
// HIP multiplexes streams onto a few hardware queues (4 by default): every extra stream a process creates can end up sharing a
// queue with the caller's stream and serialise the branches, so side streams are created lazily and only as many as are used.
int ensure_side_streams(artalk_model* m, int n) {
    if (!m->fork_ev) HIPCHK(m, hipEventCreateWithFlags(&m->fork_ev, hipEventDisableTiming));
    for (int i = 0; i < n && i < 3; ++i) {
        if (m->side_stream[i]) continue;
        if (m->cu_mask.empty()) HIPCHK(m, hipStreamCreateWithFlags(&m->side_stream[i], hipStreamNonBlocking));
        else HIPCHK(m, hipExtStreamCreateWithCUMask(&m->side_stream[i], (uint32_t)m->cu_mask.size(), m->cu_mask.data()));
        HIPCHK(m, hipEventCreateWithFlags(&m->join_ev[i], hipEventDisableTiming));
    }
    return ARTALK_OK;
}
int body_branches(const artalk_model* m, int B) {
    // measured at batch 32: 1 branch 54.9 ms, 2 branches 49.5 ms, 4 branches 77-120 ms (queue sharing / contention)
    const int want = m->branches > 0 ? m->branches : (B >= 8 ? 2 : 1);
    return B >= 2 * want ? want : (B >= 8 ? 2 : 1);
}
// Graph cache: one hipGraphExec per (active clips, clip group, precision, re-encoded clips of the group).  The key space is bounded
// (ADVICE r4): the re-encode count of a group is rounded UP to a multiple of 8 (re-encoding a clip without a next chunk is what the
// reference does anyway, app/models.py:111-114; its history is never read), so a group contributes at most 1 + size / 8 keys per
// active-batch size instead of one per (B_j, B_j+1) pair; and the cache holds at most kMaxGraphs executables, least recently used
// evicted (a serving loop with ever-changing ragged batches re-captures instead of growing without bound).
constexpr size_t kMaxGraphs = 96;
int evict_graphs(artalk_model* m, const long long* keep, int n_keep) {
    while (m->graphs.size() >= kMaxGraphs) {
        auto victim = m->graphs.end();
        for (auto it = m->graphs.begin(); it != m->graphs.end(); ++it) {
            bool pinned = false;
            for (int i = 0; i < n_keep; ++i) pinned |= it->first == keep[i];
            if (!pinned && (victim == m->graphs.end() || it->second.last_use < victim->second.last_use)) victim = it;
        }
        if (victim == m->graphs.end()) break;
        HIPCHK(m, hipDeviceSynchronize());      // the victim may still be running on a side stream of an earlier call
        (void)hipGraphExecDestroy(victim->second.exec);
        m->graphs.erase(victim);
    }
    return ARTALK_OK;
}
int run_chunk_body_graphs(artalk_model* m, int B, hipStream_t s, int n_next = -1) {
    if (n_next < 0 || n_next > B) n_next = B;
    if (B > 4095) { run_chunk_body(m, B, s, n_next); return ARTALK_OK; }      // (graph keys hold 12 bits of clip counts)
    const int NS = body_branches(m, B);
    if (NS > 1) { if (int rc = ensure_side_streams(m, NS - 1)) return rc; }
    Workspace views[4];
    int b0[5], nres[4];
    for (int h = 0; h <= NS; ++h) b0[h] = (int)((long)B * h / NS);
    hipStream_t st[4] = {s, m->side_stream[0], m->side_stream[1], m->side_stream[2]};
    long long keys[4];
    for (int h = 0; h < NS; ++h) {
        const int gsz = b0[h + 1] - b0[h];
        const int need = std::max(0, std::min(n_next - b0[h], gsz));      // clips of this group that have a next chunk
        nres[h] = std::min(gsz, (need + 7) / 8 * 8);
        keys[h] = ((long long)(((B * 8 + NS) * 4 + h) * 2 + m->precision)) * 4096 + nres[h];
    }
    for (int h = 0; h < NS; ++h) {
        views[h] = clip_view(m, b0[h], h);
        auto hit = m->graphs.find(keys[h]);
        if (hit != m->graphs.end()) { hit->second.last_use = ++m->graph_clock; continue; }
        if (int rc = evict_graphs(m, keys, NS)) return rc;
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        HIPCHK(m, hipStreamBeginCapture(st[h], hipStreamCaptureModeThreadLocal));
        m->in_graph_body = true;
        m->view = &views[h];
        run_chunk_body(m, b0[h + 1] - b0[h], st[h], nres[h]);
        m->view = nullptr;
        m->in_graph_body = false;
        HIPCHK(m, hipStreamEndCapture(st[h], &graph));
        HIPCHK(m, hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        (void)hipGraphDestroy(graph);
        m->graphs.emplace(keys[h], artalk_model::GraphEntry{exec, ++m->graph_clock});
        ++m->graph_captures;
    }
    if (NS > 1) {
        HIPCHK(m, hipEventRecord(m->fork_ev, s));
        for (int h = 1; h < NS; ++h) HIPCHK(m, hipStreamWaitEvent(st[h], m->fork_ev, 0));
    }
    for (int h = 0; h < NS; ++h) HIPCHK(m, hipGraphLaunch(m->graphs[keys[h]].exec, st[h]));
    for (int h = 1; h < NS; ++h) {
        HIPCHK(m, hipEventRecord(m->join_ev[h - 1], st[h]));
        HIPCHK(m, hipStreamWaitEvent(s, m->join_ev[h - 1], 0));
    }
    return ARTALK_OK;
}

void free_stage(artalk_model* m) {
    for (auto& st : m->stage) {
        if (st.src) (void)hipHostFree(st.src);
        if (st.has) (void)hipHostFree(st.has);
        st.src = nullptr; st.has = nullptr; st.used = false;
    }
    m->stage_cap_c = m->stage_cap_b = 0;
}
int ensure_stage(artalk_model* m, int maxB, int maxC) {
    if (!m->h_status) {
        HIPCHK(m, hipHostMalloc(reinterpret_cast<void**>(&m->h_status), artalk_model::kStatusSlots * sizeof(int), hipHostMallocDefault));
        std::memset(m->h_status, 0, artalk_model::kStatusSlots * sizeof(int));
        for (auto& e : m->status_ev) HIPCHK(m, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    if (maxB <= m->stage_cap_b && maxC <= m->stage_cap_c) return ARTALK_OK;
    for (auto& st : m->stage) if (st.used) HIPCHK(m, hipEventSynchronize(st.done));
    free_stage(m);
    for (auto& st : m->stage) {
        HIPCHK(m, hipHostMalloc(reinterpret_cast<void**>(&st.src), (size_t)maxC * sizeof(long), hipHostMallocDefault));
        HIPCHK(m, hipHostMalloc(reinterpret_cast<void**>(&st.has), (size_t)maxB, hipHostMallocDefault));
        if (!st.done) HIPCHK(m, hipEventCreateWithFlags(&st.done, hipEventDisableTiming));
    }
    m->stage_cap_b = maxB; m->stage_cap_c = maxC;
    return ARTALK_OK;
}
// next slot of the ring; waits only if the call that last used it has not consumed its copies yet
int next_stage(artalk_model* m, artalk_model::Stage** out) {
    artalk_model::Stage& st = m->stage[m->stage_next];
    m->stage_next = (m->stage_next + 1) % artalk_model::kStageSlots;
    if (st.used) HIPCHK(m, hipEventSynchronize(st.done));
    *out = &st;
    return ARTALK_OK;
}
// end of a call: hand the device status word to the host asynchronously (artalk_get_status / artalk_poll_status read it)
int publish_status(artalk_model* m, hipStream_t s) {
    const int slot = (int)(m->ticket % artalk_model::kStatusSlots);
    // the slot's previous user is kStatusSlots calls back: its copy has long landed unless the caller queued more calls than slots
    if (m->ticket >= artalk_model::kStatusSlots) HIPCHK(m, hipEventSynchronize(m->status_ev[slot]));
    HIPCHK(m, hipMemcpyAsync(m->h_status + slot, m->ws.status, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(m, hipEventRecord(m->status_ev[slot], s));
    ++m->ticket;
    m->status_pending = true;
    return ARTALK_OK;
}

int reserve(artalk_model* m, int maxB, int maxC) {
    const artalk_config& c = m->cfg;
    if (maxB <= m->ws.maxB && maxC <= m->ws.maxC) return ensure_stage(m, m->ws.maxB, m->ws.maxC);
    // grow: drop the old workspace (and the graphs that captured its pointers) and allocate the larger one
    maxB = std::max(maxB, m->ws.maxB); maxC = std::max(maxC, m->ws.maxC);
    (void)hipDeviceSynchronize();
    for (auto& g : m->graphs) (void)hipGraphExecDestroy(g.second.exec);
    m->graphs.clear();
    for (void* p : m->ws_allocs) if (p) (void)hipFree(p);
    m->ws_allocs.clear();
    m->ws = Workspace();
    m->stream_B = 0;          // the streaming history lived in the workspace that was just dropped: a session must begin again
    m->status_pending = false;
    Workspace& w = m->ws;
    const int CD = c.w2v_conv_dim, Hs = c.w2v_hidden;
    w.maxB = maxB; w.maxC = maxC; w.G = std::min(maxC, 96);
    auto F = [&](int64_t n) { w.bytes += n * 4; return dalloc_in<float>(m->ws_allocs, n); };
    const int G = w.G;
    w.src_off = dalloc_in<long>(m->ws_allocs, maxC);
    w.xnorm = F((int64_t)G * kSamplesPerChunk);
    w.convA = F(((int64_t)G * m->conv_S[0] + 16) * CD);
    w.convB = F(((int64_t)G * m->conv_S[1] + 16) * CD);
    const int64_t M = (int64_t)G * m->Ts;
    w.h0 = F(M * Hs); w.h1 = F(M * Hs); w.xln = F(M * Hs); w.qkv = F(M * 3 * Hs); w.att = F(M * Hs); w.ffn = F(M * c.w2v_ffn);
    w.silu_cond = F((int64_t)maxC * kNTok * kCond);
    w.ada = F((int64_t)maxB * kNTok * m->ada_n);
    w.style_cond = F((int64_t)maxB * kE);
    w.prev_in = F((int64_t)maxB * kNTok * kE); w.prev_in_p8 = F((int64_t)maxB * kNTok * kE);
    w.cache = F((int64_t)c.ar_depth * maxB * 2 * kNTok * 3 * kE);
    w.x = F((int64_t)maxB * 100 * kE); w.xmod = F((int64_t)maxB * 100 * kE); w.attn_out = F((int64_t)maxB * 100 * kE);
    w.ffn_h = F((int64_t)maxB * 100 * 4 * kE); w.logits = F((int64_t)maxB * 100 * 2 * c.code_dim);
    w.splitk_floats = (int64_t)8 << 20; w.splitk = F(w.splitk_floats); w.splitk_b[0] = w.splitk;
    for (int i = 1; i < 4; ++i) w.splitk_b[i] = F(w.splitk_floats);
    w.fhat = F((int64_t)maxB * 100 * c.code_dim); w.nextfeat = F((int64_t)maxB * 100 * c.code_dim);
    w.bits = dalloc_in<uint8_t>(m->ws_allocs, (int64_t)maxB * kNTok * c.code_dim);
    w.hist_bits = dalloc_in<uint8_t>(m->ws_allocs, (int64_t)maxB * kNTok * c.code_dim);
    w.has_style = dalloc_in<uint8_t>(m->ws_allocs, maxB);
    w.status = dalloc_in<int>(m->ws_allocs, 4);
    w.prev_fdec = F((int64_t)maxB * 100 * c.code_dim); w.msfeat = F((int64_t)maxB * 180 * c.code_dim);
    const int H = c.vae_hidden;
    w.dec_x = F((int64_t)maxB * 200 * c.code_dim); w.vh = F((int64_t)maxB * 200 * H); w.vln = F((int64_t)maxB * 200 * H);
    w.vqkv = F((int64_t)maxB * 200 * 3 * H); w.vatt = F((int64_t)maxB * 200 * H); w.vmlp = F((int64_t)maxB * 200 * H * 3 / 2);
    w.dec_out = F((int64_t)maxB * 200 * c.motion_dim); w.enc_in = F((int64_t)maxB * 100 * 128);
    w.enc_out = F((int64_t)maxB * 100 * c.code_dim); w.motion_chunk = F((int64_t)maxB * 100 * c.motion_dim);
    const int S = c.style_dim, SL = c.style_len;
    w.s_in = F((int64_t)maxB * SL * 128); w.s_h = F((int64_t)maxB * SL * S); w.s_qkv = F((int64_t)maxB * SL * 3 * S);
    w.s_att = F((int64_t)maxB * SL * S); w.s_ffn = F((int64_t)maxB * SL * c.style_ffn); w.s_tmp = F((int64_t)maxB * SL * S);
    for (void* p : m->ws_allocs)
        if (!p) return fail(m, ARTALK_EHIP, "hipMalloc failed while reserving workspace");
    // the zero fills above run on the null stream; callers use non-blocking streams, which do not order against it
    HIPCHK(m, hipDeviceSynchronize());
    return ensure_stage(m, maxB, maxC);
}

}  // namespace

// =================================================================================================== C ABI
extern "C" {

const char* artalk_last_error(const artalk_model* m) { return m ? m->err.c_str() : g_create_error.c_str(); }

int artalk_create(int device_id, const artalk_config* cfg, artalk_model** out) {
    if (!cfg || !out) { g_create_error = "null argument"; return ARTALK_EINVAL; }
    const artalk_config& c = *cfg;
    static const int want_pn[5] = {1, 5, 25, 50, 100};
    bool ok = c.n_levels == 5 && c.code_dim == 32 && c.motion_dim == 106 && c.vae_hidden == 512 && c.w2v_hidden == kCond &&
              c.w2v_conv_dim == 512 && c.w2v_n_conv >= 2 && c.w2v_n_conv <= 8 && c.ar_heads * 64 == kE && c.vae_heads * 64 == c.vae_hidden &&
              c.w2v_heads * 64 == c.w2v_hidden && c.style_dim == 128 && c.style_heads * 32 == c.style_dim && c.style_len == 50 &&
              c.w2v_conv_kernel[0] == 10 && c.w2v_conv_stride[0] == 5 && c.w2v_ffn % 32 == 0 && c.style_ffn % 32 == 0 &&
              c.w2v_layers >= 1 && c.w2v_layers <= 64 && c.ar_depth >= 1 && c.ar_depth <= 32 && c.vae_depth >= 1 && c.vae_depth <= 16 && c.style_layers >= 1 && c.style_layers <= 8;      // (SiteExps tables)
    for (int i = 0; ok && i < 5; ++i) ok = c.patch_nums[i] == want_pn[i];
    if (!ok) { g_create_error = "unsupported configuration (kernels are specialised for assets/config.json + XLS-R-300M widths)"; return ARTALK_EINVAL; }
    if (hipSetDevice(device_id) != hipSuccess) { g_create_error = "hipSetDevice failed"; return ARTALK_EHIP; }
    artalk_model* m = new artalk_model();
    m->cfg = c; m->device = device_id;
    if (const char* e = getenv("ARTALK_FUSE_QKV_REDUCE")) m->fuse_qkv_reduce = atoi(e) != 0;
    // conv stack geometry: T_l valid frames; row stride S_l per chunk with S_l = 2*S_{l+1} so that one GEMM covers all chunks
    int T = kSamplesPerChunk;
    for (int i = 0; i < c.w2v_n_conv; ++i) { T = (T - c.w2v_conv_kernel[i]) / c.w2v_conv_stride[i] + 1; m->conv_T[i] = T; }
    m->n_conv = c.w2v_n_conv; m->Tw = T; m->Ts = T + 1;
    int S = m->Ts;
    for (int i = c.w2v_n_conv - 1; i >= 0; --i) {
        m->conv_S[i] = S;
        if (i > 0) {
            if (c.w2v_conv_stride[i] != 2 || m->conv_T[i - 1] > 2 * S) { g_create_error = "conv stack geometry unsupported"; delete m; return ARTALK_EINVAL; }
            S *= 2;
        }
    }
    for (int i = 0; i < 5; ++i) { m->pn[i] = c.patch_nums[i]; m->off[i + 1] = m->off[i] + c.patch_nums[i]; }
    if (init_ms_tables() != 0) { g_create_error = "uploading the interpolation tables to the device failed"; delete m; return ARTALK_EHIP; }
    attention_prepare();
    gemm_p8_prepare();
    const int rc = build_registry(m);
    if (rc != ARTALK_OK) { g_create_error = m->err; artalk_destroy(m); return rc; }
    *out = m;
    return ARTALK_OK;
}

void artalk_destroy(artalk_model* m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    (void)hipDeviceSynchronize();
    for (auto& g : m->graphs) (void)hipGraphExecDestroy(g.second.exec);
    for (auto e : m->ev_pool) (void)hipEventDestroy(e);
    for (auto e : m->ktimer.pool) (void)hipEventDestroy(e);
    if (m->own_stream) (void)hipStreamDestroy(m->own_stream);
    for (int i = 0; i < 3; ++i) {
        if (m->side_stream[i]) (void)hipStreamDestroy(m->side_stream[i]);
        if (m->join_ev[i]) (void)hipEventDestroy(m->join_ev[i]);
    }
    if (m->fork_ev) (void)hipEventDestroy(m->fork_ev);
    free_stage(m);
    for (auto& st : m->stage) if (st.done) (void)hipEventDestroy(st.done);
    if (m->h_status) (void)hipHostFree(m->h_status);
    for (auto& e : m->status_ev) if (e) (void)hipEventDestroy(e);
    for (void* p : m->allocs) if (p) (void)hipFree(p);
    for (void* p : m->ws_allocs) if (p) (void)hipFree(p);
    delete m;
}

int artalk_set_tensor(artalk_model* m, const char* key, const void* host_ptr, int dtype, int ndim, const int64_t* shape) {
    if (!m || !key || !host_ptr || !shape) return ARTALK_EINVAL;
    if (m->finalized) return fail(m, ARTALK_ESTATE, "weights are final after artalk_finalize_weights (derived layouts and packed copies exist); create a new model");
    auto it = m->slots.find(key);
    if (it == m->slots.end()) return fail(m, ARTALK_EKEY, std::string("Unexpected key in state_dict: ") + key);
    Slot& s = it->second;
    bool same = (int)s.shape.size() == ndim && dtype == s.dtype;
    for (int i = 0; same && i < ndim; ++i) same = s.shape[i] == shape[i];
    if (!same) return fail(m, ARTALK_EINVAL, std::string("size mismatch for ") + key);
    (void)hipSetDevice(m->device);
    const int64_t n = s.numel();
    if (dtype == ARTALK_DTYPE_I64) {
        s.host_i64.assign((const int64_t*)host_ptr, (const int64_t*)host_ptr + n);
        s.set = true;
        return ARTALK_OK;
    }
    const float* src = (const float*)host_ptr;
    if (s.kind == SK_HOST || n <= 4096) s.host.assign(src, src + n);
    int rc = ARTALK_OK;
    switch (s.kind) {
        case SK_DIRECT: rc = upload(m, s.dst, src, n); break;
        case SK_PADK: {   // [out, in] -> [out, pad_to], zero padded
            const int64_t o = s.shape[0], in = s.shape[1];
            std::vector<float> t((size_t)o * s.pad_to, 0.f);
            for (int64_t r = 0; r < o; ++r) std::memcpy(&t[r * s.pad_to], src + r * in, in * sizeof(float));
            rc = upload(m, s.dst, t.data(), (int64_t)t.size());
            break;
        }
        case SK_CONV: {   // [out, cin, k] -> [out, k, cin]: the K index of the conv-as-GEMM is tap*cin + ci
            const int64_t o = s.shape[0], ci = s.shape[1], k = s.shape[2];
            std::vector<float> t((size_t)n);
            for (int64_t a = 0; a < o; ++a)
                for (int64_t b = 0; b < ci; ++b)
                    for (int64_t d = 0; d < k; ++d) t[(a * k + d) * ci + b] = src[(a * ci + b) * k + d];
            rc = upload(m, s.dst, t.data(), n);
            break;
        }
        default: break;
    }
    if (rc == ARTALK_OK) s.set = true;
    return rc;
}

int artalk_finalize_weights(artalk_model* m) {
    if (!m) return ARTALK_EINVAL;
    (void)hipSetDevice(m->device);
    std::string missing;
    int nmiss = 0;
    for (auto& kv : m->slots)
        if (!kv.second.set) { if (nmiss++ < 8) missing += (missing.empty() ? "" : ", ") + kv.first; }
    if (nmiss) return fail(m, ARTALK_EMISSING, "Missing key(s) in state_dict: " + missing + (nmiss > 8 ? ", ..." : ""));
    const artalk_config& c = m->cfg;
    // level / position tables (app/models.py:74-75)
    {
        const Slot& li = m->slots["lvl_idx"]; const Slot& le = m->slots["lvl_embed.weight"];
        const Slot& pe = m->slots["pos_embed"]; const Slot& ppe = m->slots["prev_pos_embed"];
        std::vector<float> a((size_t)kNTok * kE), b((size_t)kNTok * kE);
        int tok = 0;
        for (int p = 0; p < c.n_levels; ++p)
            for (int i = 0; i < c.patch_nums[p]; ++i, ++tok)
                if (li.host_i64[tok] != p) return fail(m, ARTALK_EINVAL, "lvl_idx does not match V_PATCH_NUMS");
        for (int t = 0; t < kNTok; ++t)
            for (int e = 0; e < kE; ++e) {
                const float lv = le.host[(size_t)li.host_i64[t] * kE + e];
                a[(size_t)t * kE + e] = lv + pe.host[(size_t)t * kE + e];
                b[(size_t)t * kE + e] = lv + ppe.host[(size_t)t * kE + e];
            }
        if (int rc = upload(m, m->lvl_pos, a.data(), (int64_t)a.size())) return rc;
        if (int rc = upload(m, m->prev_lvl_pos, b.data(), (int64_t)b.size())) return rc;
        // the masks are implied by patch_nums (block-causal by level; VAE: history rows see history only); verify
        const Slot& mk = m->slots["attn_bias_for_masking"];
        for (int q = 0; q < kNTok; ++q)
            for (int k = 0; k < 2 * kNTok; ++k) {
                const float v = mk.host[(size_t)q * 2 * kNTok + k];
                const bool vis = k < kNTok || li.host_i64[q] >= li.host_i64[k - kNTok];
                if (vis ? (v != 0.f) : !(std::isinf(v) && v < 0)) return fail(m, ARTALK_EINVAL, "attn_bias_for_masking is not the level-causal mask");
            }
        const Slot& vm = m->slots["basic_vae.attn_mask"];
        for (int q = 0; q < 200; ++q)
            for (int k = 0; k < 200; ++k) {
                const float v = vm.host[(size_t)q * 200 + k];
                const bool vis = !(q < 100 && k >= 100);
                if (vis ? (v != 0.f) : !(std::isinf(v) && v < 0)) return fail(m, ARTALK_EINVAL, "basic_vae.attn_mask is not the expected mask");
            }
    }
    // learned attention scale: exp(min(scale_mul, log 100))  (app/transformer.py:72)
    for (int i = 0; i < c.ar_depth; ++i) {
        const Slot& sm = m->slots["attn_blocks." + std::to_string(i) + ".attn.scale_mul_1H11"];
        std::vector<float> q(c.ar_heads);
        for (int h = 0; h < c.ar_heads; ++h) q[h] = std::exp(std::min(sm.host[h], (float)std::log(100.0)));
        if (int rc = upload(m, m->ar[i].qscale, q.data(), c.ar_heads)) return rc;
    }
    // style PE row pe[:, seq_len]  (style_encoder.py:59)
    {
        const Slot& pe = m->slots["style_encoder.PE.pe"];
        if (int rc = upload(m, m->st_pe, pe.host.data() + (size_t)c.style_len * c.style_dim, c.style_dim)) return rc;
    }
    // pos-conv weight-norm fold (dim=2): w[o,i,k] = g[k] * v[o,i,k] / ||v[:,:,k]||, re-laid out [o][k][i]
    {
        const std::string pc = "audio_encoder.encoder.pos_conv_embed.conv.parametrizations.weight.";
        const Slot& g = m->slots[pc + "original0"]; const Slot& v = m->slots[pc + "original1"];
        const int64_t O = v.shape[0], I = v.shape[1], K = v.shape[2];
        std::vector<double> nrm(K, 0.0);
        for (int64_t o = 0; o < O; ++o)
            for (int64_t i = 0; i < I; ++i)
                for (int64_t k = 0; k < K; ++k) { const double x = v.host[(o * I + i) * K + k]; nrm[k] += x * x; }
        std::vector<float> t((size_t)O * I * K);
        for (int64_t k = 0; k < K; ++k) {
            const float nk = (float)std::sqrt(nrm[k]);
            for (int64_t o = 0; o < O; ++o)
                for (int64_t i = 0; i < I; ++i) t[(o * K + k) * I + i] = v.host[(o * I + i) * K + k] * (g.host[k] / nk);
        }
        if (int rc = upload(m, m->pos_w, t.data(), (int64_t)t.size())) return rc;
    }
    for (auto& kv : m->slots) { std::vector<float>().swap(kv.second.host); }
    // packed f16x3 copies of every weight matrix (same size and indexing as the fp32 original)
    for (auto& r : m->wranges) {
        if (r.n < 1024 || r.packed) continue;
        r.packed = dalloc<unsigned int>(m, r.n);
        if (!r.packed) return fail(m, ARTALK_EHIP, "hipMalloc failed for packed weights");
        launch_pack_split(r.base, r.packed, r.n, true, nullptr);
    }
    HIPCHK(m, hipDeviceSynchronize());
    m->finalized = true;
    return ARTALK_OK;
}

int artalk_reserve(artalk_model* m, int max_batch, int max_total_chunks) {
    if (!m || max_batch <= 0 || max_total_chunks < max_batch) return ARTALK_EINVAL;
    (void)hipSetDevice(m->device);
    return reserve(m, max_batch, max_total_chunks);
}
int64_t artalk_workspace_bytes(const artalk_model* m) { return m ? m->ws.bytes : 0; }
int64_t artalk_weight_bytes(const artalk_model* m) { return m ? m->weight_bytes : 0; }

int artalk_set_profiling(artalk_model* m, int level) { if (!m || level < 0 || level > 3) return ARTALK_EINVAL; m->profiling = level; return ARTALK_OK; }
int artalk_set_precision(artalk_model* m, int mode) {
    if (!m || (mode != 0 && mode != 1)) return ARTALK_EINVAL;
    m->precision = mode;   // captured graphs are keyed by mode: switching costs nothing and keeps both sets
    return ARTALK_OK;
}
int artalk_set_graphs(artalk_model* m, int enable) {
    if (!m) return ARTALK_EINVAL;
    m->use_graphs = (enable & 0xff) != 0;
    const int br = (enable >> 8) & 0xff;          // tuning: enable | (branches << 8) forces 1 / 2 / 4 concurrent clip groups
    if (br >= 0 && br <= 4) m->branches = br; else return ARTALK_EINVAL;
    if ((enable >> 16) & 0xffff) {                // tuning: split-K thresholds, (tiles/16) << 16 | (target/16) << 24
        m->splitk_tiles = ((enable >> 16) & 0xff) * 16; m->splitk_target = ((enable >> 24) & 0xff) * 16;
    }
    (void)hipSetDevice(m->device); (void)hipDeviceSynchronize();
    for (auto& g : m->graphs) (void)hipGraphExecDestroy(g.second.exec);
    m->graphs.clear();
    m->init_hist[0].valid = m->init_hist[1].valid = false;      // "the same run_reencode as every later history": the split-K tuning changed
    return ARTALK_OK;
}
// number of captured body graphs the model holds (bounded: run_chunk_body_graphs) / captures since the model was created
int artalk_graph_count(const artalk_model* m, long long* captures) {
    if (!m) return ARTALK_EINVAL;
    if (captures) *captures = m->graph_captures;
    return (int)m->graphs.size();
}

// Headroom audit of the f16x3 operand format (tools/p8_headroom.py): while enabled, every artalk_infer runs without graphs and
// records, per producer site of a P8 operand, max |x| * 16 (the value that must stay below fp16's 65504).  artalk_get_audit
// synchronises the device and returns the number of sites; names are written NUL-separated into names_buf.
int artalk_set_audit(artalk_model* m, int enable) {
    if (!m) return ARTALK_EINVAL;
    (void)hipSetDevice(m->device);
    if (enable && !m->audit_vals) {
        HIPCHK(m, hipMalloc(reinterpret_cast<void**>(&m->audit_vals), kAuditSlots * sizeof(unsigned int)));
        m->allocs.push_back(m->audit_vals);
    }
    if (enable) {
        HIPCHK(m, hipDeviceSynchronize());
        HIPCHK(m, hipMemset(m->audit_vals, 0, kAuditSlots * sizeof(unsigned int)));
        m->audit_names.clear(); m->audit_index.clear(); m->audit_exp.clear();
    }
    m->audit = enable != 0;
    return ARTALK_OK;
}
// Per-site operand scales of the f16x3 format from the audit's maxima (SiteExps): after artalk_infer calls with the audit on - in EXACT-F32
// mode, where no site can overflow on the way to a later one - every site whose max |x| * 2^e * headroom exceeds fp16's 65504 gets the
// largest exponent e (<= the default 4, >= -8) that fits.  Exponents only go down (a later calibration on tamer inputs never undoes an
// earlier one; artalk_reset_scales does).  Captured graphs and the initial-history cache are dropped when anything changed.
// Returns the number of sites changed (>= 0), or a negative error: ARTALK_ESTATE without audit data, ARTALK_EINVAL for a non-finite
// maximum (run the pass in f32 mode) or one that no exponent >= -8 can hold.
int artalk_calibrate(artalk_model* m, float headroom) {
    if (!m || !(headroom >= 1.0f)) return ARTALK_EINVAL;
    if (!m->audit_vals || m->audit_names.empty()) { m->err = "artalk_calibrate: no audit data (artalk_set_audit(1), then artalk_infer in f32 mode)"; return ARTALK_ESTATE; }
    (void)hipSetDevice(m->device);
    if (hipDeviceSynchronize() != hipSuccess) return ARTALK_EHIP;
    const int n = (int)m->audit_names.size();
    std::vector<unsigned int> bits((size_t)n);
    if (hipMemcpy(bits.data(), m->audit_vals, n * sizeof(unsigned int), hipMemcpyDeviceToHost) != hipSuccess) return ARTALK_EHIP;
    int changed = 0;
    for (int i = 0; i < n; ++i) {
        int* ex = m->audit_exp[i];
        if (!ex) continue;
        float mx;
        std::memcpy(&mx, &bits[i], sizeof(float));
        if (!std::isfinite(mx)) { m->err = "artalk_calibrate: site " + m->audit_names[i] + " is not finite (calibrate in f32 mode)"; return ARTALK_EINVAL; }
        if (mx <= 0.f) continue;
        int e = *ex;
        while (e > -8 && (double)mx * std::ldexp(1.0, e) * headroom > 65504.0) --e;
        if ((double)mx * std::ldexp(1.0, e) > 65504.0) { m->err = "artalk_calibrate: site " + m->audit_names[i] + " exceeds every supported scale"; return ARTALK_EINVAL; }
        if (e != *ex) { *ex = e; ++changed; }
    }
    if (changed) {
        for (auto& g : m->graphs) (void)hipGraphExecDestroy(g.second.exec);      // the exponents are kernel arguments of the captured launches
        m->graphs.clear();
        m->init_hist[0].valid = m->init_hist[1].valid = false;
    }
    const int* p = reinterpret_cast<const int*>(&m->ex);
    m->scales_changed = 0;
    for (size_t i = 0; i < sizeof(SiteExps) / sizeof(int); ++i) m->scales_changed += p[i] != kActExp;
    return changed;
}
int artalk_reset_scales(artalk_model* m) {
    if (!m) return ARTALK_EINVAL;
    (void)hipSetDevice(m->device); (void)hipDeviceSynchronize();
    m->ex = SiteExps();
    m->scales_changed = 0;
    for (auto& g : m->graphs) (void)hipGraphExecDestroy(g.second.exec);
    m->graphs.clear();
    m->init_hist[0].valid = m->init_hist[1].valid = false;
    return ARTALK_OK;
}
// exps[i] = exponent of audit site i (the order of artalk_get_audit); returns the number written
int artalk_get_scales(artalk_model* m, int* exps, int max_n) {
    if (!m || !exps || max_n <= 0) return ARTALK_EINVAL;
    const int n = std::min<int>((int)m->audit_names.size(), max_n);
    for (int i = 0; i < n; ++i) exps[i] = m->audit_exp[i] ? *m->audit_exp[i] : kActExp;
    return n;
}
int artalk_get_audit(artalk_model* m, char* names_buf, int buf_len, float* values, int max_n) {
    if (!m || !names_buf || !values || buf_len <= 0 || max_n <= 0) return ARTALK_EINVAL;
    if (!m->audit_vals) return 0;
    (void)hipSetDevice(m->device);
    if (hipDeviceSynchronize() != hipSuccess) return ARTALK_EHIP;
    const int n = std::min<int>((int)m->audit_names.size(), max_n);
    std::vector<unsigned int> bits((size_t)std::max(n, 1));
    if (n && hipMemcpy(bits.data(), m->audit_vals, n * sizeof(unsigned int), hipMemcpyDeviceToHost) != hipSuccess) return ARTALK_EHIP;
    int pos = 0, written = 0;
    for (int i = 0; i < n; ++i) {
        const std::string& nm = m->audit_names[i];
        if (pos + (int)nm.size() + 1 > buf_len) break;
        std::memcpy(names_buf + pos, nm.c_str(), nm.size() + 1);
        pos += (int)nm.size() + 1;
        std::memcpy(&values[i], &bits[i], sizeof(float));
        ++written;
    }
    return written;
}

// Intermediate taps for the parity tests (SURVEY.md 8c "reference-captured intermediates"): while a tap buffer is set, artalk_infer runs
// the AR/VAE body eagerly as ONE clip group (no graphs) and copies, per chunk index j and clip position b (sorted order), into slot
// (j * max_batch + b) of `tap_dev`: the layout of artalk_tap_layout().  tap_dev = NULL switches it off.
int artalk_set_tap(artalk_model* m, float* tap_dev, int max_batch, int max_chunks) {
    if (!m || max_batch < 0 || max_chunks < 0) return ARTALK_EINVAL;
    (void)hipSetDevice(m->device); (void)hipDeviceSynchronize();
    m->tap = tap_dev; m->tap_B = tap_dev ? max_batch : 0; m->tap_maxch = tap_dev ? max_chunks : 0; m->tap_chunk = 0;
    return ARTALK_OK;
}
// out[0..6] = float offsets of the fields blk0_in, blk0_out, blkL_out, prev_in, logits, dec_out inside a slot, out[6] = floats per slot
int artalk_tap_layout(int64_t* out, int n) {
    if (!out || n < 7) return ARTALK_EINVAL;
    const int64_t v[7] = {TAP_BLK0_IN, TAP_BLK0_OUT, TAP_BLKL_OUT, TAP_PREV_IN, TAP_LOGITS, TAP_DEC_OUT, TAP_SLOT};
    for (int i = 0; i < 7; ++i) out[i] = v[i];
    return ARTALK_OK;
}

// Restrict the model to a set of compute units (bit i of mask = CU i / 8 of XCD i % 8 on MI355X): the streams the library creates
// itself (the second clip group of the AR/VAE body) get the mask, the persistent GEMM kernels size their grids to it.  The caller
// passes a stream with the same mask to artalk_infer (artalk_op_create_masked_stream).  n_words = 0 clears the mask.
int artalk_set_cu_mask(artalk_model* m, const uint32_t* mask, int n_words) {
    if (!m || n_words < 0 || (n_words > 0 && !mask)) return ARTALK_EINVAL;
    (void)hipSetDevice(m->device); (void)hipDeviceSynchronize();
    m->cu_mask.assign(mask, mask + n_words);
    int n = 0;
    for (uint32_t w : m->cu_mask) n += __builtin_popcount(w);
    if (n_words > 0 && n < 8) { m->cu_mask.clear(); m->n_cus = 0; return fail(m, ARTALK_EINVAL, "a CU partition needs at least 8 compute units"); }
    int dev_cus = 0;      // mask bits beyond the device's CUs select nothing: the grids must not count them
    if (hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, m->device) == hipSuccess && dev_cus > 0) n = std::min(n, dev_cus);
    m->n_cus = n;
    for (int i = 0; i < 3; ++i) {      // side streams are re-created (masked) on demand
        if (m->side_stream[i]) { (void)hipStreamDestroy(m->side_stream[i]); m->side_stream[i] = nullptr; }
        if (m->join_ev[i]) { (void)hipEventDestroy(m->join_ev[i]); m->join_ev[i] = nullptr; }
    }
    for (auto& g : m->graphs) (void)hipGraphExecDestroy(g.second.exec);      // the GEMM grids were captured for the old CU count
    m->graphs.clear();
    m->init_hist[0].valid = m->init_hist[1].valid = false;
    return ARTALK_OK;
}

int artalk_infer(artalk_model* m, const float* audio_dev, int64_t audio_clip_stride, const int64_t* n_chunks, int B,
                 const float* style_motion_dev, const uint8_t* has_style, float* out_motion_dev, int64_t out_clip_stride,
                 uint8_t* out_bits_dev, uint8_t* out_hist_bits_dev, float* out_w2v_dev, void* stream) {
    if (!m || !audio_dev || !n_chunks || !out_motion_dev || B <= 0) return ARTALK_EINVAL;
    if (!m->finalized) return fail(m, ARTALK_ESTATE, "artalk_infer before artalk_finalize_weights");
    (void)hipSetDevice(m->device);
    if (!stream && !m->own_stream) HIPCHK(m, hipStreamCreate(&m->own_stream));
    hipStream_t s = stream ? (hipStream_t)stream : m->own_stream;
    const artalk_config& c = m->cfg;
    int64_t C = 0, maxch = n_chunks[0];
    for (int b = 0; b < B; ++b) {
        if (n_chunks[b] <= 0 || (b > 0 && n_chunks[b] > n_chunks[b - 1])) return fail(m, ARTALK_EINVAL, "n_chunks must be positive and non-increasing");
        C += n_chunks[b];
    }
    bool encode_style = false;      // validated before anything is enqueued or any state changes
    if (style_motion_dev && has_style) {
        for (int b = 0; b < B; ++b) {
            if (has_style[b] > 2) return fail(m, ARTALK_EINVAL, "has_style entries must be 0, 1 or 2");
            encode_style |= has_style[b] == 1;
        }
    }
    if (m->tap && (B > m->tap_B || maxch > m->tap_maxch)) return fail(m, ARTALK_EINVAL, "the tap buffer (artalk_set_tap) is smaller than this call");
    if (B > m->ws.maxB || C > m->ws.maxC) { if (int rc = reserve(m, B, (int)C)) return rc; }
    Workspace& w = m->ws;
    // chunk list, chunk-index major: chunk (j, b) for all b with n_chunks[b] > j  -> active clips are a prefix
    artalk_model::Stage* stg = nullptr;
    if (int rc = next_stage(m, &stg)) return rc;
    long* src = stg->src;
    std::vector<int> base((size_t)maxch + 1, 0), Bj((size_t)maxch, 0);
    {
        int idx = 0;
        for (int64_t j = 0; j < maxch; ++j) {
            base[j] = idx;
            for (int b = 0; b < B && n_chunks[b] > j; ++b) { src[idx++] = (long)b * audio_clip_stride + (long)j * kSamplesPerChunk; Bj[j]++; }
        }
        base[maxch] = idx;
    }
    HIPCHK(m, hipMemcpyAsync(w.src_off, src, C * sizeof(long), hipMemcpyHostToDevice, s));
    if (style_motion_dev && has_style) {
        std::memcpy(stg->has, has_style, (size_t)B);
        HIPCHK(m, hipMemcpyAsync(w.has_style, stg->has, B, hipMemcpyHostToDevice, s));
    }
    HIPCHK(m, hipEventRecord(stg->done, s));   // pinned slot: no synchronisation, the slot is reused kStageSlots calls later
    stg->used = true;
    m->stream_B = 0;   // the batch call reuses the workspace that holds the streaming history
    HIPCHK(m, hipMemsetAsync(w.status, 0, 4 * sizeof(int), s));
    m->ev_used = 0; m->dom_events.clear(); m->marks.clear(); m->prof_stream = s;
    // level 3: every kernel of this call is launched with its own start / stop events (kernels.h ARTALK_LAUNCH) - on this thread, for the
    // duration of the call; the body then runs eagerly as ONE clip group, like level 2
    struct TimerScope { bool on; TimerScope(artalk_model* x) : on(x->profiling == 3) { if (on) { x->ktimer.used = 0; x->ktimer.recs.clear(); x->ktimer.bucket = KB_OTHER; g_ktimer = &x->ktimer; } }
                        ~TimerScope() { if (on) g_ktimer = nullptr; } } timer_scope(m);
    stage_mark(m, s, PB_OTHER);
    run_style(m, (style_motion_dev && has_style) ? style_motion_dev : nullptr, B, s, encode_style);
    stage_mark(m, s, PB_STYLE);
    // (wav2vec2 of chunk index j + 1 on a stream of its own beside the AR/VAE body of index j was an option until round 3: with the
    // persistent one-workgroup-per-CU GEMM kernels, whose workgroups hold a CU's LDS for a whole launch, the body's short kernels wait
    // behind them - 82 -> 128 ms per step; removed)
    for (int c0 = 0; c0 < C; c0 += w.G) run_wav2vec(m, audio_dev, c0, (int)std::min<int64_t>(w.G, C - c0), out_w2v_dev, s);
    // initial history: encode + quantise an all-zero motion (app/models.py:86-89) - the same for every clip: cached per model
    if (int rc = run_init_history(m, B, s)) return rc;
    const size_t bits_row = (size_t)kNTok * c.code_dim;
    if (out_hist_bits_dev)
        HIPCHK(m, hipMemcpy2DAsync(out_hist_bits_dev, (size_t)(maxch + 1) * bits_row, w.hist_bits, bits_row, bits_row, B,
                                   hipMemcpyDeviceToDevice, s));
    stage_mark(m, s, PB_VAE);
    const bool graphs = m->use_graphs && m->profiling < 2 && !m->audit && !m->tap;
    // AdaLN table of a chunk index for all blocks + head: SiLU(cond) @ [W_0;...;W_11;W_head]^T, in line in front of the body that reads it.
    // (The table of chunk index j + 1 on a stream of its own beside body j was measured in round 4 and removed in round 5: the bodies slow
    // down by what the table GEMM takes, profiles/r04_ada_overlap_sweep.log.)
    auto ada_table = [&](int64_t j) {
        roctxRangePushA("artalk.ar.adaln_table");
        kbucket(m, KB_ADA);
        linear(m, w.silu_cond + (long)base[j] * kNTok * kCond, kCond, m->ada_w, m->ada_b, w.ada, m->ada_n, Bj[j] * kNTok, m->ada_n, kCond,
               ACT_NONE, nullptr, s, m->precision == 1 ? LF_A_P8 : 0, nullptr, m->ex.silu_cond);
        roctxRangePop();
    };
    for (int64_t j = 0; j < maxch; ++j) {
        const int Bn = Bj[j];
        m->tap_chunk = (int)j;
        // clips whose motion must be re-encoded into a history: those with a chunk j + 1 (a prefix: clips are sorted by length) - all of
        // them when the caller asked for the history bits of every chunk (the reference computes the trailing ones too and drops them)
        const int n_next = out_hist_bits_dev ? Bn : (j + 1 < maxch ? Bj[j + 1] : 0);
        ada_table(j);
        stage_mark(m, s, PB_ADA);
        if (graphs) {
            Range r_body("artalk.body.graph");
            if (int brc = run_chunk_body_graphs(m, Bn, s, n_next)) return brc;
            stage_mark(m, s, PB_AR);   // light profiling: the whole captured body (AR steps + VAE) is charged to the AR bucket
        } else {
            if (int brc = run_chunk_body_split(m, Bn, s, n_next)) return brc;
        }
        const size_t mrow = (size_t)100 * c.motion_dim * 4;
        HIPCHK(m, hipMemcpy2DAsync(out_motion_dev + j * 100 * c.motion_dim, (size_t)out_clip_stride * 4, w.motion_chunk, mrow, mrow, Bn,
                                   hipMemcpyDeviceToDevice, s));
        if (out_bits_dev)
            HIPCHK(m, hipMemcpy2DAsync(out_bits_dev + j * bits_row, (size_t)maxch * bits_row, w.bits, bits_row, bits_row, Bn,
                                       hipMemcpyDeviceToDevice, s));
        if (out_hist_bits_dev)
            HIPCHK(m, hipMemcpy2DAsync(out_hist_bits_dev + (j + 1) * bits_row, (size_t)(maxch + 1) * bits_row, w.hist_bits, bits_row,
                                       bits_row, Bn, hipMemcpyDeviceToDevice, s));
    }
    stage_mark(m, s, PB_OTHER);
    if (int rc = publish_status(m, s)) return rc;
    HIPCHK(m, hipGetLastError());
    if (m->sticky_error) { m->sticky_error = false; return ARTALK_ESTATE; }
    return ARTALK_OK;
}

// Style-clip cache support: the style condition of app/models.py:67-73 depends on the style clip only,
// and the reference's engine keeps ONE style clip across many inference calls (inference.py:41-45,115-118).  This computes the
// conditions of n clips once; a later artalk_infer / artalk_stream_begin takes such a condition in place of the clip (flag 2).
int artalk_style_encode(artalk_model* m, const float* style_motion_dev, int n, float* out_cond_dev, void* stream) {
    if (!m || !style_motion_dev || !out_cond_dev || n <= 0) return ARTALK_EINVAL;
    if (!m->finalized) return fail(m, ARTALK_ESTATE, "artalk_style_encode before artalk_finalize_weights");
    (void)hipSetDevice(m->device);
    if (!stream && !m->own_stream) HIPCHK(m, hipStreamCreate(&m->own_stream));
    hipStream_t s = stream ? (hipStream_t)stream : m->own_stream;
    if (m->stream_B > 0) return fail(m, ARTALK_ESTATE, "artalk_style_encode during a streaming session (it shares the workspace)");
    if (n > m->ws.maxB) { if (int rc = reserve(m, n, std::max(n, m->ws.maxC))) return rc; }
    Workspace& w = m->ws;
    HIPCHK(m, hipMemsetAsync(w.has_style, 1, n, s));
    {
        ProfilingOff quiet(m);
        run_style(m, style_motion_dev, n, s, true);
    }
    HIPCHK(m, hipMemcpyAsync(out_cond_dev, w.style_cond, (size_t)n * kE * sizeof(float), hipMemcpyDeviceToDevice, s));
    HIPCHK(m, hipGetLastError());
    return ARTALK_OK;
}

// ---------------------------------------------------------------------------------- streaming
// The path already produces 4-second blocks causally (app/models.py:92-114: chunk j needs only the re-encoded motion of
// chunk j-1); these two calls expose that with the history kept in the model's workspace between calls.
int artalk_stream_begin(artalk_model* m, int B, const float* style_motion_dev, const uint8_t* has_style, void* stream) {
    if (!m || B <= 0) return ARTALK_EINVAL;
    if (!m->finalized) return fail(m, ARTALK_ESTATE, "artalk_stream_begin before artalk_finalize_weights");
    bool encode_style = false;
    if (style_motion_dev && has_style)
        for (int b = 0; b < B; ++b) {
            if (has_style[b] > 2) return fail(m, ARTALK_EINVAL, "has_style entries must be 0, 1 or 2");
            encode_style |= has_style[b] == 1;
        }
    (void)hipSetDevice(m->device);
    if (!stream && !m->own_stream) HIPCHK(m, hipStreamCreate(&m->own_stream));
    hipStream_t s = stream ? (hipStream_t)stream : m->own_stream;
    if (B > m->ws.maxB || B > m->ws.maxC) { if (int rc = reserve(m, B, B)) return rc; }
    if (int rc = ensure_stage(m, m->ws.maxB, m->ws.maxC)) return rc;
    Workspace& w = m->ws;
    if (style_motion_dev && has_style) {
        artalk_model::Stage* stg = nullptr;
        if (int rc = next_stage(m, &stg)) return rc;
        std::memcpy(stg->has, has_style, (size_t)B);
        HIPCHK(m, hipMemcpyAsync(w.has_style, stg->has, B, hipMemcpyHostToDevice, s));
        HIPCHK(m, hipEventRecord(stg->done, s));
        stg->used = true;
    }
    HIPCHK(m, hipMemsetAsync(w.status, 0, 4 * sizeof(int), s));   // the status word covers the whole streaming session
    {
        ProfilingOff quiet(m);
        run_style(m, (style_motion_dev && has_style) ? style_motion_dev : nullptr, B, s, encode_style);
        if (int rc = run_init_history(m, B, s)) return rc;
    }
    m->stream_B = B;
    if (int rc = publish_status(m, s)) return rc;
    HIPCHK(m, hipGetLastError());
    return ARTALK_OK;
}

// audio_dev: [B][chunk_stride] f32, the next 64000 samples of every stream (zero padded by the caller at the end of a
// clip); out_motion_dev: [B][out_stride] receives 100 x 106 codes per stream.
int artalk_stream_chunk(artalk_model* m, const float* audio_dev, int64_t chunk_stride, float* out_motion_dev, int64_t out_stride,
                        void* stream) {
    if (!m || !audio_dev || !out_motion_dev) return ARTALK_EINVAL;
    if (m->stream_B <= 0) return fail(m, ARTALK_ESTATE, "artalk_stream_chunk before artalk_stream_begin");
    (void)hipSetDevice(m->device);
    if (!stream && !m->own_stream) HIPCHK(m, hipStreamCreate(&m->own_stream));
    hipStream_t s = stream ? (hipStream_t)stream : m->own_stream;
    const int B = m->stream_B;
    Workspace& w = m->ws;
    if (B > w.maxB) return fail(m, ARTALK_ESTATE, "workspace was re-reserved since artalk_stream_begin; begin again");
    artalk_model::Stage* stg = nullptr;
    if (int rc = next_stage(m, &stg)) return rc;
    for (int b = 0; b < B; ++b) stg->src[b] = (long)b * chunk_stride;
    HIPCHK(m, hipMemcpyAsync(w.src_off, stg->src, B * sizeof(long), hipMemcpyHostToDevice, s));
    HIPCHK(m, hipEventRecord(stg->done, s));
    stg->used = true;
    ProfilingOff quiet(m);      // restored on every exit path
    for (int c0 = 0; c0 < B; c0 += w.G) run_wav2vec(m, audio_dev, c0, std::min(w.G, B - c0), nullptr, s);
    linear(m, w.silu_cond, kCond, m->ada_w, m->ada_b, w.ada, m->ada_n, B * kNTok, m->ada_n, kCond, ACT_NONE, nullptr, s,
           m->precision == 1 ? LF_A_P8 : 0, nullptr, m->ex.silu_cond);
    if (m->use_graphs) {
        if (int brc = run_chunk_body_graphs(m, B, s)) return brc;
    } else {
        if (int brc = run_chunk_body_split(m, B, s)) return brc;
    }
    const size_t mrow = (size_t)100 * m->cfg.motion_dim * 4;
    HIPCHK(m, hipMemcpy2DAsync(out_motion_dev, (size_t)out_stride * 4, w.motion_chunk, mrow, mrow, B, hipMemcpyDeviceToDevice, s));
    if (int rc = publish_status(m, s)) return rc;
    HIPCHK(m, hipGetLastError());
    return ARTALK_OK;
}

int artalk_stream_end(artalk_model* m) {
    if (!m) return ARTALK_EINVAL;
    m->stream_B = 0;      // the history in the workspace is dead; artalk_style_encode / artalk_reserve may use the workspace again
    return ARTALK_OK;
}

// Numerical health of the work enqueued since the last artalk_infer / artalk_stream_begin started.  Every call ends with an
// asynchronous copy of the device status word into pinned host memory; artalk_get_status waits for that copy (the only
// synchronisation on this boundary, and only if the caller asks), artalk_poll_status never blocks (ARTALK_EBUSY while the call is
// still running).  Bits: 0 = a logit was NaN/Inf (the pairwise argmax would have turned it into a 0 bit), 1 = a re-encoder
// output was NaN/Inf, 2 = a FLAME code was NaN/Inf, 3 = an activation left the range of the P8 split format at its producer.
// Non-zero in f16x3 mode means an activation left fp16's range: redo the call in f32 mode.
int artalk_get_status(artalk_model* m, int* flags, void* stream) {
    (void)stream;      // kept for ABI compatibility: the wait is on the library's own event, not on a stream
    if (!m || !flags) return ARTALK_EINVAL;
    if (!m->status_pending) { *flags = 0; return ARTALK_OK; }
    (void)hipSetDevice(m->device);
    const int slot = (int)((m->ticket - 1) % artalk_model::kStatusSlots);
    HIPCHK(m, hipEventSynchronize(m->status_ev[slot]));
    *flags = m->h_status[slot];
    return ARTALK_OK;
}
long long artalk_last_ticket(artalk_model* m) { return m ? m->ticket : -1; }
int artalk_get_status_of(artalk_model* m, long long ticket, int* flags) {
    if (!m || !flags || ticket <= 0 || ticket > m->ticket) return ARTALK_EINVAL;
    if (ticket + artalk_model::kStatusSlots <= m->ticket) { m->err = "artalk_get_status_of: that call's status slot has been reused"; return ARTALK_EINVAL; }
    if (!m->status_pending) { *flags = 0; return ARTALK_OK; }      // the workspace was re-created since: nothing of that call is left
    (void)hipSetDevice(m->device);
    const int slot = (int)((ticket - 1) % artalk_model::kStatusSlots);
    HIPCHK(m, hipEventSynchronize(m->status_ev[slot]));
    *flags = m->h_status[slot];
    return ARTALK_OK;
}
int artalk_poll_status(artalk_model* m, int* flags) {
    if (!m || !flags) return ARTALK_EINVAL;
    if (!m->status_pending) { *flags = 0; return ARTALK_OK; }
    (void)hipSetDevice(m->device);
    const int slot = (int)((m->ticket - 1) % artalk_model::kStatusSlots);
    const hipError_t e = hipEventQuery(m->status_ev[slot]);
    if (e == hipErrorNotReady) return ARTALK_EBUSY;
    if (e != hipSuccess) { m->err = std::string("hipEventQuery: ") + hipGetErrorString(e); return ARTALK_EHIP; }
    *flags = m->h_status[slot];
    return ARTALK_OK;
}

int artalk_get_profile(artalk_model* m, double* out, int n) {
    if (!m || !out || n < 10) return ARTALK_EINVAL;
    if (!m->profiling || m->marks.size() < 2) return fail(m, ARTALK_ESTATE, "profiling was not enabled for the last artalk_infer");
    (void)hipSetDevice(m->device);
    HIPCHK(m, hipStreamSynchronize(m->prof_stream));
    auto ms = [&](size_t a, size_t b) { float t = 0.f; (void)hipEventElapsedTime(&t, m->ev_pool[a], m->ev_pool[b]); return (double)t; };
    double r[10] = {0};
    for (size_t i = 1; i < m->marks.size(); ++i) {
        const int b = m->marks[i].first;
        if (b >= 0 && b < 6) r[b] += ms(m->marks[i - 1].second, m->marks[i].second);
    }
    r[6] = ms(m->marks.front().second, m->marks.back().second);
    for (auto& d : m->dom_events) { r[7] += 1.0; r[8] += ms(d.first, d.first + 1); r[9] += d.second; }
    for (int i = 0; i < 10; ++i) out[i] = r[i];
    return ARTALK_OK;
}

// Kernel-time budget of the last artalk_infer run at profiling level 3: out[b] = sum of the durations (ms) of the kernels launched in
// bucket b - 0 style, 1 conv stack, 2 encoder, 3 AdaLN tables, 4 history K/V + glue, 5..9 scale steps 0..4, 10 VAE decode, 11 re-encode,
// 12 other - and out[13] = number of kernels timed.  Durations are the dispatches' own begin / end timestamps (hipExtLaunchKernelGGL
// events), i.e. without launch gaps: stage event time minus this sum is the time the stage's stream sat idle.  n >= 14.
int artalk_get_kernel_sums(artalk_model* m, double* out, int n) {
    if (!m || !out || n < KB_N + 1) return ARTALK_EINVAL;
    if (m->ktimer.recs.empty()) return fail(m, ARTALK_ESTATE, "no kernel was timed (artalk_set_profiling(3) before artalk_infer)");
    (void)hipSetDevice(m->device);
    HIPCHK(m, hipDeviceSynchronize());
    for (int i = 0; i <= KB_N; ++i) out[i] = 0.0;
    for (const auto& r : m->ktimer.recs) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, m->ktimer.pool[r.second], m->ktimer.pool[r.second + 1]) != hipSuccess) continue;
        const int b = r.first >= 0 && r.first < KB_N ? r.first : KB_OTHER;
        out[b] += t;
        out[KB_N] += 1.0;
    }
    return ARTALK_OK;
}

int artalk_savgol(artalk_model* m, const float* in_dev, float* out_dev, int T, void* stream) {
    if (!m || !in_dev || !out_dev) return ARTALK_EINVAL;
    if (T < 9) return fail(m, ARTALK_EINVAL, "savgol (mode='interp') needs window_length <= T (T >= 9)");
    (void)hipSetDevice(m->device);
    launch_savgol(in_dev, out_dev, T, m->cfg.motion_dim, (hipStream_t)stream);
    return ARTALK_OK;
}

// ---------------------------------------------------------------------------------- single-kernel entry points
int artalk_op_gemm(const float* A, int64_t lda, const float* W, const float* bias, const float* gate, const float* R, float* C,
                   int M, int N, int K, int act, void* stream) {
    if (!A || !W || !C || K % 32 != 0 || M < 0 || N <= 0) return ARTALK_EINVAL;
    GemmArgs g;
    g.A = A; g.lda = lda; g.W = W; g.ldw = K; g.bias = bias; g.C = C; g.ldc = N; g.gate = gate; g.ldg = N; g.R = R; g.ldr = N;
    g.M = M; g.N = N; g.K = K; g.act = act;
    launch_gemm(g, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? ARTALK_OK : ARTALK_EHIP;
}

int artalk_op_gemm_ex(const float* A, int64_t lda, const float* W, const float* bias, float* C, int M, int N, int K, int act,
                      int force_cfg, void* stream) {
    if (!A || !W || !C || K % 32 != 0 || M < 0 || N <= 0) return ARTALK_EINVAL;
    GemmArgs g;
    g.A = A; g.lda = lda; g.W = W; g.ldw = K; g.bias = bias; g.C = C; g.ldc = N; g.M = M; g.N = N; g.K = K; g.act = act;
    g.force_cfg = force_cfg;
    launch_gemm(g, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? ARTALK_OK : ARTALK_EHIP;
}

// out_dev: blocks*256 floats of scratch; *flops receives the FLOPs of the launch (time it with events on `stream`)
int artalk_op_mfma_f32_peak(float* out_dev, int blocks, int iters, int nacc, double* flops, void* stream) {
    if (!out_dev || !flops || blocks <= 0 || iters <= 0) return ARTALK_EINVAL;
    *flops = launch_mfma_f32_peak(out_dev, blocks, iters, nacc, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? ARTALK_OK : ARTALK_EHIP;
}

// f16x3 split GEMM on fp32 inputs (W is packed into a temporary): same contract as artalk_op_gemm_ex; cfg 0: 128x128, 1: 64x64, -1: heuristic
int artalk_op_gemm_f16s(const float* A, int64_t lda, const float* W, const float* bias, float* C, int M, int N, int K, int act,
                        int force_cfg, void* stream) {
    if (!A || !W || !C || K % 32 != 0 || M <= 32 || N <= 0) return ARTALK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    unsigned int* wp = nullptr;
    if (hipMalloc(&wp, (size_t)N * K * 4) != hipSuccess) return ARTALK_EHIP;
    launch_pack_split(W, wp, (long)N * K, true, s);
    GemmArgs g;
    g.A = A; g.lda = lda; g.W = W; g.Wp = wp; g.ldw = K; g.bias = bias; g.C = C; g.ldc = N; g.M = M; g.N = N; g.K = K; g.act = act;
    g.force_cfg = force_cfg & 0xff;
    unsigned int* ap = nullptr;
    if (force_cfg >= 0 && (force_cfg & 0x100)) {   // tuning: pre-packed A (what a fused producer would hand over)
        if (hipMalloc(&ap, (size_t)M * lda * 4) != hipSuccess) return ARTALK_EHIP;
        launch_pack_split(A, ap, (long)M * lda, false, s);
        g.A = reinterpret_cast<const float*>(ap); g.a_packed = 1;
    }
    launch_gemm_f16s(g, s);
    (void)hipStreamSynchronize(s);
    (void)hipFree(wp);
    if (ap) (void)hipFree(ap);
    return hipGetLastError() == hipSuccess ? ARTALK_OK : ARTALK_EHIP;
}

// split-K scratch of the tuning entry point below (never used by the model path): the current buffer and the ones it outgrew
static float* g_op_scratch = nullptr;
static size_t g_op_scratch_cap = 0;
static std::vector<float*> g_op_retired;
// Frees every scratch buffer of artalk_op_gemm_f16s_packed (current and retired).  The caller guarantees that no captured graph
// that replays such a launch is used afterwards (tools/* destroy their graphs first; pytest calls it at session end).
int artalk_op_release_scratch(void) {
    if (hipDeviceSynchronize() != hipSuccess) return ARTALK_EHIP;
    for (float* p : g_op_retired) (void)hipFree(p);
    g_op_retired.clear();
    if (g_op_scratch) (void)hipFree(g_op_scratch);
    g_op_scratch = nullptr; g_op_scratch_cap = 0;
    return ARTALK_OK;
}
// building blocks for tuning the split GEMM without allocation noise: pack once, then launch on packed operands
int artalk_op_pack_split(const float* in, void* out_u32, int64_t n, int is_weight, void* stream) {
    if (!in || !out_u32 || n <= 0) return ARTALK_EINVAL;
    launch_pack_split(in, (unsigned int*)out_u32, n, is_weight != 0, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? ARTALK_OK : ARTALK_EHIP;
}
int artalk_op_gemm_f16s_packed(const void* A, int a_packed, int64_t lda, const void* Wp, const float* bias, float* C, int M, int N,
                               int K, int act, int force_cfg, void* stream) {
    if (!A || !Wp || !C || K % 32 != 0 || M <= 0 || N <= 0 || (M <= 32 && (force_cfg & 0xff) < 20)) return ARTALK_EINVAL;
    GemmArgs g;
    g.A = (const float*)A; g.a_packed = a_packed; g.lda = lda; g.W = nullptr; g.Wp = (const unsigned int*)Wp; g.ldw = K; g.bias = bias;
    g.C = C; g.ldc = N; g.M = M; g.N = N; g.K = K; g.act = act & 0xff; g.force_cfg = force_cfg;
    g.c_p8 = (act >> 8) & 1;      // tuning: bit 8 of `act` = result in the P8 split format (same pitch)
    if ((act >> 9) & 1) { g.R = C; g.ldr = N; }      // tuning: bit 9 = residual read from C (in place, as the encoder's out-projection / FFN-out run)
    if (force_cfg >= 2) {   // LDS-DMA kernels (both operands in P8): 7 / 12 = persistent 256x256 / 320x256 tiles, 8 = persistent 128x128, 13 = non-persistent 256x256 (17: with wall-clock stamps)
        if (!a_packed) return ARTALK_EINVAL;
        g.force_cfg = force_cfg == 99 ? -1 : force_cfg;   // 99: the engine's own choice between the production kernels
        if (force_cfg == 17) { g.partial = (float*)bias; g.bias = nullptr; }     // (17: gemm_p8_256_kernel with stamps)   // timing build: `bias` carries the stamp buffer (8 x u64 per tile)
        if (force_cfg != 99 && (force_cfg & 0xff) >= 20) {     // 20 / 23 / 24: small-grid LDS-DMA kernel; bits 8-15: split-K factor (slabs in a temporary)
            g.force_cfg = force_cfg & 0xff;
            g.w_nt = (force_cfg >> 16) & 1;      // tuning: bit 16 = non-temporal weight pieces
            const int S = (force_cfg >> 8) & 0xff;
            if (S > 1) {
                const size_t need = (size_t)S * M * N * 4;
                if (need > g_op_scratch_cap) {
                    // a smaller scratch is NOT freed here: graphs captured from earlier calls (tools/gemm_f16s_bench.py replays one per
                    // variant) still write to it - freeing it turned their replay into a memory fault.  It is RETIRED instead and freed by
                    // artalk_op_release_scratch(), which the owner of those graphs calls once they are gone.  At least 64 MB, so that it rarely grows.
                    const size_t cap = std::max(need, (size_t)64 << 20);
                    float* fresh = nullptr;
                    if (hipMalloc(&fresh, cap) != hipSuccess) return ARTALK_EHIP;
                    if (g_op_scratch) g_op_retired.push_back(g_op_scratch);
                    g_op_scratch = fresh; g_op_scratch_cap = cap;
                }
                g.splitk = S; g.partial = g_op_scratch;
            }
            launch_gemm_p8_sm(g, (hipStream_t)stream);
            if (S > 1) launch_splitk_reduce(g, (hipStream_t)stream);
        } else {
            launch_gemm_p8(g, (hipStream_t)stream);
        }
    } else {
        launch_gemm_f16s(g, (hipStream_t)stream);
    }
    return hipGetLastError() == hipSuccess ? ARTALK_OK : ARTALK_EHIP;
}

// A HIP stream restricted to a set of compute units (hipExtStreamCreateWithCUMask): bit i of the mask = CU (i / 8) of XCD (i % 8) on
// MI355X (tools/cu_mask_probe.hip), so the low / high 128 bits are the two halves of every XCD.  The stream a model restricted with
// artalk_set_cu_mask is driven on.
int artalk_op_create_masked_stream(const uint32_t* mask, int n_words, void** out_stream) {
    if (!mask || !out_stream || n_words <= 0) return ARTALK_EINVAL;
    hipStream_t s = nullptr;
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, mask) != hipSuccess) return ARTALK_EHIP;
    *out_stream = s;
    return ARTALK_OK;
}
int artalk_op_destroy_stream(void* stream) { return hipStreamDestroy((hipStream_t)stream) == hipSuccess ? ARTALK_OK : ARTALK_EHIP; }

int artalk_op_gemm_p8_plan(int M, int N, int K, int residual) {
    GemmArgs g;
    g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldc = N;
    if (residual) { g.R = reinterpret_cast<const float*>(16); g.ldr = N; }     // only its presence and alignment are looked at
    const int v = gemm_p8_variant(g);
    return v == 1 ? 7 : (v == 2 ? 12 : 8);
}

int artalk_op_layernorm(const float* X, float* Y, const float* w, const float* b, const float* scale, const float* shift, int M,
                        int D, float eps, int act, void* stream) {
    if (!X || !Y || (D != 128 && D != 512 && D != 768 && D != 1024)) return ARTALK_EINVAL;
    LnArgs a;
    a.X = X; a.ldx = D; a.Y = Y; a.ldy = D; a.w = w; a.b = b; a.scale = scale; a.shift = shift; a.ldm = D; a.M = M; a.D = D;
    a.eps = eps; a.act = act & 0xff; a.out_p8 = (act & 0x100) ? 1 : 0;   // act | 0x100: write Y in the P8 split format
    launch_layernorm(a, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? ARTALK_OK : ARTALK_EHIP;
}

int artalk_op_attention(const float* Q, const float* K, const float* V, float* O, int B, int H, int HD, int Lq, int Lk, float scale,
                        int l2norm, const float* qscale, int split, void* stream) {
    if (!Q || !K || !V || !O || (HD != 64 && HD != 32) || ((l2norm & 1) && !qscale)) return ARTALK_EINVAL;
    AttnArgs a;
    a.split16 = (l2norm >> 1) & 1;      // l2norm | 2: the fp16 operand-split kernel of f16x3 mode
    a.qkv_p8 = (l2norm >> 2) & 1;       // l2norm | 4 (with | 2, without L2 norm): Q, K, V are given in the P8 split format
    l2norm &= 1;
    const long D = (long)H * HD;
    a.Q = Q; a.K = K; a.V = V; a.O = O; a.ldq = a.ldk = a.ldv = a.ldo = D;
    a.q_bstride = a.o_bstride = (long)Lq * D; a.k_bstride = a.v_bstride = (long)Lk * D;
    a.B = B; a.H = H; a.HD = HD; a.Lq = Lq; a.Lk = Lk; a.scale = scale; a.l2norm = l2norm; a.qscale = qscale;
    a.split_q = split; a.split_k = split;
    launch_attention(a, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? ARTALK_OK : ARTALK_EHIP;
}

int artalk_op_w2v_front(const float* audio, int C, int n, const float* w, const float* bias, const float* lnw, const float* lnb,
                        float* xnorm_out, float* Y, void* stream) {
    if (!audio || !xnorm_out || !Y || C <= 0 || n < 10) return ARTALK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    std::vector<long> off(C);
    for (int i = 0; i < C; ++i) off[i] = (long)i * n;
    long* doff = nullptr;
    if (hipMalloc(&doff, C * sizeof(long)) != hipSuccess) return ARTALK_EHIP;
    (void)hipMemcpy(doff, off.data(), C * sizeof(long), hipMemcpyHostToDevice);
    const int T = (n - 10) / 5 + 1;
    launch_audio_normalize(audio, doff, xnorm_out, C, n, s);
    launch_conv0(xnorm_out, n, w, bias, lnw, lnb, Y, C, T, T, s);
    (void)hipStreamSynchronize(s);
    (void)hipFree(doff);
    return hipGetLastError() == hipSuccess ? ARTALK_OK : ARTALK_EHIP;
}

int artalk_op_resample_mean(const float* x, int nch, int n, const float* taps, int orig, int nw, int width, float* out, int n_out,
                            void* stream) {
    if (!x || !taps || !out || nch <= 0 || n <= 0 || orig <= 0 || nw <= 0 || width < 0) return ARTALK_EINVAL;
    launch_resample_mean(x, nch, n, taps, orig, nw, width, out, n_out, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? ARTALK_OK : ARTALK_EHIP;
}

int artalk_op_pool_silu(const float* X, int C, int T, int D, float* Y, void* stream) {
    if (!X || !Y || D % 4 != 0) return ARTALK_EINVAL;
    static const int pn[5] = {1, 5, 25, 50, 100};
    launch_pool_silu(X, T, T, Y, C, pn, 5, D, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? ARTALK_OK : ARTALK_EHIP;
}

int artalk_op_bsq_history(const float* enc_out, uint8_t* hist_bits, float* prev_fdec, float* msfeat, int B, void* stream) {
    if (!enc_out || !hist_bits || !prev_fdec || !msfeat || B <= 0) return ARTALK_EINVAL;
    if (init_ms_tables() != 0) return ARTALK_EHIP;
    launch_bsq_history(enc_out, hist_bits, prev_fdec, msfeat, B, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? ARTALK_OK : ARTALK_EHIP;
}

// ---------------------------------------------------------------------------------- FLAME vertices
struct artalk_flame {
    int device = 0, V = 0, NB = 0, NBp = 0, P = 0, Pp = 0, cap = 0;
    float scale = 1.f;
    float *v_template = nullptr, *shapedirs = nullptr, *posedirs = nullptr, *jreg = nullptr, *weights = nullptr;
    int* parents = nullptr;
    float *betas = nullptr, *feat = nullptr, *rot = nullptr, *J = nullptr, *vshaped = nullptr, *vposed = nullptr;
    std::vector<void*> allocs, ws;
    std::string err;
};

static thread_local std::string g_flame_error;
const char* artalk_flame_last_error(const artalk_flame* f) { return f ? f->err.c_str() : g_flame_error.c_str(); }

void artalk_flame_destroy(artalk_flame* f) {
    if (!f) return;
    (void)hipSetDevice(f->device);
    (void)hipDeviceSynchronize();
    for (void* p : f->allocs) if (p) (void)hipFree(p);
    for (void* p : f->ws) if (p) (void)hipFree(p);
    delete f;
}

// Host arrays in the layouts of the reference's FLAMEModel buffers (app/flame_model/FLAME.py:36-45):
// v_template [V][3], shapedirs [V][3][NB], posedirs_t [V*3][P] (= the reference buffer transposed), J_regressor [5][V],
// parents [5] (parents[0] = -1), lbs_weights [V][5].
int artalk_flame_create(int device_id, int V, int NB, int P, const float* v_template, const float* shapedirs, const float* posedirs_t,
                        const float* J_regressor, const int32_t* parents, const float* lbs_weights, float scale, artalk_flame** out) {
    if (!out || !v_template || !shapedirs || !posedirs_t || !J_regressor || !parents || !lbs_weights || V <= 0 || NB <= 0 || P != 36) {
        g_flame_error = "bad argument (FLAME has 5 joints: posedirs must have 36 rows)";
        return ARTALK_EINVAL;
    }
    if (parents[0] != -1) { g_flame_error = "parents[0] must be -1"; return ARTALK_EINVAL; }
    for (int j = 1; j < 5; ++j) if (parents[j] < 0 || parents[j] >= j) { g_flame_error = "parents must form a tree in topological order"; return ARTALK_EINVAL; }
    if (hipSetDevice(device_id) != hipSuccess) { g_flame_error = "hipSetDevice failed"; return ARTALK_EHIP; }
    artalk_flame* f = new artalk_flame();
    f->device = device_id; f->V = V; f->NB = NB; f->P = P; f->scale = scale;
    f->NBp = (NB + 31) / 32 * 32; f->Pp = 64;
    auto up = [&](const float* h, int64_t rows, int64_t cols, int64_t cols_pad) -> float* {
        float* d = dalloc_in<float>(f->allocs, rows * cols_pad);
        if (!d) return nullptr;
        if (cols == cols_pad) { (void)hipMemcpy(d, h, rows * cols * 4, hipMemcpyHostToDevice); }
        else (void)hipMemcpy2D(d, cols_pad * 4, h, cols * 4, cols * 4, rows, hipMemcpyHostToDevice);
        return d;
    };
    f->v_template = up(v_template, 1, (int64_t)V * 3, (int64_t)V * 3);
    f->shapedirs = up(shapedirs, (int64_t)V * 3, NB, f->NBp);       // GEMM weight [N = 3V][K = NB padded]
    f->posedirs = up(posedirs_t, (int64_t)V * 3, P, f->Pp);
    f->jreg = up(J_regressor, 5, V, V);
    f->weights = up(lbs_weights, V, 5, 5);
    f->parents = dalloc_in<int>(f->allocs, 5);
    for (void* p : f->allocs) if (!p) { g_flame_error = "hipMalloc failed"; artalk_flame_destroy(f); return ARTALK_EHIP; }
    (void)hipMemcpy(f->parents, parents, 5 * sizeof(int), hipMemcpyHostToDevice);
    if (hipDeviceSynchronize() != hipSuccess) { g_flame_error = "upload failed"; artalk_flame_destroy(f); return ARTALK_EHIP; }
    *out = f;
    return ARTALK_OK;
}

// betas_dev [T][NB] (shape ++ expression), full_pose_dev [T][15] (global, neck, jaw, eye_l, eye_r axis-angles: FLAME.py:131-136)
// -> out_dev [T][V][3] = vertices * scale
int artalk_flame_verts(artalk_flame* f, const float* betas_dev, const float* full_pose_dev, int T, float* out_dev, void* stream) {
    if (!f || !betas_dev || !full_pose_dev || !out_dev || T <= 0) return ARTALK_EINVAL;
    (void)hipSetDevice(f->device);
    hipStream_t s = (hipStream_t)stream;
    if (T > f->cap) {
        (void)hipDeviceSynchronize();
        for (void* p : f->ws) if (p) (void)hipFree(p);
        f->ws.clear();
        const int64_t N = (int64_t)f->V * 3;
        f->betas = dalloc_in<float>(f->ws, (int64_t)T * f->NBp);
        f->feat = dalloc_in<float>(f->ws, (int64_t)T * f->Pp);
        f->rot = dalloc_in<float>(f->ws, (int64_t)T * 45);
        f->J = dalloc_in<float>(f->ws, (int64_t)T * 15);
        f->vshaped = dalloc_in<float>(f->ws, T * N);
        f->vposed = dalloc_in<float>(f->ws, T * N);
        for (void* p : f->ws) if (!p) { f->err = "hipMalloc failed"; f->cap = 0; return ARTALK_EHIP; }
        (void)hipDeviceSynchronize();   // zero fills run on the null stream
        f->cap = T;
    }
    const int N = f->V * 3;
    // betas into the K-padded GEMM operand (pad columns stay zero from allocation)
    if (hipMemcpy2DAsync(f->betas, (size_t)f->NBp * 4, betas_dev, (size_t)f->NB * 4, (size_t)f->NB * 4, T, hipMemcpyDeviceToDevice, s) != hipSuccess) {
        f->err = "copy failed"; return ARTALK_EHIP;
    }
    GemmArgs g;     // v_shaped = v_template + betas . shapedirs^T      (exact fp32 MFMA)
    g.A = f->betas; g.lda = f->NBp; g.W = f->shapedirs; g.ldw = f->NBp; g.C = f->vshaped; g.ldc = N; g.R = f->v_template; g.ldr = 0;
    g.M = T; g.N = N; g.K = f->NBp;
    launch_gemm(g, s);
    launch_flame_joints(f->vshaped, f->jreg, f->J, T, f->V, s);
    launch_flame_pose(full_pose_dev, f->rot, f->feat, T, f->Pp, s);
    GemmArgs p;     // v_posed = v_shaped + pose_feature . posedirs
    p.A = f->feat; p.lda = f->Pp; p.W = f->posedirs; p.ldw = f->Pp; p.C = f->vposed; p.ldc = N; p.R = f->vshaped; p.ldr = N;
    p.M = T; p.N = N; p.K = f->Pp;
    launch_gemm(p, s);
    launch_flame_skin(f->vposed, f->rot, f->J, f->parents, f->weights, out_dev, T, f->V, f->scale, s);
    if (hipGetLastError() != hipSuccess) { f->err = "kernel launch failed"; return ARTALK_EHIP; }
    return ARTALK_OK;
}

}  // extern "C"
