// fp32 attention on the gfx950 matrix cores (v_mfma_f32_16x16x4_f32), used by all four attention sites:
//   wav2vec2 encoder  hf:467-548            16 heads x 64, T=199, scale 1/8, no mask
//   AR blocks         app/transformer.py:65-79   12 heads x 64, L2-normalised q,k, per-head learned scale, SDPA scale 1.
//                     With the per-layer KV cache the keys present (181 history + levels <= current) are exactly the
//                     visible set of app/models.py:123-135, so no bias tensor is needed.
//   VAE blocks        app/modules/bitwise_vae.py:194-215  8 heads x 64, scale = 512^-0.5, mask: rows<100 see keys<100
//   style encoder     nn.MultiheadAttention  4 heads x 32
//
// One workgroup = 4 wavefronts x 16 queries.  K/V blocks of 64 keys are staged in LDS (K rows padded to
// HD+8, V rows to HD+4 floats: conflict-free for the fragment reads below) and shared by the 4 waves.
// Each wave computes S^T = K Q^T (key on the accumulator rows, query on the lane), so that
//  * the softmax statistics of a query are reduced over 4 registers and two cross-lane steps (xor 16, 32),
//  * the S^T accumulator is directly the B operand of O^T += V^T P^T (guide section 3 "accumulator as next
//    operand"): k-slot g of step j carries key 4g+j, and the V fragment is read with the same permutation.
#include <cstdlib>
#include <type_traits>
#include "common.h"

namespace artalk {

template <int HD>
__global__ __launch_bounds__(256, 2) void attention_kernel(const AttnArgs a) {
    constexpr int KB = 64;
    constexpr int KLD = HD + 8, VLD = HD + 16;   // V rows: lane r reads NDT consecutive floats (one ds_read_b128 for HD = 64)
    constexpr int NC = HD / 16;    // 16-byte chunks of a q/k row held per lane (chunk index g + 4c)
    constexpr int NDT = HD / 16;   // 16-wide d tiles of O^T
    constexpr int TPR = HD / 4;    // staging threads per row
    __shared__ __attribute__((aligned(16))) float Ks[KB * KLD];
    __shared__ __attribute__((aligned(16))) float Vs[KB * VLD];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 15, g = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y;
    const int qblk0 = blockIdx.x * 64;
    const int q0 = qblk0 + wave * 16;
    const int qi = q0 + r;
    const bool qvalid = qi < a.Lq;
    const bool wave_active = q0 < a.Lq;

    // ---- Q fragment (B operand of S^T): lane (r,g) holds Q[qi][4(g+4c)+e] ----
    f32x4 qf[NC];
    {
        const float* qp = a.Q + (long)b * a.q_bstride + (long)qi * a.ldq + h * HD;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            qf[c] = qvalid ? *reinterpret_cast<const f32x4*>(qp + 4 * (g + 4 * c)) : z;
        }
        if (a.l2norm) {
            float ss = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) ss += qf[c][e] * qf[c][e];
            ss += lane_xor<16>(ss);
            ss += lane_xor<32>(ss);
            const float den = fmaxf(sqrtf(ss), 1e-12f);
            const float mul = a.qscale[h];
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) qf[c][e] = (qf[c][e] / den) * mul;
        }
        if (a.scale != 1.0f) {
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) qf[c][e] *= a.scale;
        }
    }
    const int klim = (a.split_q > 0 && qi < a.split_q) ? a.split_k : a.Lk;   // this query sees keys < klim
    const int qblk_last = min(qblk0 + 63, a.Lq - 1);
    const int lk_wg = (a.split_q > 0 && qblk_last < a.split_q) ? a.split_k : a.Lk;

    float m_run = -INFINITY, l_part = 0.f;
    f32x4 ot[NDT];
#pragma unroll
    for (int d = 0; d < NDT; ++d) { f32x4 z = {0.f, 0.f, 0.f, 0.f}; ot[d] = z; }

    const float* Kb = a.K + (long)b * a.k_bstride + h * HD;
    const float* Vb = a.V + (long)b * a.v_bstride + h * HD;

    for (int kb0 = 0; kb0 < lk_wg; kb0 += KB) {
        __syncthreads();
        // ---- stage K (optionally L2-normalised) and V rows kb0..kb0+63; rows >= Lk are zero ----
#pragma unroll
        for (int i = 0; i < KB * TPR / 256; ++i) {
            const int idx = tid + i * 256;
            const int row = idx / TPR, c4 = (idx % TPR) * 4;
            const int kr = kb0 + row;
            f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
            if (kr < a.Lk) {
                kv = *reinterpret_cast<const f32x4*>(Kb + (long)kr * a.ldk + c4);
                vv = *reinterpret_cast<const f32x4*>(Vb + (long)kr * a.ldv + c4);
            }
            if (a.l2norm) {
                float ss = kv[0] * kv[0] + kv[1] * kv[1] + kv[2] * kv[2] + kv[3] * kv[3];
                ss = group_sum<TPR>(ss);
                const float den = fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
                for (int e = 0; e < 4; ++e) kv[e] = kv[e] / den;
            }
            *reinterpret_cast<f32x4*>(Ks + row * KLD + c4) = kv;
            *reinterpret_cast<f32x4*>(Vs + row * VLD + c4) = vv;
        }
        __syncthreads();
        if (!wave_active) continue;

        // ---- S^T tiles: st[t][j] = S[key kb0+16t+4g+j][query qi]; 16-key tiles entirely past Lk are skipped (wave-uniform) ----
        const int ntile = min(4, (a.Lk - kb0 + 15) >> 4);
        f32x4 st[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (t < ntile)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const f32x4 kf = *reinterpret_cast<const f32x4*>(Ks + (t * 16 + r) * KLD + 4 * (g + 4 * c));
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[e], qf[c][e], acc, 0, 0, 0);
            }
            st[t] = acc;
        }
        // ---- online softmax over this key block (per query = per lane column) ----
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kidx = kb0 + t * 16 + 4 * g + j;
                const float sv = (kidx < klim) ? st[t][j] : -INFINITY;
                st[t][j] = sv;
                mx = fmaxf(mx, sv);
            }
        mx = fmaxf(mx, lane_xor<16>(mx));
        mx = fmaxf(mx, lane_xor<32>(mx));
        const float m_new = fmaxf(m_run, mx);
        const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = expf(m_run - m_safe);
        m_run = m_new;
        float ps = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float p = expf(st[t][j] - m_safe);
                st[t][j] = p;
                ps += p;
            }
        l_part = l_part * alpha + ps;
#pragma unroll
        for (int d = 0; d < NDT; ++d) ot[d] *= alpha;
        // ---- O^T += V^T P^T: step (t,j): k-slot g <-> key 16t+4g+j; B = st[t][j].  Output row i of d-tile dd is d = NDT*i + dd,
        // so lane r's A values for the NDT tiles are V[key][NDT*r .. NDT*r+NDT-1]: one vector LDS read instead of NDT scalar ones.
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (t < ntile)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float* vrow = Vs + (t * 16 + 4 * g + j) * VLD + NDT * r;
                float vv[NDT];
                if constexpr (NDT == 4) {
                    const f32x4 tv = *reinterpret_cast<const f32x4*>(vrow);
#pragma unroll
                    for (int d = 0; d < NDT; ++d) vv[d] = tv[d];
                } else {
                    const float2 tv = *reinterpret_cast<const float2*>(vrow);
                    vv[0] = tv.x; vv[NDT - 1] = tv.y;
                }
#pragma unroll
                for (int d = 0; d < NDT; ++d)
                    ot[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[d], st[t][j], ot[d], 0, 0, 0);
            }
    }
    if (!wave_active) return;
    float l = l_part;
    l += lane_xor<16>(l);
    l += lane_xor<32>(l);
    if (qvalid) {
        const float inv = 1.0f / l;
        float* op = a.O + (long)b * a.o_bstride + (long)qi * a.ldo + h * HD;
        // ot[dd][reg] = O[d = NDT*(4g+reg) + dd]: for each reg the NDT tiles hold NDT consecutive d
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int d0 = NDT * (4 * g + reg);
            if constexpr (NDT == 4) {
                const float o0 = ot[0][reg] * inv, o1 = ot[1][reg] * inv, o2 = ot[2][reg] * inv, o3 = ot[3][reg] * inv;
                if (a.out_p8) store_p8x4(op, d0, o0, o1, o2, o3, a.status, a.o_exp);   // op is 32-byte aligned (HD % 8 == 0)
                else { f32x4 o = {o0, o1, o2, o3}; *reinterpret_cast<f32x4*>(op + d0) = o; }
            } else {
                *reinterpret_cast<float2*>(op + d0) = make_float2(ot[0][reg] * inv, ot[NDT - 1][reg] * inv);
            }
        }
    }
}

// ---- short-query variant (AR scale steps: 1..50 queries against 182..312 keys).  In the kernel above such a launch has
// one to four active waves per workgroup and walks the keys serially through LDS behind two barriers per 64 keys: 28 us per launch,
// all latency.  Here ONE 16-query tile is shared by the 4 waves of a workgroup and the KEYS are split over them (16-key tiles
// t = w, w+4, ...): K and V fragments go straight from global memory into MFMA operand registers (their lane maps are
// row-contiguous: 64-byte pieces of a key row for K, whole 256-byte rows for V), the next tile's loads are issued before
// the current tile's MFMAs, no LDS and no barrier in the loop; the four partial (m, l, O) are merged through LDS at the end.
template <int HD, bool SLABS = false>      // SLABS: AttnArgs::slabs is set (its own instantiation: the slab path costs 27 registers, and the
                                           // 25- / 50-query steps, which never take it, need three workgroups per CU)
__global__ __launch_bounds__(256, 2) void attention_short_kernel(const AttnArgs a) {      // (2 blocks per CU: at most 256 registers per wave, which is what makes hipcc keep the MFMA accumulators in VGPRs - with the full 512 it put them in AGPRs and moved them out and back around the softmax and the rescale, 123 v_accvgpr moves and 66 hazard nops per tile pair)
    static_assert(HD == 64, "head dim");
    constexpr int NC = 4, NDT = 4;
    __shared__ __attribute__((aligned(16))) float Os[4][16][HD + 4];
    __shared__ float Ms[4][16], Ls[4][16];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 15, g = lane >> 4;
    // 1-D grid, workgroup id -> (query tile, head, clip) so that the query tiles of one (clip, head) run on ONE XCD (workgroup id % 8
    // selects the XCD): they read the same K and V rows, and spread over XCDs each pulled its own copy into its own L2 - at the
    // 50-query scale step 4 tiles x 192 heads x 134 KB = 103 MB of fabric traffic per launch, which was what bounded it (24.5 us).
    // Pairs go in groups of 8 (one per XCD), the query tile is the slower index inside a group.
    const int nq = (a.Lq + 15) >> 4;
    const int grp = (int)blockIdx.x / (8 * nq), rem = (int)blockIdx.x - grp * 8 * nq;
    const int pair = grp * 8 + (rem & 7);
    if (pair >= a.H * a.B) return;      // (the grid is rounded up to whole groups; uniform per workgroup, before any barrier)
    const int b = pair / a.H, h = pair - b * a.H;
    const int q0 = (rem >> 3) * 16;
    const int qi = q0 + r;
    const bool qvalid = qi < a.Lq;

    // Lq <= 16 with split-K slabs (AttnArgs::slabs): the q | k | v rows of the NEW tokens (the queries, and the last Lq keys) are not in
    // the cache yet - a lane that needs four of their values sums them from the slabs itself, in splitk_reduce_kernel's order (0 + slab 0
    // + slab 1 + ... + bias), straight into the fragment register the cache row would have been loaded into: no reduce launch, no trip
    // through memory.  The lanes that own a new K / V row also write it to the cache (raw, before the L2 norm) for the later scale steps.
    const int E = a.H * HD;
    const int new0 = a.Lk - a.Lq;                                  // first new key row
    auto from_slabs = [&](int t, int col) {                        // row t (0 .. Lq-1) of this clip, 4 columns from `col` of q | k | v
        // every load of a batch of 8 slabs is issued before the first add (the slabs were written a moment ago by workgroups on other XCDs:
        // each load is an L2 miss, and a run-time loop pays those latencies one after the other); the adds keep the slab order
        const float* p = a.slabs + ((long)b * a.Lq + t) * a.slab_ld + col;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const f32x4 bias = a.slab_bias ? *reinterpret_cast<const f32x4*>(a.slab_bias + col) : v;
        for (int y0 = 0; y0 < a.n_slabs; y0 += 8) {
            f32x4 tt[8];
#pragma unroll
            for (int y = 0; y < 8; ++y) tt[y] = *reinterpret_cast<const f32x4*>(p + (long)min(y0 + y, a.n_slabs - 1) * a.slab_stride);
#pragma unroll
            for (int y = 0; y < 8; ++y) if (y0 + y < a.n_slabs) v += tt[y];
        }
        if (a.slab_bias) v += bias;
        return v;
    };
    f32x4 qf[NC];
    {
        const float* qp = a.Q + (long)b * a.q_bstride + (long)qi * a.ldq + h * HD;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            if (SLABS) qf[c] = qvalid ? from_slabs(qi, h * HD + 4 * (g + 4 * c)) : z;
            else qf[c] = qvalid ? *reinterpret_cast<const f32x4*>(qp + 4 * (g + 4 * c)) : z;
        }
        if (a.l2norm) {
            float ss = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) ss += qf[c][e] * qf[c][e];
            ss += lane_xor<16>(ss);
            ss += lane_xor<32>(ss);
            const float den = fmaxf(sqrtf(ss), 1e-12f);
            const float mul = a.qscale[h];
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) qf[c][e] = (qf[c][e] / den) * mul;
        }
        if (a.scale != 1.0f) {
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) qf[c][e] *= a.scale;
        }
    }
    const float* Kb = a.K + (long)b * a.k_bstride + h * HD;
    const float* Vb = a.V + (long)b * a.v_bstride + h * HD;
    const int ntiles = (a.Lk + 15) >> 4;

    // fragment loads of key tile t: K rows t*16 + r (chunks g + 4c), V rows t*16 + 4g + j (columns 4r..4r+3); rows >= Lk read row Lk-1
    f32x4 kf[2][NC], vf[2][4];
    auto load_tile = [&](int t, int slot) {
        const int kr = min(t * 16 + r, a.Lk - 1);
        float* kp = const_cast<float*>(Kb) + (long)kr * a.ldk;
        if (SLABS && kr >= new0) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                kf[slot][c] = from_slabs(kr - new0, E + h * HD + 4 * (g + 4 * c));
                if (t * 16 + r < a.Lk) *reinterpret_cast<f32x4*>(kp + 4 * (g + 4 * c)) = kf[slot][c];
            }
        } else {
#pragma unroll
            for (int c = 0; c < NC; ++c) kf[slot][c] = *reinterpret_cast<const f32x4*>(kp + 4 * (g + 4 * c));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int vr = min(t * 16 + 4 * g + j, a.Lk - 1);
            float* vp = const_cast<float*>(Vb) + (long)vr * a.ldv + NDT * r;
            if (SLABS && vr >= new0) {
                vf[slot][j] = from_slabs(vr - new0, 2 * E + h * HD + NDT * r);
                if (t * 16 + 4 * g + j < a.Lk) *reinterpret_cast<f32x4*>(vp) = vf[slot][j];
            } else {
                vf[slot][j] = *reinterpret_cast<const f32x4*>(vp);
            }
        }
    };

    float m_run = -INFINITY, l_part = 0.f;
    f32x4 ot[NDT];
#pragma unroll
    for (int d = 0; d < NDT; ++d) { f32x4 z = {0.f, 0.f, 0.f, 0.f}; ot[d] = z; }

    auto do_tile = [&](int t, int slot) {
        if (a.l2norm) {     // key row norm: this lane holds 16 of the 64 elements, the rest sit in the lanes g' != g of the same r
            float ss = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) ss += kf[slot][c][e] * kf[slot][c][e];
            ss += lane_xor<16>(ss);
            ss += lane_xor<32>(ss);
            const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);     // one division per key row; x * (1/n) instead of x / n: 1 ulp
#pragma unroll
            for (int c = 0; c < NC; ++c) kf[slot][c] *= inv;
        }
        f32x4 st = {0.f, 0.f, 0.f, 0.f};          // st[j] = S[key t*16 + 4g + j][query qi]
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) st = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[slot][c][e], qf[c][e], st, 0, 0, 0);
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float sv = (t * 16 + 4 * g + j < a.Lk) ? st[j] : -INFINITY;
            st[j] = sv;
            mx = fmaxf(mx, sv);
        }
        mx = fmaxf(mx, lane_xor<16>(mx));
        mx = fmaxf(mx, lane_xor<32>(mx));
        const float m_new = fmaxf(m_run, mx);
        const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = expf(m_run - m_safe);
        m_run = m_new;
        float ps = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float p = expf(st[j] - m_safe); st[j] = p; ps += p; }
        l_part = l_part * alpha + ps;
#pragma unroll
        for (int d = 0; d < NDT; ++d) ot[d] *= alpha;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int d = 0; d < NDT; ++d) ot[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[slot][j][d], st[j], ot[d], 0, 0, 0);
    };

    int t = wave;
    if (t < ntiles) load_tile(t, 0);
    while (t < ntiles) {            // unrolled by two so that the fragment slots are compile-time register sets (a third slot - loads two tiles
        if (t + 4 < ntiles) load_tile(t + 4, 1);      // ahead - costs the registers that keep all 768 workgroups of the 50-query step resident: 19.8 -> 23.7 us)
        do_tile(t, 0);
        t += 4;
        if (t >= ntiles) break;
        if (t + 4 < ntiles) load_tile(t + 4, 0);
        do_tile(t, 1);
        t += 4;
    }
    float l = l_part;
    l += lane_xor<16>(l);
    l += lane_xor<32>(l);
    // ---- merge the four waves' partial results: ot[dd][reg] = O[query r][d = NDT*(4g+reg) + dd] ----
    if (g == 0) { Ms[wave][r] = m_run; Ls[wave][r] = l; }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const f32x4 o = {ot[0][reg], ot[1][reg], ot[2][reg], ot[3][reg]};
        *reinterpret_cast<f32x4*>(&Os[wave][r][NDT * (4 * g + reg)]) = o;
    }
    __syncthreads();
    {
        const int q = tid & 15, dc = tid >> 4;        // query, 4-float chunk of d
        float M = -INFINITY;
#pragma unroll
        for (int w = 0; w < 4; ++w) M = fmaxf(M, Ms[w][q]);
        float L = 0.f;
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float mw = Ms[w][q];
            const float sc = (mw == -INFINITY) ? 0.f : expf(mw - M);
            L += Ls[w][q] * sc;
            const f32x4 ow = *reinterpret_cast<const f32x4*>(&Os[w][q][4 * dc]);
            o += ow * sc;
        }
        if (q0 + q < a.Lq) {
            const float inv = 1.0f / L;
            float* op = a.O + (long)b * a.o_bstride + (long)(q0 + q) * a.ldo + h * HD;
            if (a.out_p8) store_p8x4(op, 4 * dc, o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv, a.status, a.o_exp);
            else { const f32x4 v = {o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv}; *reinterpret_cast<f32x4*>(op + 4 * dc) = v; }
        }
    }
}

// ---- f16x3 mode: the same attention on the fp16 matrix cores with every operand split hi + lo (the arithmetic of gemm_f16s.hip,
// unscaled: |q|, |k|, |v| are O(1..10), p <= 1).  v_mfma_f32_16x16x32_f16, three products per fp32 product:
//   S^T tile (16 keys x 16 queries) = 2 k-blocks x {Kh Qh, Kl Qh, Kh Ql}                       6 MFMAs of 16 cycles (fp32: 16 of 32)
//   O^T (16 d x 16 queries) += V^T P^T over 32 keys = {Vh Ph, Vl Ph, Vh Pl}                     3 MFMAs per d tile
// K and V are staged into LDS already split (row-major fp16 hi / lo images, 160-byte rows: conflict-free for both read kinds).
// The V^T operand needs 8 consecutive KEYS per lane at one d: read by ds_read_b64_tr_b16 (4 keys x 16 d block per 16 lanes,
// delivered column-major), two reads per operand.  The k slots of the PV MFMA are assigned so that the S^T accumulator is the
// P^T operand without any data movement: slot (g, e) = key 32T + 4g + e for e < 4 (tile 2T), key 32T + 16 + 4g + (e - 4) (tile 2T+1).
#ifndef ATT_ABL
#define ATT_ABL 0
#endif
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x4_raw __attribute__((__vector_size__(4 * sizeof(__fp16))));
__device__ __forceinline__ void split4(const f32x4 x, f16x4_t& hi, f16x4_t& lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { const _Float16 a = (_Float16)x[e]; hi[e] = a; lo[e] = (_Float16)(x[e] - (float)a); }
}
// QKVP8: Q, K and V arrive already in the P8 split format (x16, hi | lo per 8 elements: common.h), written by the qkv GEMM's
// epilogue: the fragments and the LDS images are then plain copies (no conversion: the staging VALU work was ~40 % of this kernel's);
// the x16 of q and k is taken out of the scores together with the softmax scale, the x16 of v out of the final normalisation.
template <int FASTEXP, int QKVP8 = 0>
__global__ __launch_bounds__(256, 4) void attention_f16_kernel(const AttnArgs a) {
    constexpr int HD = 64, KB = 64, PB = 160;      // keys per block, bytes per LDS row (64 halves + pad)
    __shared__ __attribute__((aligned(16))) unsigned char Kh[KB * PB], Kl[KB * PB], Vh[KB * PB], Vl[KB * PB];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 15, g = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y;
    const int qblk0 = blockIdx.x * 64;
    const int q0 = qblk0 + wave * 16;
    const int qi = q0 + r;
    const bool qvalid = qi < a.Lq;
    const bool wave_active = q0 < a.Lq;

    // ---- Q fragments (B operand of S^T): lane (r, g) holds Q[qi][8g + 32kb .. +7], split ----
    h8_t qh[2], ql[2];
    if constexpr (QKVP8) {
        const unsigned char* qp = reinterpret_cast<const unsigned char*>(a.Q + (long)b * a.q_bstride + (long)qi * a.ldq + h * HD);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const h8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
            qh[kb] = qvalid ? *reinterpret_cast<const h8_t*>(qp + (g + 4 * kb) * 32) : z;          // 8-element group g + 4 kb: 16 B hi, 16 B lo
            ql[kb] = qvalid ? *reinterpret_cast<const h8_t*>(qp + (g + 4 * kb) * 32 + 16) : z;
        }
    } else {
        const float* qp = a.Q + (long)b * a.q_bstride + (long)qi * a.ldq + h * HD;
        f32x4 x[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                f32x4 z = {0.f, 0.f, 0.f, 0.f};
                x[kb][u] = qvalid ? *reinterpret_cast<const f32x4*>(qp + 8 * g + 32 * kb + 4 * u) : z;
            }
        if (a.l2norm) {
            float ss = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) ss += x[kb][u][e] * x[kb][u][e];
            ss += lane_xor<16>(ss);
            ss += lane_xor<32>(ss);
            const float den = fmaxf(sqrtf(ss), 1e-12f);
            const float mul = a.qscale[h];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[kb][u][e] = (x[kb][u][e] / den) * mul;
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f16x4_t h0, l0, h1, l1;
            split4(x[kb][0] * a.scale, h0, l0);
            split4(x[kb][1] * a.scale, h1, l1);
            qh[kb] = h8_t{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
            ql[kb] = h8_t{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
        }
    }
    const float sfix = QKVP8 ? a.scale * p8_scale_of(-2 * a.qkv_exp) : 1.0f;      // scores of P8 operands carry the square of their site scale (16 * 16 by default) and no softmax scale yet
    const int klim = (a.split_q > 0 && qi < a.split_q) ? a.split_k : a.Lk;
    const int qblk_last = min(qblk0 + 63, a.Lq - 1);
    const int lk_wg = (a.split_q > 0 && qblk_last < a.split_q) ? a.split_k : a.Lk;

    float m_run = -INFINITY, l_part = 0.f;
    f32x4 ot[4];       // ot[dt][reg] = O[query r][d = 16 dt + 4 g + reg] (unnormalised)
#pragma unroll
    for (int d = 0; d < 4; ++d) { f32x4 z = {0.f, 0.f, 0.f, 0.f}; ot[d] = z; }

    const float* Kb = a.K + (long)b * a.k_bstride + h * HD;
    const float* Vb = a.V + (long)b * a.v_bstride + h * HD;
    // transposed-read lane role inside its group of 16: lane 4q + p supplies row q, columns 4p .. 4p+3
    const int trq = r >> 2, trp = r & 3;

    for (int kb0 = 0; kb0 < lk_wg; kb0 += KB) {
        __syncthreads();
        if constexpr (QKVP8) {
#pragma unroll
            for (int i = 0; i < KB * 8 / 256; ++i) {      // one 8-element group (16 B hi + 16 B lo) of K and of V per thread and pass
                const int idx = tid + i * 256;
                const int row = idx >> 3, g8 = idx & 7;
                const int kr = kb0 + row;
                h8_t kh = {0, 0, 0, 0, 0, 0, 0, 0}, kl = kh, vh = kh, vl = kh;
                if (kr < a.Lk) {
                    const unsigned char* kp = reinterpret_cast<const unsigned char*>(Kb + (long)kr * a.ldk) + g8 * 32;
                    const unsigned char* vp = reinterpret_cast<const unsigned char*>(Vb + (long)kr * a.ldv) + g8 * 32;
                    kh = *reinterpret_cast<const h8_t*>(kp); kl = *reinterpret_cast<const h8_t*>(kp + 16);
                    vh = *reinterpret_cast<const h8_t*>(vp); vl = *reinterpret_cast<const h8_t*>(vp + 16);
                }
                *reinterpret_cast<h8_t*>(Kh + row * PB + g8 * 16) = kh;
                *reinterpret_cast<h8_t*>(Kl + row * PB + g8 * 16) = kl;
                *reinterpret_cast<h8_t*>(Vh + row * PB + g8 * 16) = vh;
                *reinterpret_cast<h8_t*>(Vl + row * PB + g8 * 16) = vl;
            }
        } else {
#pragma unroll
        for (int i = 0; i < KB * 16 / 256; ++i) {
            const int idx = tid + i * 256;
            const int row = idx >> 4, c4 = (idx & 15) * 4;
            const int kr = kb0 + row;
            f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
            if (kr < a.Lk) {
                kv = *reinterpret_cast<const f32x4*>(Kb + (long)kr * a.ldk + c4);
                vv = *reinterpret_cast<const f32x4*>(Vb + (long)kr * a.ldv + c4);
            }
            if (a.l2norm) {
                float ss = kv[0] * kv[0] + kv[1] * kv[1] + kv[2] * kv[2] + kv[3] * kv[3];
                ss = group_sum<16>(ss);
                const float den = fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
                for (int e = 0; e < 4; ++e) kv[e] = kv[e] / den;
            }
            f16x4_t hh, ll;
            split4(kv, hh, ll);
            *reinterpret_cast<f16x4_t*>(Kh + row * PB + c4 * 2) = hh;
            *reinterpret_cast<f16x4_t*>(Kl + row * PB + c4 * 2) = ll;
            split4(vv, hh, ll);
            *reinterpret_cast<f16x4_t*>(Vh + row * PB + c4 * 2) = hh;
            *reinterpret_cast<f16x4_t*>(Vl + row * PB + c4 * 2) = ll;
        }
        }
        __syncthreads();
        if (!wave_active) continue;      // wave-uniform: EXEC stays all ones for the transposed reads below

        const int ntile = min(4, (a.Lk - kb0 + 15) >> 4);
        f32x4 st[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (t < ntile) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    const int off = (t * 16 + r) * PB + (8 * g + 32 * kb) * 2;
                    const h8_t kh = *reinterpret_cast<const h8_t*>(Kh + off), kl = *reinterpret_cast<const h8_t*>(Kl + off);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, qh[kb], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl, qh[kb], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, ql[kb], acc, 0, 0, 0);
                }
            }
            st[t] = acc;
        }
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kidx = kb0 + t * 16 + 4 * g + j;
                const float sv = (kidx < klim) ? (QKVP8 ? st[t][j] * sfix : st[t][j]) : -INFINITY;
                st[t][j] = sv;
                mx = fmaxf(mx, sv);
            }
        mx = fmaxf(mx, lane_xor<16>(mx));
        mx = fmaxf(mx, lane_xor<32>(mx));
        const float m_new = fmaxf(m_run, mx);
        const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = FASTEXP ? __expf(m_run - m_safe) : expf(m_run - m_safe);
        m_run = m_new;
        float ps = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float p = FASTEXP ? __expf(st[t][j] - m_safe) : expf(st[t][j] - m_safe);
                st[t][j] = p;
                ps += p;
            }
        l_part = l_part * alpha + ps;
#pragma unroll
        for (int d = 0; d < 4; ++d) ot[d] *= alpha;
#pragma unroll
        for (int T = 0; T < 2; ++T) {
            if (2 * T < ntile) {        // tile 2T+1 may be past Lk: its p are exp(-inf) = 0 and its V rows were staged as zeros
                f16x4_t h0, l0, h1, l1;
                split4(st[2 * T], h0, l0);
                split4(st[2 * T + 1], h1, l1);
                const h8_t ph = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                const h8_t pl = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
                const int row1 = T * 32 + 4 * g + trq, row2 = row1 + 16;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const int col = (16 * dt + 4 * trp) * 2;
                    const f16x4_t a1 = __builtin_bit_cast(f16x4_t, __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                        (__attribute__((address_space(3))) fp16x4_raw*)(Vh + row1 * PB + col)));
                    const f16x4_t a2 = __builtin_bit_cast(f16x4_t, __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                        (__attribute__((address_space(3))) fp16x4_raw*)(Vh + row2 * PB + col)));
                    const f16x4_t b1 = __builtin_bit_cast(f16x4_t, __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                        (__attribute__((address_space(3))) fp16x4_raw*)(Vl + row1 * PB + col)));
                    const f16x4_t b2 = __builtin_bit_cast(f16x4_t, __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                        (__attribute__((address_space(3))) fp16x4_raw*)(Vl + row2 * PB + col)));
                    const h8_t vh = {a1[0], a1[1], a1[2], a1[3], a2[0], a2[1], a2[2], a2[3]};
                    const h8_t vl = {b1[0], b1[1], b1[2], b1[3], b2[0], b2[1], b2[2], b2[3]};
                    ot[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, ph, ot[dt], 0, 0, 0);
                    ot[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl, ph, ot[dt], 0, 0, 0);
                    ot[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, pl, ot[dt], 0, 0, 0);
                }
            }
        }
    }
    if (!wave_active) return;
    float l = l_part;
    l += lane_xor<16>(l);
    l += lane_xor<32>(l);
    if (qvalid) {
        const float inv = QKVP8 ? 1.0f / (l * p8_scale_of(a.qkv_exp)) : 1.0f / l;      // P8 values carry their site scale (x16 by default)
        float* op = a.O + (long)b * a.o_bstride + (long)qi * a.ldo + h * HD;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const int d0 = 16 * dt + 4 * g;
            const float o0 = ot[dt][0] * inv, o1 = ot[dt][1] * inv, o2 = ot[dt][2] * inv, o3 = ot[dt][3] * inv;
            if (a.out_p8) store_p8x4(op, d0, o0, o1, o2, o3, a.status, a.o_exp);
            else { const f32x4 o = {o0, o1, o2, o3}; *reinterpret_cast<f32x4*>(op + d0) = o; }
        }
    }
}

// ---- One workgroup per (clip, head) for sequences of up to 208 queries x 256 keys with P8 operands (the wav2vec2 encoder: 199 x 199,
// the VAE stacks): attention_f16_kernel stages every 64-key block of K and V once per 64-query workgroup - four workgroups per head
// each pay four exposed global -> LDS round trips, and a launch is those latencies, not its 3 us of matrix work per workgroup.  Here
// ALL keys of the head are staged once (K and V, hi and lo images: 4 x 224 rows x 160 B = 140 KB, every load of the workgroup in
// flight before the first wait), one barrier, then up to 13 waves of 16 queries run attention_f16_kernel's block loop out of LDS with
// no further synchronisation.  Same blocks, same order, same arithmetic per query: results are bit-identical to that kernel's.
// wav2vec2 layer 120 -> 111 us, VAE decoder 15.4 -> 14.1 us (the launch is now one staging latency + ~10 us of matrix / softmax work
// per head with one workgroup per CU; start staggers between the waves of a SIMD changed nothing), 0.7 ms per step in the model.
constexpr int kWideMaxWaves = 13, kWideMaxKeys = 256;
template <int FASTEXP>
__global__ __launch_bounds__(kWideMaxWaves * 64) void attention_f16_wide_kernel(const AttnArgs a) {
    constexpr int HD = 64, KB = 64, PB = 160, NT = kWideMaxWaves * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char wide_smem[];
    const int lkp = (a.Lk + 31) & ~31;             // rows staged: the P V step walks pairs of 16-key tiles
    unsigned char* const Kh = wide_smem;
    unsigned char* const Kl = Kh + lkp * PB;
    unsigned char* const Vh = Kl + lkp * PB;
    unsigned char* const Vl = Vh + lkp * PB;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 15, g = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q0 = wave * 16;
    const int qi = q0 + r;
    const bool qvalid = qi < a.Lq;
    const bool wave_active = q0 < a.Lq;

    const float* Kb = a.K + (long)b * a.k_bstride + h * HD;
    const float* Vb = a.V + (long)b * a.v_bstride + h * HD;
    // ---- stage: one 8-element group (16 B hi + 16 B lo) of K and of V per thread and pass, all passes' loads issued first ----
    constexpr int NPASS = (kWideMaxKeys * 8 + NT - 1) / NT;      // 3
    {
        h8_t kh[NPASS], kl[NPASS], vh[NPASS], vl[NPASS];
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const int idx = tid + i * NT;
            const int row = idx >> 3, g8 = idx & 7;
            const h8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
            kh[i] = z; kl[i] = z; vh[i] = z; vl[i] = z;
#if ATT_ABL != 1      // (timing builds, results wrong: ATT_ABL 1 no K / V loads, 2 no softmax arithmetic, 3 one MFMA product of three, 4 no P split)
            if (row < a.Lk) {
                const unsigned char* kp = reinterpret_cast<const unsigned char*>(Kb + (long)row * a.ldk) + g8 * 32;
                const unsigned char* vp = reinterpret_cast<const unsigned char*>(Vb + (long)row * a.ldv) + g8 * 32;
                kh[i] = *reinterpret_cast<const h8_t*>(kp); kl[i] = *reinterpret_cast<const h8_t*>(kp + 16);
                vh[i] = *reinterpret_cast<const h8_t*>(vp); vl[i] = *reinterpret_cast<const h8_t*>(vp + 16);
            }
#endif
        }
        // Q fragments (B operand of S^T): lane (r, g) holds Q[qi][8g + 32kb .. +7], hi | lo as stored
        h8_t qh_[2], ql_[2];
        {
            const unsigned char* qp = reinterpret_cast<const unsigned char*>(a.Q + (long)b * a.q_bstride + (long)min(qi, a.Lq - 1) * a.ldq + h * HD);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                qh_[kb] = *reinterpret_cast<const h8_t*>(qp + (g + 4 * kb) * 32);
                ql_[kb] = *reinterpret_cast<const h8_t*>(qp + (g + 4 * kb) * 32 + 16);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const int idx = tid + i * NT;
            const int row = idx >> 3, g8 = idx & 7;
            if (row < lkp) {
                *reinterpret_cast<h8_t*>(Kh + row * PB + g8 * 16) = kh[i];
                *reinterpret_cast<h8_t*>(Kl + row * PB + g8 * 16) = kl[i];
                *reinterpret_cast<h8_t*>(Vh + row * PB + g8 * 16) = vh[i];
                *reinterpret_cast<h8_t*>(Vl + row * PB + g8 * 16) = vl[i];
            }
        }
        __syncthreads();
        if (!wave_active) return;
        const h8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
        const h8_t qh[2] = {qvalid ? qh_[0] : z, qvalid ? qh_[1] : z}, ql[2] = {qvalid ? ql_[0] : z, qvalid ? ql_[1] : z};
        const float sfix = a.scale * p8_scale_of(-2 * a.qkv_exp);      // scores of P8 operands carry the square of their site scale (16 * 16 by default) and no softmax scale yet
        const int klim = (a.split_q > 0 && qi < a.split_q) ? a.split_k : a.Lk;
        // attention_f16_kernel decides per 64-query workgroup how far the key loop runs; the same bound here keeps the block sequence
        // (and with it every rounding) of a query identical to that kernel's
        const int qblk_last = min((q0 & ~63) + 63, a.Lq - 1);
        const int lk_wg = (a.split_q > 0 && qblk_last < a.split_q) ? a.split_k : a.Lk;

        float m_run = -INFINITY, l_part = 0.f;
        f32x4 ot[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) { f32x4 zz = {0.f, 0.f, 0.f, 0.f}; ot[d] = zz; }
        const int trq = r >> 2, trp = r & 3;

        for (int kb0 = 0; kb0 < lk_wg; kb0 += KB) {
            const int ntile = min(4, (a.Lk - kb0 + 15) >> 4);
            f32x4 st[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                if (t < ntile) {
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) {
                        const int off = (kb0 + t * 16 + r) * PB + (8 * g + 32 * kb) * 2;
                        const h8_t kh2 = *reinterpret_cast<const h8_t*>(Kh + off), kl2 = *reinterpret_cast<const h8_t*>(Kl + off);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh2, qh[kb], acc, 0, 0, 0);
#if ATT_ABL != 3
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl2, qh[kb], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh2, ql[kb], acc, 0, 0, 0);
#else
                        acc[0] += (float)kl2[0] * (float)ql[kb][0];
#endif
                    }
                }
                st[t] = acc;
            }
            float mx = -INFINITY;
            // the key mask only where a block can hold masked keys: the last block of the sequence, or any block under a split
            // attention mask (the test is uniform over the workgroup; the values are the same either way)
            if (kb0 + KB <= a.Lk && a.split_q <= 0) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float sv = st[t][j] * sfix;
                        st[t][j] = sv;
                        mx = fmaxf(mx, sv);
                    }
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int kidx = kb0 + t * 16 + 4 * g + j;
                        const float sv = (kidx < klim) ? st[t][j] * sfix : -INFINITY;
                        st[t][j] = sv;
                        mx = fmaxf(mx, sv);
                    }
            }
#if ATT_ABL == 2
            l_part += mx;
#else
            mx = fmaxf(mx, lane_xor<16>(mx));
            mx = fmaxf(mx, lane_xor<32>(mx));
            const float m_new = fmaxf(m_run, mx);
            const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = FASTEXP ? __expf(m_run - m_safe) : expf(m_run - m_safe);
            m_run = m_new;
            float ps = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float p = FASTEXP ? __expf(st[t][j] - m_safe) : expf(st[t][j] - m_safe);
                    st[t][j] = p;
                    ps += p;
                }
            l_part = l_part * alpha + ps;
#pragma unroll
            for (int d = 0; d < 4; ++d) ot[d] *= alpha;
#endif
#pragma unroll
            for (int T = 0; T < 2; ++T) {
                if (2 * T < ntile) {
                    f16x4_t h0, l0, h1, l1;
#if ATT_ABL == 4
                    for (int e = 0; e < 4; ++e) { h0[e] = (_Float16)st[2 * T][e]; h1[e] = (_Float16)st[2 * T + 1][e]; l0[e] = h0[e]; l1[e] = h1[e]; }
#else
                    split4(st[2 * T], h0, l0);
                    split4(st[2 * T + 1], h1, l1);
#endif
                    const h8_t ph = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                    const h8_t pl = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
                    const int row1 = kb0 + T * 32 + 4 * g + trq, row2 = row1 + 16;
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        const int col = (16 * dt + 4 * trp) * 2;
                        const f16x4_t a1 = __builtin_bit_cast(f16x4_t, __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                            (__attribute__((address_space(3))) fp16x4_raw*)(Vh + row1 * PB + col)));
                        const f16x4_t a2 = __builtin_bit_cast(f16x4_t, __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                            (__attribute__((address_space(3))) fp16x4_raw*)(Vh + row2 * PB + col)));
                        const f16x4_t b1 = __builtin_bit_cast(f16x4_t, __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                            (__attribute__((address_space(3))) fp16x4_raw*)(Vl + row1 * PB + col)));
                        const f16x4_t b2 = __builtin_bit_cast(f16x4_t, __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                            (__attribute__((address_space(3))) fp16x4_raw*)(Vl + row2 * PB + col)));
                        const h8_t vh2 = {a1[0], a1[1], a1[2], a1[3], a2[0], a2[1], a2[2], a2[3]};
                        const h8_t vl2 = {b1[0], b1[1], b1[2], b1[3], b2[0], b2[1], b2[2], b2[3]};
                        ot[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh2, ph, ot[dt], 0, 0, 0);
#if ATT_ABL != 3
                        ot[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl2, ph, ot[dt], 0, 0, 0);
                        ot[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh2, pl, ot[dt], 0, 0, 0);
#else
                        ot[dt][0] += (float)vl2[0] * (float)pl[0];
#endif
                    }
                }
            }
        }
        float l = l_part;
        l += lane_xor<16>(l);
        l += lane_xor<32>(l);
        if (qvalid) {
            const float inv = 1.0f / (l * p8_scale_of(a.qkv_exp));      // P8 values carry their site scale (x16 by default)
            float* op = a.O + (long)b * a.o_bstride + (long)qi * a.ldo + h * HD;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const int d0 = 16 * dt + 4 * g;
                const float o0 = ot[dt][0] * inv, o1 = ot[dt][1] * inv, o2 = ot[dt][2] * inv, o3 = ot[dt][3] * inv;
                if (a.out_p8) store_p8x4(op, d0, o0, o1, o2, o3, a.status, a.o_exp);
                else { const f32x4 o = {o0, o1, o2, o3}; *reinterpret_cast<f32x4*>(op + d0) = o; }
            }
        }
    }
}

// ---- Persistent ping-pong form of the wide kernel for the wav2vec2 encoder (199 x 199 per head, 1536 heads per launch).  The wide
// kernel's launch is 6 rounds of (stage all keys: ~5 us of exposed global -> LDS traffic, 315 MB per launch in bursts) + (compute:
// ~13 us); profiles/r04_attn_ablation.log: without the K / V loads the launch takes 79 instead of 108 us.  Here ONE workgroup per CU
// walks heads, and the keys of a head live in two LDS buffers - A = keys 0..127, B = keys 128..223 (4 images each, 160-byte rows as
// above: 80 + 60 KB) - filled by LDS-DMA (global_load_lds with per-lane source addresses: lane L of piece n supplies the 16-byte slot
// 64 n + L of the image, i.e. chunk (64 n + L) % 10 of row (64 n + L) / 10; slots 8 and 9 of a row are its padding) while the waves
// compute on the other buffer: B of this head is fetched under the first two 64-key blocks, A of the NEXT head (and its Q fragments)
// under the last two.  No staging registers, no ds_write, two barriers per head.  The block loop is attention_f16_kernel's: same
// blocks, same order, same arithmetic per query - bit-identical results (tests/test_ops_gpu.py).
constexpr int kPPWaves = 13, kPPRowsA = 128, kPPRowsB = 96;
// transposed 64-bit LDS reads of the V operand for one d tile: rows r1 / r1 + 16 of the hi image and of the lo image (LOFF bytes on)
template <int OFF>
__device__ __forceinline__ f16x4_t pp_read_tr(unsigned addr) {
    f16x4_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
template <int LOFF, int COL>
__device__ __forceinline__ void pp_read_v(unsigned va, f16x4_t& a1, f16x4_t& a2, f16x4_t& b1, f16x4_t& b2) {
    a1 = pp_read_tr<COL>(va);
    a2 = pp_read_tr<16 * 160 + COL>(va);
    b1 = pp_read_tr<LOFF + COL>(va);
    b2 = pp_read_tr<LOFF + 16 * 160 + COL>(va);
}
template <int N>
__device__ __forceinline__ void pp_wait_lgkm() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <int FASTEXP>
__global__ __launch_bounds__(kPPWaves * 64) void attention_f16_pp_kernel(const AttnArgs a) {
    constexpr int HD = 64, KB = 64, PB = 160;
    extern __shared__ __attribute__((aligned(16))) unsigned char pp_smem[];
    unsigned char* const bufA = pp_smem;
    unsigned char* const bufB = pp_smem + 4 * kPPRowsA * PB;

    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int r = lane & 15, g = lane >> 4;
    const int n_items = a.B * a.H;
    const int q0 = wave * 16;
    const int qi = q0 + r;
    const bool qvalid = qi < a.Lq;
    const bool wave_active = q0 < a.Lq;
    const int trq = r >> 2, trp = r & 3;
    const float sfix = a.scale * p8_scale_of(-2 * a.qkv_exp);

    // one buffer of one head: 4 images x R rows, piece idx = (image, n) handled by wave idx % 13
    auto issue_buffer = [&](int item, unsigned char* buf, const int R, int key0) {
        const int b = item / a.H, h = item - b * a.H;
        const unsigned char* Kb = reinterpret_cast<const unsigned char*>(a.K + (long)b * a.k_bstride + h * HD);
        const unsigned char* Vb = reinterpret_cast<const unsigned char*>(a.V + (long)b * a.v_bstride + h * HD);
        const int per = R * 10 / 64;
        for (int idx = wave; idx < 4 * per; idx += kPPWaves) {
            const int im = idx / per, n = idx - im * per;
            const int ci = n * 64 + lane;
            const int row = ci / 10, slot = ci - row * 10;
            const int key = min(key0 + row, a.Lk - 1);               // rows past Lk: a copy of the last key (masked in the softmax; finite)
            const int g8 = slot < 8 ? slot : 0;                       // padding slots: any valid 16 bytes
            const unsigned char* src = (im < 2 ? Kb + (long)key * a.ldk * 4 : Vb + (long)key * a.ldv * 4) + g8 * 32 + (im & 1) * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(buf + im * R * PB + n * 1024), 16, 0, 0);
        }
    };
    auto load_q = [&](int item, h8_t (&qh)[2], h8_t (&ql)[2]) {
        const int b = item / a.H, h = item - b * a.H;
        const unsigned char* qp = reinterpret_cast<const unsigned char*>(a.Q + (long)b * a.q_bstride + (long)min(qi, a.Lq - 1) * a.ldq + h * HD);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {      // (raw: nothing consumes these before the end of the current head, so no wait is placed behind the loads)
            qh[kb] = *reinterpret_cast<const h8_t*>(qp + (g + 4 * kb) * 32);
            ql[kb] = *reinterpret_cast<const h8_t*>(qp + (g + 4 * kb) * 32 + 16);
        }
    };
    const h8_t z8 = {0, 0, 0, 0, 0, 0, 0, 0};

    int item = blockIdx.x;
    if (item >= n_items) return;
    h8_t qh[2], ql[2], qnh[2], qnl[2];
    issue_buffer(item, bufA, kPPRowsA, 0);
    load_q(item, qnh, qnl);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) { qh[kb] = qvalid ? qnh[kb] : z8; ql[kb] = qvalid ? qnl[kb] : z8; }      // rows past Lq: zero fragments
    for (; item < n_items; item += gridDim.x) {
        const int next = item + (int)gridDim.x;
        float m_run = -INFINITY, l_part = 0.f;
        f32x4 ot[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) { f32x4 zz = {0.f, 0.f, 0.f, 0.f}; ot[d] = zz; }

        // 64-key blocks [kfrom, kto) out of a buffer whose row 0 is key `key0`
        auto phase = [&](const unsigned char* buf, auto r_tag, int key0, int kfrom, int kto) {
            constexpr int R = decltype(r_tag)::value;
            const unsigned char* const Kh = buf;
            const unsigned char* const Kl = buf + R * PB;
            const unsigned vh_lds = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)(buf + 2 * R * PB);
            for (int kb0 = kfrom; kb0 < kto; kb0 += KB) {
                const int ntile = min(4, (a.Lk - kb0 + 15) >> 4);
                const int rb = kb0 - key0;
                f32x4 st[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    if (t < ntile) {
#pragma unroll
                        for (int kb = 0; kb < 2; ++kb) {
                            const int off = (rb + t * 16 + r) * PB + (8 * g + 32 * kb) * 2;
                            const h8_t kh2 = *reinterpret_cast<const h8_t*>(Kh + off), kl2 = *reinterpret_cast<const h8_t*>(Kl + off);
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh2, qh[kb], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl2, qh[kb], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh2, ql[kb], acc, 0, 0, 0);
                        }
                    }
                    st[t] = acc;
                }
                float mx = -INFINITY;
                if (kb0 + KB <= a.Lk) {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float sv = st[t][j] * sfix;
                            st[t][j] = sv;
                            mx = fmaxf(mx, sv);
                        }
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int kidx = kb0 + t * 16 + 4 * g + j;
                            const float sv = (kidx < a.Lk) ? st[t][j] * sfix : -INFINITY;
                            st[t][j] = sv;
                            mx = fmaxf(mx, sv);
                        }
                }
                mx = fmaxf(mx, lane_xor<16>(mx));
                mx = fmaxf(mx, lane_xor<32>(mx));
                const float m_new = fmaxf(m_run, mx);
                const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
                const float alpha = FASTEXP ? __expf(m_run - m_safe) : expf(m_run - m_safe);
                m_run = m_new;
                float ps = 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float p = FASTEXP ? __expf(st[t][j] - m_safe) : expf(st[t][j] - m_safe);
                        st[t][j] = p;
                        ps += p;
                    }
                l_part = l_part * alpha + ps;
#pragma unroll
                for (int d = 0; d < 4; ++d) ot[d] *= alpha;
#pragma unroll
                for (int T = 0; T < 2; ++T) {
                    if (2 * T < ntile) {
                        f16x4_t h0, l0, h1, l1;
                        split4(st[2 * T], h0, l0);
                        split4(st[2 * T + 1], h1, l1);
                        const h8_t ph = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                        const h8_t pl = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
                        // The 16 transposed V reads of this pair of tiles go out as asm (the builtin makes hipcc wait for the LDS-DMA in
                        // flight - s_waitcnt vmcnt(0) in front of every read - which would serialise the fetch of the other buffer with
                        // this arithmetic); their results return in order and are waited for with counted lgkmcnt, four reads per d tile.
                        const unsigned va = vh_lds + (unsigned)((rb + T * 32 + 4 * g + trq) * PB + 8 * trp);
                        auto pv = [&](int dt, const f16x4_t& x1, const f16x4_t& x2, const f16x4_t& y1, const f16x4_t& y2) {
                            const h8_t vh2 = {x1[0], x1[1], x1[2], x1[3], x2[0], x2[1], x2[2], x2[3]};
                            const h8_t vl2 = {y1[0], y1[1], y1[2], y1[3], y2[0], y2[1], y2[2], y2[3]};
                            ot[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh2, ph, ot[dt], 0, 0, 0);
                            ot[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl2, ph, ot[dt], 0, 0, 0);
                            ot[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh2, pl, ot[dt], 0, 0, 0);
                        };
                        {      // d tiles 0, 1 then 2, 3: eight reads in flight at a time (16 registers; all sixteen at once spilled)
                            f16x4_t a1[2], a2[2], b1[2], b2[2];
                            pp_read_v<R * PB, 0>(va, a1[0], a2[0], b1[0], b2[0]);
                            pp_read_v<R * PB, 32>(va, a1[1], a2[1], b1[1], b2[1]);
                            pp_wait_lgkm<4>();
                            pv(0, a1[0], a2[0], b1[0], b2[0]);
                            pp_wait_lgkm<0>();
                            pv(1, a1[1], a2[1], b1[1], b2[1]);
                        }
                        {
                            f16x4_t a1[2], a2[2], b1[2], b2[2];
                            pp_read_v<R * PB, 64>(va, a1[0], a2[0], b1[0], b2[0]);
                            pp_read_v<R * PB, 96>(va, a1[1], a2[1], b1[1], b2[1]);
                            pp_wait_lgkm<4>();
                            pv(2, a1[0], a2[0], b1[0], b2[0]);
                            pp_wait_lgkm<0>();
                            pv(3, a1[1], a2[1], b1[1], b2[1]);
                        }
                    }
                }
            }
        };

        // The LDS-DMA of a buffer is ordered before the other waves' reads of it by the issuing wave's OWN vmcnt(0) in front of the
        // barrier - spelled out: hipcc happened to place one at the first barrier (from the fence) and none at the second
        // (buffer A of the next head was complete only through a data dependence on later global loads); gfx950 inserts no wait by itself.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                 // A of this head has landed for every wave (vmcnt(0) + barrier); everyone is done with B of the previous head
        issue_buffer(item, bufB, kPPRowsB, kPPRowsA);
        if (wave_active) phase(bufA, std::integral_constant<int, kPPRowsA>{}, 0, 0, min(a.Lk, kPPRowsA));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                 // B has landed; everyone is done with A
        if (next < n_items) {
            issue_buffer(next, bufA, kPPRowsA, 0);
            load_q(next, qnh, qnl);
        }
        if (wave_active) {
            phase(bufB, std::integral_constant<int, kPPRowsB>{}, kPPRowsA, kPPRowsA, a.Lk);
            float l = l_part;
            l += lane_xor<16>(l);
            l += lane_xor<32>(l);
            if (qvalid) {
                const int b = item / a.H, h = item - b * a.H;
                const float inv = 1.0f / (l * p8_scale_of(a.qkv_exp));      // P8 values carry their site scale (x16 by default)
                float* op = a.O + (long)b * a.o_bstride + (long)qi * a.ldo + h * HD;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const int d0 = 16 * dt + 4 * g;
                    const float o0 = ot[dt][0] * inv, o1 = ot[dt][1] * inv, o2 = ot[dt][2] * inv, o3 = ot[dt][3] * inv;
                    if (a.out_p8) store_p8x4(op, d0, o0, o1, o2, o3, a.status, a.o_exp);
                    else { const f32x4 o = {o0, o1, o2, o3}; *reinterpret_cast<f32x4*>(op + d0) = o; }
                }
            }
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) { qh[kb] = qvalid ? qnh[kb] : z8; ql[kb] = qvalid ? qnl[kb] : z8; }
    }
}

// ---- The same idea for the 100-query scale step of the AR decoder (fp32 q | k | v rows of the KV cache, L2-normalised q and k,
// 362 keys): one workgroup of 7 waves per (clip, head) stages 192 keys at a time (K and V, hi and lo images: 120 KB) - two staging
// round trips per head instead of six per 64-query workgroup, and the normalise + split work of a key done once instead of twice.
// Waves walk attention_f16_kernel<.., 0>'s 64-key blocks in the same order with the same arithmetic: bit-identical results.
constexpr int kWideArWaves = 7, kWideArKeys = 192;
// QKVP8 = 1: q, k, v rows in the P8 format (no normalisation; the encoder / VAE form), several workgroups of NW waves per head.
template <int FASTEXP, int QKVP8 = 0, int NW = kWideArWaves, int KPH = kWideArKeys>
__global__ __launch_bounds__(NW * 64) void attention_f16_wide_ar_kernel(const AttnArgs a) {
    constexpr int HD = 64, KB = 64, PB = 160, NT = NW * 64;
    constexpr int NPASS = (KPH * 16 + NT - 1) / NT;      // 7
    extern __shared__ __attribute__((aligned(16))) unsigned char wide_smem[];
    unsigned char* const Kh = wide_smem;
    unsigned char* const Kl = Kh + KPH * PB;
    unsigned char* const Vh = Kl + KPH * PB;
    unsigned char* const Vl = Vh + KPH * PB;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 15, g = lane >> 4;
    // (clip, head, query part) of this workgroup.  With several query parts per head (the VAE stacks: two) the parts of one head are
    // put on ONE XCD - the hardware deals workgroups to XCDs by linear id % 8 - because they stage the same K and V rows: heads go
    // in groups of 8 (one per XCD), the part is the slower index inside a group (as attention_short_kernel does).
    int b = blockIdx.z, h = blockIdx.y, part = blockIdx.x;
    if (gridDim.x > 1 && ((gridDim.y * gridDim.z) & 7) == 0) {
        const int np = gridDim.x, lin = blockIdx.x + np * (blockIdx.y + gridDim.y * blockIdx.z);
        const int grp = lin / (8 * np), rem = lin - grp * 8 * np;
        const int pair = grp * 8 + (rem & 7);
        part = rem >> 3;
        b = pair / (int)gridDim.y;
        h = pair - b * (int)gridDim.y;
    }
    const int q0 = part * (NW * 16) + wave * 16;
    const int qi = q0 + r;
    const bool qvalid = qi < a.Lq;
    const bool wave_active = q0 < a.Lq;

    // ---- Q fragments (B operand of S^T): lane (r, g) holds Q[qi][8g + 32kb .. +7], normalised, scaled, split ----
    h8_t qh[2], ql[2];
    if constexpr (QKVP8) {
        const unsigned char* qp = reinterpret_cast<const unsigned char*>(a.Q + (long)b * a.q_bstride + (long)min(qi, a.Lq - 1) * a.ldq + h * HD);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const h8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
            const h8_t th = *reinterpret_cast<const h8_t*>(qp + (g + 4 * kb) * 32), tl = *reinterpret_cast<const h8_t*>(qp + (g + 4 * kb) * 32 + 16);
            qh[kb] = qvalid ? th : z;
            ql[kb] = qvalid ? tl : z;
        }
    } else {
        const float* qp = a.Q + (long)b * a.q_bstride + (long)min(qi, a.Lq - 1) * a.ldq + h * HD;
        f32x4 x[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                const f32x4 t = *reinterpret_cast<const f32x4*>(qp + 8 * g + 32 * kb + 4 * u);
                x[kb][u] = qvalid ? t : z;
            }
        if (a.l2norm) {
            float ss = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) ss += x[kb][u][e] * x[kb][u][e];
            ss += lane_xor<16>(ss);
            ss += lane_xor<32>(ss);
            const float den = fmaxf(sqrtf(ss), 1e-12f);
            const float mul = a.qscale[h];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[kb][u][e] = (x[kb][u][e] / den) * mul;
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f16x4_t h0, l0, h1, l1;
            split4(x[kb][0] * a.scale, h0, l0);
            split4(x[kb][1] * a.scale, h1, l1);
            qh[kb] = h8_t{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
            ql[kb] = h8_t{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
        }
    }
    const float sfix = QKVP8 ? a.scale * p8_scale_of(-2 * a.qkv_exp) : 1.0f;      // scores of P8 operands carry the square of their site scale (16 * 16 by default) and no softmax scale yet
    const int klim = (a.split_q > 0 && qi < a.split_q) ? a.split_k : a.Lk;
    // the 64-query workgroup of attention_f16_kernel this wave's queries belong to decides how far its key loop runs there: the same
    // bound here keeps the block sequence of a query identical; the phase loop runs to the furthest bound of the workgroup's waves
    const int lk_wave = (a.split_q > 0 && min((q0 & ~63) + 63, a.Lq - 1) < a.split_q) ? a.split_k : a.Lk;
    const int wg_last = min((part + 1) * (NW * 16) - 1, a.Lq - 1);
    const int lk_loop = (a.split_q > 0 && wg_last < a.split_q) ? a.split_k : a.Lk;
    float m_run = -INFINITY, l_part = 0.f;
    f32x4 ot[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) { f32x4 z = {0.f, 0.f, 0.f, 0.f}; ot[d] = z; }
    const float* Kb = a.K + (long)b * a.k_bstride + h * HD;
    const float* Vb = a.V + (long)b * a.v_bstride + h * HD;
    const int trq = r >> 2, trp = r & 3;

    for (int k0 = 0; k0 < lk_loop; k0 += KPH) {
        __syncthreads();                         // every wave is done with the previous phase's rows
        if constexpr (QKVP8) {      // one 8-element group (16 B hi + 16 B lo) of K and of V per thread and pass: plain copies
            constexpr int NP8 = (KPH * 8 + NT - 1) / NT;
            h8_t kh[NP8], kl[NP8], vh[NP8], vl[NP8];
#pragma unroll
            for (int i = 0; i < NP8; ++i) {
                const int idx = tid + i * NT;
                const int row = idx >> 3, g8 = idx & 7;
                const int kr = k0 + row;
                const h8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
                kh[i] = z; kl[i] = z; vh[i] = z; vl[i] = z;
                if (row < KPH && kr < a.Lk) {
                    const unsigned char* kp = reinterpret_cast<const unsigned char*>(Kb + (long)kr * a.ldk) + g8 * 32;
                    const unsigned char* vp = reinterpret_cast<const unsigned char*>(Vb + (long)kr * a.ldv) + g8 * 32;
                    kh[i] = *reinterpret_cast<const h8_t*>(kp); kl[i] = *reinterpret_cast<const h8_t*>(kp + 16);
                    vh[i] = *reinterpret_cast<const h8_t*>(vp); vl[i] = *reinterpret_cast<const h8_t*>(vp + 16);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NP8; ++i) {
                const int idx = tid + i * NT;
                const int row = idx >> 3, g8 = idx & 7;
                if (row < KPH) {
                    *reinterpret_cast<h8_t*>(Kh + row * PB + g8 * 16) = kh[i];
                    *reinterpret_cast<h8_t*>(Kl + row * PB + g8 * 16) = kl[i];
                    *reinterpret_cast<h8_t*>(Vh + row * PB + g8 * 16) = vh[i];
                    *reinterpret_cast<h8_t*>(Vl + row * PB + g8 * 16) = vl[i];
                }
            }
        } else
        {   // stage keys k0 .. k0 + KPH - 1: 16 threads per key row, all loads first
            f32x4 kv[NPASS], vv[NPASS];
#pragma unroll
            for (int i = 0; i < NPASS; ++i) {
                const int idx = tid + i * NT;
                const int row = idx >> 4, c4 = (idx & 15) * 4;
                const int kr = k0 + row;
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                kv[i] = z; vv[i] = z;
                if (row < KPH && kr < a.Lk) {
                    kv[i] = *reinterpret_cast<const f32x4*>(Kb + (long)kr * a.ldk + c4);
                    vv[i] = *reinterpret_cast<const f32x4*>(Vb + (long)kr * a.ldv + c4);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NPASS; ++i) {
                const int idx = tid + i * NT;
                const int row = idx >> 4, c4 = (idx & 15) * 4;
                f32x4 kx = kv[i];
                if (a.l2norm) {
                    float ss = kx[0] * kx[0] + kx[1] * kx[1] + kx[2] * kx[2] + kx[3] * kx[3];
                    ss = group_sum<16>(ss);
                    const float den = fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
                    for (int e = 0; e < 4; ++e) kx[e] = kx[e] / den;
                }
                if (row < KPH) {
                    f16x4_t hh, ll;
                    split4(kx, hh, ll);
                    *reinterpret_cast<f16x4_t*>(Kh + row * PB + c4 * 2) = hh;
                    *reinterpret_cast<f16x4_t*>(Kl + row * PB + c4 * 2) = ll;
                    split4(vv[i], hh, ll);
                    *reinterpret_cast<f16x4_t*>(Vh + row * PB + c4 * 2) = hh;
                    *reinterpret_cast<f16x4_t*>(Vl + row * PB + c4 * 2) = ll;
                }
            }
        }
        __syncthreads();
        if (!wave_active) continue;              // wave-uniform
        const int kend = min(k0 + KPH, lk_wave);
        for (int kb0 = k0; kb0 < kend; kb0 += KB) {
            const int lb = kb0 - k0;             // first LDS row of this block
            const int ntile = min(4, (a.Lk - kb0 + 15) >> 4);
            f32x4 st[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                if (t < ntile) {
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) {
                        const int off = (lb + t * 16 + r) * PB + (8 * g + 32 * kb) * 2;
                        const h8_t kh2 = *reinterpret_cast<const h8_t*>(Kh + off), kl2 = *reinterpret_cast<const h8_t*>(Kl + off);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh2, qh[kb], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl2, qh[kb], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh2, ql[kb], acc, 0, 0, 0);
                    }
                }
                st[t] = acc;
            }
            float mx = -INFINITY;
            if (kb0 + KB <= a.Lk && a.split_q <= 0) {      // a full block without a split mask: no key to hide (uniform test, same values)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float sv = QKVP8 ? st[t][j] * sfix : st[t][j];
                        st[t][j] = sv;
                        mx = fmaxf(mx, sv);
                    }
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int kidx = kb0 + t * 16 + 4 * g + j;
                        const float sv = (kidx < klim) ? (QKVP8 ? st[t][j] * sfix : st[t][j]) : -INFINITY;
                        st[t][j] = sv;
                        mx = fmaxf(mx, sv);
                    }
            }
            mx = fmaxf(mx, lane_xor<16>(mx));
            mx = fmaxf(mx, lane_xor<32>(mx));
            const float m_new = fmaxf(m_run, mx);
            const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = FASTEXP ? __expf(m_run - m_safe) : expf(m_run - m_safe);
            m_run = m_new;
            float ps = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float p = FASTEXP ? __expf(st[t][j] - m_safe) : expf(st[t][j] - m_safe);
                    st[t][j] = p;
                    ps += p;
                }
            l_part = l_part * alpha + ps;
#pragma unroll
            for (int d = 0; d < 4; ++d) ot[d] *= alpha;
#pragma unroll
            for (int T = 0; T < 2; ++T) {
                if (2 * T < ntile) {
                    f16x4_t h0, l0, h1, l1;
                    split4(st[2 * T], h0, l0);
                    split4(st[2 * T + 1], h1, l1);
                    const h8_t ph = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                    const h8_t pl = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
                    const int row1 = lb + T * 32 + 4 * g + trq, row2 = row1 + 16;
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        const int col = (16 * dt + 4 * trp) * 2;
                        const f16x4_t a1 = __builtin_bit_cast(f16x4_t, __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                            (__attribute__((address_space(3))) fp16x4_raw*)(Vh + row1 * PB + col)));
                        const f16x4_t a2 = __builtin_bit_cast(f16x4_t, __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                            (__attribute__((address_space(3))) fp16x4_raw*)(Vh + row2 * PB + col)));
                        const f16x4_t b1 = __builtin_bit_cast(f16x4_t, __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                            (__attribute__((address_space(3))) fp16x4_raw*)(Vl + row1 * PB + col)));
                        const f16x4_t b2 = __builtin_bit_cast(f16x4_t, __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                            (__attribute__((address_space(3))) fp16x4_raw*)(Vl + row2 * PB + col)));
                        const h8_t vh2 = {a1[0], a1[1], a1[2], a1[3], a2[0], a2[1], a2[2], a2[3]};
                        const h8_t vl2 = {b1[0], b1[1], b1[2], b1[3], b2[0], b2[1], b2[2], b2[3]};
                        ot[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh2, ph, ot[dt], 0, 0, 0);
                        ot[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl2, ph, ot[dt], 0, 0, 0);
                        ot[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh2, pl, ot[dt], 0, 0, 0);
                    }
                }
            }
        }
    }
    if (!wave_active) return;
    float l = l_part;
    l += lane_xor<16>(l);
    l += lane_xor<32>(l);
    if (qvalid) {
        const float inv = QKVP8 ? 1.0f / (l * p8_scale_of(a.qkv_exp)) : 1.0f / l;      // P8 values carry their site scale (x16 by default)
        float* op = a.O + (long)b * a.o_bstride + (long)qi * a.ldo + h * HD;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const int d0 = 16 * dt + 4 * g;
            const float o0 = ot[dt][0] * inv, o1 = ot[dt][1] * inv, o2 = ot[dt][2] * inv, o3 = ot[dt][3] * inv;
            if (a.out_p8) store_p8x4(op, d0, o0, o1, o2, o3, a.status, a.o_exp);
            else { const f32x4 o = {o0, o1, o2, o3}; *reinterpret_cast<f32x4*>(op + d0) = o; }
        }
    }
}

void attention_prepare() {      // more than the default 64 KB of dynamic LDS for the wide kernel; called at model creation (outside any capture)
    static bool done[64] = {};      // per device: the attribute belongs to the function on the current device
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || done[dev]) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_f16_wide_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              4 * kWideMaxKeys * 160);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_f16_pp_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              4 * (kPPRowsA + kPPRowsB) * 160);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_f16_wide_ar_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              4 * kWideArKeys * 160);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_f16_wide_ar_kernel<1, 1, 7, 128>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              4 * 128 * 160);
    done[dev] = true;
}
void launch_attention(const AttnArgs& a, hipStream_t s) {
    if (a.B <= 0 || a.Lq <= 0) return;
    if (a.HD == 64 && a.split_q == 0 && a.Lq <= 64 && a.Lk >= 64 && !a.qkv_p8) {      // AR scale steps 0-3: key-split kernel
        const dim3 grid(((a.H * a.B + 7) / 8) * 8 * ((a.Lq + 15) / 16));
        if (a.slabs && a.Lq <= 16) ARTALK_LAUNCH((attention_short_kernel<64, true>), grid, dim3(256), 0, s, a);
        else if (a.slabs) abort();      // (slabs are only handed over for one query tile per head: engine.hip run_chunk_body)
        else ARTALK_LAUNCH((attention_short_kernel<64, false>), grid, dim3(256), 0, s, a);
        return;
    }
    dim3 grid((a.Lq + 63) / 64, a.H, a.B), block(256);
    // the f16 kernels use the v_exp_f32-based exp (1.2e-6 relative at |x| = 20, where p = 2e-9): 154 -> 137 us per wav2vec2 layer; the fp32 kernels keep expf
    static const int wide = getenv("ARTALK_ATTN_WIDE") ? atoi(getenv("ARTALK_ATTN_WIDE")) : 1;      // 0 = 64-query workgroups everywhere (tests: the wide kernels are bit-identical to it)
    if (a.HD == 64 && a.split16 && a.qkv_p8 && !a.l2norm && wide && a.Lq > 128 && a.Lq <= kWideMaxWaves * 16 && a.Lk <= kWideMaxKeys) {      // (100 queries: 7.2 vs 6.4 us, the two-workgroup form wins)
        const size_t lds = (size_t)4 * ((a.Lk + 31) & ~31) * 160;
        attention_prepare();
        // fewer (clip, head) pairs than CUs (the VAE stacks of one clip group: 128): two 7-wave workgroups per head, 128 keys per phase
        // (14.1 -> 11.9 us); the encoder's 1536 pairs run the same either way (109 us: bound by the softmax arithmetic, not by staging)
        // ARTALK_ATTN_PP [1]: the persistent ping-pong kernel where every CU gets several heads (the encoder); 0 = the one-head-per-workgroup kernel
        static const int pp = getenv("ARTALK_ATTN_PP") ? atoi(getenv("ARTALK_ATTN_PP")) : 1;
        static int n_cu_dev[64] = {};      // per device: the CURRENT one (a model may live on any GPU of the node)
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev < 0 || dev >= 64) dev = 0;
        int& n_cu = n_cu_dev[dev];
        if (!n_cu && (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0)) n_cu = 256;
        const int cus = a.cus > 0 ? (a.cus < n_cu ? a.cus : n_cu) : n_cu;
        if (pp && wide != 2 && a.split_q <= 0 && a.Lk > kPPRowsA && a.Lk <= kPPRowsA + kPPRowsB && (long)a.B * a.H >= 2L * cus) {
            ARTALK_LAUNCH((attention_f16_pp_kernel<1>), dim3(cus), dim3(kPPWaves * 64), (size_t)4 * (kPPRowsA + kPPRowsB) * 160, s, a);
            return;
        }
        if (wide == 2 || (long)a.B * a.H < 256) ARTALK_LAUNCH((attention_f16_wide_ar_kernel<1, 1, 7, 128>), dim3((a.Lq + 111) / 112, a.H, a.B), dim3(7 * 64), (size_t)4 * 128 * 160, s, a);
        else ARTALK_LAUNCH((attention_f16_wide_kernel<1>), dim3(1, a.H, a.B), dim3(kWideMaxWaves * 64), lds, s, a);
    } else if (a.HD == 64 && a.split16 && a.qkv_p8 && !a.l2norm)
        ARTALK_LAUNCH((attention_f16_kernel<1, 1>), grid, block, 0, s, a);
    else if (a.HD == 64 && a.split16 && wide && !a.qkv_p8 && a.split_q == 0 && a.Lq > 32 && a.Lq <= kWideArWaves * 16 && a.Lk > 64) {
        attention_prepare();      // the 100-query scale step of the AR decoder: one workgroup per (clip, head), 192 keys per staging phase
        ARTALK_LAUNCH((attention_f16_wide_ar_kernel<1>), dim3(1, a.H, a.B), dim3(kWideArWaves * 64), (size_t)4 * kWideArKeys * 160, s, a);
    } else if (a.HD == 64 && a.split16)
        ARTALK_LAUNCH(attention_f16_kernel<1>, grid, block, 0, s, a);
    else if (a.HD == 64)
        ARTALK_LAUNCH(attention_kernel<64>, grid, block, 0, s, a);
    else if (a.HD == 32)
        ARTALK_LAUNCH(attention_kernel<32>, grid, block, 0, s, a);
    else
        abort();
}

}  // namespace artalk
