"""CPU oracle for the ARTalk audio->motion path.  TEST INFRASTRUCTURE ONLY.

A plain torch-CPU fp32 restatement of what the reference computes on this path, executed the
way the reference executes it (batch 1, no KV cache, every scale step re-runs all tokens of
levels <= current).  It is the checker for the HIP path and the ``cpu_baseline`` leg of
``bench.py``; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline may
import it.  Nothing under ``artalk_amd/`` imports it and the product path never falls back to it.

Parity status: PINNED.  ``oracle/make_golden.py`` imports the reference from ``/root/reference``
in the build container, loads the same deterministic synthetic weights with ``strict=True`` and
writes ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this file against those
vectors (bits exact, FLAME codes to 1e-5).  The wav2vec2 arithmetic is third-party
(``transformers``, reference pin 4.45.1, container 5.15.0) and is pinned by those same vectors.

Every function cites the reference ``file:line`` it restates (paths relative to /root/reference,
``hf:`` = transformers/models/wav2vec2/modeling_wav2vec2.py).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


class ARTalkOracle:
    def __init__(self, cfg, state_dict):
        self.cfg = cfg
        self.w = {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(v)) for k, v in state_dict.items()}
        self.patch_nums = list(cfg.patch_nums)
        w = self.w
        # effective pos-conv weight: weight_norm(dim=2)  (hf:355; torch parametrizations._WeightNorm)
        pc = "audio_encoder.encoder.pos_conv_embed.conv.parametrizations.weight."
        self.pos_conv_w = torch._weight_norm(w[pc + "original1"], w[pc + "original0"], 2)

    # ------------------------------------------------------------------ wav2vec2
    @staticmethod
    def normalize_audio(x):
        # app/modules/wav2vec.py:22-27  (unbiased std, eps added to std)
        mean = x.mean(dim=-1, keepdim=True)
        std = x.std(dim=-1, keepdim=True)
        return (x - mean) / (std + 1e-6)

    def w2v_feature_extractor(self, x):
        # hf:409-419 over hf:275-299 (Wav2Vec2LayerNormConvLayer): conv -> LN over channels -> GELU(erf)
        w, c = self.w, self.cfg.w2v
        h = x[:, None]
        for i, s in enumerate(c["conv_stride"]):
            p = f"audio_encoder.feature_extractor.conv_layers.{i}."
            h = F.conv1d(h, w[p + "conv.weight"], w[p + "conv.bias"], stride=s)
            h = h.transpose(-2, -1)
            h = F.layer_norm(h, (h.shape[-1],), w[p + "layer_norm.weight"], w[p + "layer_norm.bias"], 1e-5)
            h = h.transpose(-2, -1)
            h = F.gelu(h)
        return h

    def w2v_pos_conv(self, h):
        # hf:360-368 (+ SamePad hf:371-379: kernel 128 is even -> drop last frame)
        w, c = self.w, self.cfg.w2v
        k = c["num_conv_pos_embeddings"]
        y = F.conv1d(h.transpose(1, 2), self.pos_conv_w, w["audio_encoder.encoder.pos_conv_embed.conv.bias"],
                     padding=k // 2, groups=c["num_conv_pos_embedding_groups"])
        if k % 2 == 0:
            y = y[:, :, :-1]
        return F.gelu(y).transpose(1, 2)

    def w2v_layer(self, i, h):
        # hf:631-654 Wav2Vec2EncoderLayerStableLayerNorm; attention hf:467-548 (sdpa, scale 1/sqrt(64), no mask)
        w, c = self.w, self.cfg.w2v
        p = f"audio_encoder.encoder.layers.{i}."
        eps, nh = c["layer_norm_eps"], c["num_attention_heads"]
        B, T, C = h.shape
        res = h
        x = F.layer_norm(h, (C,), w[p + "layer_norm.weight"], w[p + "layer_norm.bias"], eps)
        q = F.linear(x, w[p + "attention.q_proj.weight"], w[p + "attention.q_proj.bias"]).view(B, T, nh, -1).transpose(1, 2)
        k = F.linear(x, w[p + "attention.k_proj.weight"], w[p + "attention.k_proj.bias"]).view(B, T, nh, -1).transpose(1, 2)
        v = F.linear(x, w[p + "attention.v_proj.weight"], w[p + "attention.v_proj.bias"]).view(B, T, nh, -1).transpose(1, 2)
        a = F.scaled_dot_product_attention(q, k, v, attn_mask=None, dropout_p=0.0, scale=(C // nh) ** -0.5)
        a = a.transpose(1, 2).reshape(B, T, C).contiguous()
        h = res + F.linear(a, w[p + "attention.out_proj.weight"], w[p + "attention.out_proj.bias"])
        x = F.layer_norm(h, (C,), w[p + "final_layer_norm.weight"], w[p + "final_layer_norm.bias"], eps)
        x = F.gelu(F.linear(x, w[p + "feed_forward.intermediate_dense.weight"], w[p + "feed_forward.intermediate_dense.bias"]))
        x = F.linear(x, w[p + "feed_forward.output_dense.weight"], w[p + "feed_forward.output_dense.bias"])
        return h + x

    def wav2vec(self, chunk):
        # app/modules/wav2vec.py:11-20
        w, c = self.w, self.cfg.w2v
        x = self.normalize_audio(chunk)
        h = self.w2v_feature_extractor(x).transpose(1, 2)
        # feature projection hf:429-434
        h = F.layer_norm(h, (h.shape[-1],), w["audio_encoder.feature_projection.layer_norm.weight"],
                         w["audio_encoder.feature_projection.layer_norm.bias"], c["layer_norm_eps"])
        h = F.linear(h, w["audio_encoder.feature_projection.projection.weight"],
                     w["audio_encoder.feature_projection.projection.bias"])
        # encoder hf:741-802 (stable layer norm variant)
        h = h + self.w2v_pos_conv(h)
        for i in range(c["num_hidden_layers"]):
            h = self.w2v_layer(i, h)
        return F.layer_norm(h, (h.shape[-1],), w["audio_encoder.encoder.layer_norm.weight"],
                            w["audio_encoder.encoder.layer_norm.bias"], c["layer_norm_eps"])

    # ------------------------------------------------------------------ AR block
    def ar_block(self, i, feat, prev_feat, cond, attn_bias):
        # app/transformer.py:30-43 (AdaLNSelfAttn) and :65-79 (ModifiedSelfAttention)
        w, nh = self.w, self.cfg.ar_heads
        p = f"attn_blocks.{i}."
        B, L, C = feat.shape
        ada = F.linear(F.silu(cond), w[p + "ada_lin.1.weight"], w[p + "ada_lin.1.bias"])
        gamma1, gamma2, scale1, scale2, shift1, shift2 = ada.view(B, cond.shape[1], 6, -1).unbind(2)
        x = F.layer_norm(feat, (C,), None, None, 1e-6).mul(scale1.add(1)).add_(shift1)
        kv_in = torch.cat([prev_feat, x], dim=1)
        Lk = kv_in.shape[1]
        q = F.linear(x, w[p + "attn.query.weight"], w[p + "attn.query.bias"]).view(B, L, nh, -1).transpose(1, 2)
        k = F.linear(kv_in, w[p + "attn.key.weight"], None).view(B, Lk, nh, -1).transpose(1, 2)
        v = F.linear(kv_in, w[p + "attn.value.weight"], w[p + "attn.value.bias"]).view(B, Lk, nh, -1).transpose(1, 2)
        scale_mul = w[p + "attn.scale_mul_1H11"].clamp_max(math.log(100)).exp()
        q = F.normalize(q, dim=-1).mul(scale_mul)
        k = F.normalize(k, dim=-1)
        a = F.scaled_dot_product_attention(query=q, key=k, value=v, scale=1, attn_mask=attn_bias, dropout_p=0.0)
        a = a.transpose(1, 2).reshape(B, L, C)
        a = F.linear(a, w[p + "attn.proj.weight"], w[p + "attn.proj.bias"])
        feat = feat + a.mul_(gamma1)
        x = F.layer_norm(feat, (C,), None, None, 1e-6).mul(scale2.add(1)).add_(shift2)
        x = F.linear(x, w[p + "ffn.0.weight"], w[p + "ffn.0.bias"])
        x = F.gelu(x, approximate="tanh")
        x = F.linear(x, w[p + "ffn.2.weight"], w[p + "ffn.2.bias"])
        return feat + x.mul(gamma2)

    def ar_head(self, feat, cond):
        # app/models.py:145-148 (AdaLNBeforeHead) + :103 logits_head
        w = self.w
        B, L, C = feat.shape
        ada = F.linear(F.silu(cond), w["cond_logits_head.ada_lin.1.weight"], w["cond_logits_head.ada_lin.1.bias"])
        scale, shift = ada.view(B, cond.shape[1], 2, -1).unbind(2)
        x = F.layer_norm(feat, (C,), None, None, 1e-6).mul(scale.add(1)).add_(shift)
        return F.linear(x, w["logits_head.weight"], w["logits_head.bias"])

    # ------------------------------------------------------------------ VAE pieces
    def vae_attn(self, p, x, attn_mask):
        # app/modules/bitwise_vae.py:194-215 SimpleSelfAttention (scale = hidden_dim**-0.5, qkv no bias)
        w, nh = self.w, self.cfg.vae_heads
        B, L, C = x.shape
        qkv = F.linear(F.layer_norm(x, (C,), w[p + "norm.weight"], w[p + "norm.bias"], 1e-5), w[p + "to_qkv.weight"], None)
        q, k, v = qkv.view(B, L, 3, nh, C // nh).permute(2, 0, 3, 1, 4).unbind(0)
        o = F.scaled_dot_product_attention(query=q, key=k, value=v, scale=int(C) ** (-0.5), attn_mask=attn_mask, dropout_p=0.0)
        o = o.permute(0, 2, 1, 3).reshape(B, L, C)
        return F.linear(o, w[p + "to_out.weight"], w[p + "to_out.bias"])

    def vae_stack(self, side, x, attn_mask):
        # app/modules/bitwise_vae.py:149-157 / :183-191
        w = self.w
        stack = "encoder_transformer" if side == "encoder" else "decoder_transformer"
        x = F.leaky_relu(F.linear(x, w[f"basic_vae.{side}.inp_mapping.0.weight"], w[f"basic_vae.{side}.inp_mapping.0.bias"]), 0.2)
        for i in range(self.cfg.vae_depth):
            x = x + self.vae_attn(f"basic_vae.{side}.{stack}.{2 * i}.", x, attn_mask)
            p = f"basic_vae.{side}.{stack}.{2 * i + 1}."
            y = F.gelu(F.linear(x, w[p + "0.weight"], w[p + "0.bias"]), approximate="tanh")
            x = x + F.linear(y, w[p + "2.weight"], w[p + "2.bias"])
        out = "code_mapping" if side == "encoder" else "out_mapping"
        return F.linear(x, w[f"basic_vae.{side}.{out}.weight"], w[f"basic_vae.{side}.{out}.bias"])

    def bsq(self, f, flips=()):
        # app/modules/bitwise_vae.py:316-334 (losses are computed and discarded by the caller)
        z = F.normalize(f, dim=-1)
        q_scale = 1.0 / (self.cfg.code_dim ** 0.5)
        sign = torch.where(z > 0, torch.tensor(1.0), torch.tensor(-1.0))
        for row, bit in flips:       # test-fixture derivation only (oracle/make_alt_golden.py): take the OTHER sign at a decision
            sign[:, row, bit] = -sign[:, row, bit]          # whose margin |z| is at rounding level
        zhat = q_scale * sign
        q = z + (zhat - z)
        return q, (q > 0).int(), z

    def ms_bsq(self, f, flips=()):
        # app/modules/bitwise_vae.py:227-242 MultiScaleBSQ.forward; flips: (token of the 181, bit) decisions to invert (see bsq)
        B, T, C = f.shape
        residual = f
        bits, margins = [], []
        off = 0
        for pt in self.patch_nums:
            r = F.interpolate(residual.permute(0, 2, 1), size=(pt), mode="area").permute(0, 2, 1).contiguous() if pt != T else residual
            q, b, z = self.bsq(r, [(t - off, c) for t, c in flips if off <= t < off + pt])
            off += pt
            q = F.interpolate(q.permute(0, 2, 1), size=(T), mode="linear").permute(0, 2, 1).contiguous() if pt != T else q
            residual = residual - q
            bits.append(b)
            margins.append(z.abs())
        return torch.cat(bits, dim=1), torch.cat(margins, dim=1)

    def quant_to_vqidx(self, motion, flips=()):
        # app/modules/bitwise_vae.py:78-93, this_motion=None branch
        w, T = self.w, self.cfg.frames_per_chunk
        enc_in = (motion - w["basic_vae.motion_mean"]) / w["basic_vae.motion_std"]      # :59-61
        enc_out = self.vae_stack("encoder", enc_in + w["basic_vae.enc_pos_embed"][:, :T], w["basic_vae.attn_mask"][:, :, :T, :T])
        return self.ms_bsq(enc_out, flips)

    def bits_to_h(self, bits):
        return (bits.float() * 2 - 1.0) / (self.cfg.code_dim ** 0.5)

    def vqidx_to_feat(self, bits, multi_scale):
        # app/modules/bitwise_vae.py:264-288
        pn = self.patch_nums
        B, T, C = bits.shape[0], pn[-1], self.cfg.code_dim
        h = self.bits_to_h(bits)
        s, e = 0, pn[0]
        if multi_scale:
            hT = h.permute(0, 2, 1).contiguous()
            f_hat = torch.zeros(B, C, T)
            outs = []
            for pidx in range(len(pn) - 1):
                f_hat.add_(F.interpolate(hT[..., s:e], size=(T), mode="linear"))
                s, e = e, e + pn[pidx + 1]
                outs.append(F.interpolate(f_hat, size=(pn[pidx + 1]), mode="area"))
            return torch.cat(outs, dim=-1).permute(0, 2, 1).contiguous()
        f_hat = torch.zeros(B, T, C)
        for pidx in range(len(pn) - 1):
            up = F.interpolate(h[:, s:e].permute(0, 2, 1).contiguous(), size=(T), mode="linear")
            f_hat.add_(up.permute(0, 2, 1).contiguous())
            s, e = e, e + pn[pidx + 1]
        f_hat.add_(h[:, s:])
        return f_hat

    def vqidx_to_ar_vqfeat(self, this_pidx, bits):
        # app/modules/bitwise_vae.py:291-305
        pn = self.patch_nums
        B, T, C = bits.shape[0], pn[-1], self.cfg.code_dim
        f_hat = torch.zeros(B, C, T)
        hT = self.bits_to_h(bits).permute(0, 2, 1).contiguous()
        s, e = 0, pn[0]
        outs = []
        for pidx in range(this_pidx + 1):
            f_hat.add_(F.interpolate(hT[..., s:e], size=(T), mode="linear").contiguous())
            s, e = e, e + pn[pidx + 1]
            outs.append(F.interpolate(f_hat.clone(), size=(pn[pidx + 1]), mode="area").contiguous())
        return torch.cat(outs, dim=-1).permute(0, 2, 1).contiguous()

    def vqidx_to_motion(self, prev_bits, this_bits):
        # app/modules/bitwise_vae.py:105-113
        w, T = self.w, self.cfg.frames_per_chunk
        vq = torch.cat([self.vqidx_to_feat(prev_bits, False), self.vqidx_to_feat(this_bits, False)], dim=1)
        dec = self.vae_stack("decoder", vq + w["basic_vae.dec_pos_embed"], w["basic_vae.attn_mask"])
        self._last_dec_out = dec                                                            # (tap: decoder output before unnorm)
        motion = dec * w["basic_vae.motion_std"] + w["basic_vae.motion_mean"]              # :63-65
        return motion[:, :T], motion[:, T:]

    # ------------------------------------------------------------------ style
    def style_encoder(self, motion):
        # app/modules/style_encoder.py:26-38; PositionalEncoding quirk :58-60 adds pe[:, seq_len] to every token;
        # nn.TransformerEncoderLayer(d_model=128, nhead=4, dim_feedforward=512, gelu, batch_first, post-LN) restated.
        w, c = self.w, self.cfg
        B, L, _ = motion.shape
        x = (motion.clone() - w["style_encoder.motion_mean"]) / w["style_encoder.motion_std"]
        x = F.linear(x, w["style_encoder.encoder.motion_proj.weight"], w["style_encoder.encoder.motion_proj.bias"])
        x = x + w["style_encoder.PE.pe"][:, L, :]
        S, nh = c.style_dim, c.style_heads
        for i in range(c.style_layers):
            p = f"style_encoder.encoder.transformer.layers.{i}."
            qkv = F.linear(x, w[p + "self_attn.in_proj_weight"], w[p + "self_attn.in_proj_bias"])
            q, k, v = qkv.view(B, L, 3, nh, S // nh).permute(2, 0, 3, 1, 4).unbind(0)
            a = F.scaled_dot_product_attention(q, k, v, dropout_p=0.0)
            a = a.permute(0, 2, 1, 3).reshape(B, L, S)
            a = F.linear(a, w[p + "self_attn.out_proj.weight"], w[p + "self_attn.out_proj.bias"])
            x = F.layer_norm(x + a, (S,), w[p + "norm1.weight"], w[p + "norm1.bias"], 1e-5)
            y = F.linear(F.gelu(F.linear(x, w[p + "linear1.weight"], w[p + "linear1.bias"])), w[p + "linear2.weight"], w[p + "linear2.bias"])
            x = F.layer_norm(x + y, (S,), w[p + "norm2.weight"], w[p + "norm2.bias"], 1e-5)
        return x.mean(dim=1)

    # ------------------------------------------------------------------ full path
    @torch.no_grad()
    def inference(self, batch, record=None, force_hist=None, force_bits=None):
        """app/models.py:62-121 (with_gtmotion=False).  ``record`` (dict) collects per-chunk internals.  ``force_hist``
        ({history index: [(token, bit), ...]}) / ``force_bits`` ({chunk index: [(token, bit), ...]}) invert the listed history / AR
        decisions - used only to derive the continuation of a clip after a rounding-level flip (oracle/make_alt_golden.py,
        tests/conftest.py); the default is the reference's arithmetic."""
        force_hist = force_hist or {}
        force_bits = force_bits or {}
        w, cfg, pn = self.w, self.cfg, self.patch_nums
        audio = batch["audio"]
        B = audio.shape[0]
        assert B == 1, "Only support batch size 1 for inference."                     # :65
        seq_length = math.ceil(audio.shape[-1] / 16000 * 25.0)
        if batch.get("style_motion", None) is not None:
            style = self.style_encoder(batch["style_motion"])
            style_cond = F.linear(style, w["style_cond_embed.weight"], w["style_cond_embed.bias"])[:, None]
            style_cond = style_cond * 1.1 - w["null_style_cond"] * 0.1               # :70
        else:
            style_cond = w["null_style_cond"]
        lvl = w["lvl_embed.weight"][w["lvl_idx"]]
        lvl_pos = lvl + w["pos_embed"]
        prev_lvl_pos = lvl.repeat(1, cfg.prev_ratio, 1) + w["prev_pos_embed"]
        NT = sum(pn)
        padded_frames = math.ceil(seq_length / pn[-1]) * pn[-1]
        padded_audio = int(padded_frames / 25.0 * 16000)
        chunk_len = int(pn[-1] / 25.0 * 16000)
        chunks = torch.cat([audio, audio.new_zeros(B, padded_audio - audio.shape[1])], dim=-1).split(chunk_len, dim=-1)
        prev_motion = audio.new_zeros(B, pn[-1], cfg.motion_dim)
        prev_bits, hist_margin = self.quant_to_vqidx(prev_motion, force_hist.get(0, ()))
        prev_vqfeat = self.vqidx_to_feat(prev_bits, True)
        prev_attn_feat = torch.cat([style_cond, F.linear(prev_vqfeat, w["vqfeat_embed.weight"], w["vqfeat_embed.bias"])], dim=1)
        if record is not None:
            record.update(bits=[], hist_bits=[prev_bits.clone()], logit_margin=[], hist_margin=[hist_margin.clone()],
                          w2v=[], style_cond=style_cond.clone())
        # intermediates in the layout of the device taps (include/artalk_hip.h: artalk_set_tap; reference-captured goldens:
        # oracle/make_golden_taps.py): row t of the block / logit tensors comes from the scale step that introduces token t
        taps = record.get("taps") if record is not None else None
        if taps is not None:
            taps.update(blk0_in=[], blk0_out=[], blkL_out=[], prev_in=[], logits=[], dec_out=[])
        out = []
        for chunk in chunks:
            feat_a = self.wav2vec(chunk).permute(0, 2, 1)
            conds = [F.interpolate(feat_a, size=(p), mode="area").permute(0, 2, 1) for p in pn]    # :94
            cond_all = torch.cat(conds, dim=1)
            nxt = style_cond
            if taps is not None:
                taps["prev_in"].append((prev_attn_feat + prev_lvl_pos)[0].clone())
                for k in ("blk0_in", "blk0_out", "blkL_out"):
                    taps[k].append(torch.zeros(NT, lvl_pos.shape[-1]))
                taps["logits"].append(torch.zeros(NT, 2 * cfg.code_dim))
            for pidx in range(len(pn)):
                L = sum(pn[:pidx + 1])
                new = slice(L - pn[pidx], L)
                cond = cond_all[:, :L]
                bias = w["attn_bias_for_masking"][:, :, :L, :L + NT * cfg.prev_ratio]
                x = nxt + lvl_pos[:, :nxt.shape[1]]
                for i in range(cfg.ar_depth):
                    if taps is not None and i == 0:
                        taps["blk0_in"][-1][new] = x[0, new]
                    x = self.ar_block(i, x, prev_attn_feat + prev_lvl_pos, cond, bias)
                    if taps is not None and i == 0:
                        taps["blk0_out"][-1][new] = x[0, new]
                logits = self.ar_head(x, cond)
                if taps is not None:
                    taps["blkL_out"][-1][new] = x[0, new]
                    taps["logits"][-1][new] = logits[0, new]
                pairs = logits.view(B, L, -1, 2)
                bits = pairs.argmax(dim=-1)                                                          # :104
                for (tok, bit) in force_bits.get(len(out), ()):          # (test infrastructure only, see the docstring)
                    if tok < L:
                        bits[0, tok, bit] ^= 1
                if pidx < len(pn) - 1:
                    nxt = self.vqidx_to_ar_vqfeat(pidx, bits)
                    nxt = torch.cat([style_cond, F.linear(nxt, w["vqfeat_embed.weight"], w["vqfeat_embed.bias"])], dim=1)
            _, pred = self.vqidx_to_motion(prev_bits, bits)
            if taps is not None:
                taps["dec_out"].append(self._last_dec_out[0].clone())
            out.append(pred)
            new_prev_bits, hm = self.quant_to_vqidx(pred, force_hist.get(len(out), ()))
            if record is not None:
                record["bits"].append(bits.clone())
                record["logit_margin"].append((pairs[..., 0] - pairs[..., 1]).abs().clone())
                record["hist_bits"].append(new_prev_bits.clone())
                record["hist_margin"].append(hm.clone())
                record["w2v"].append(feat_a.permute(0, 2, 1).clone())
            prev_bits = new_prev_bits
            prev_vqfeat = self.vqidx_to_feat(prev_bits, True)
            this_prev = torch.cat([style_cond, F.linear(prev_vqfeat, w["vqfeat_embed.weight"], w["vqfeat_embed.bias"])], dim=1)
            prev_attn_feat = torch.cat([prev_attn_feat[:, this_prev.shape[1]:], this_prev], dim=1)
        return torch.cat(out, dim=1)[:, :seq_length]


# ---------------------------------------------------------------------- engine-level post-processing
def smooth_motion_savgol(motion):
    """inference.py:89-95 (scipy Savitzky-Golay, default mode='interp')."""
    from scipy.signal import savgol_filter
    m = motion.clone().detach().cpu().numpy()
    s = savgol_filter(m, window_length=5, polyorder=2, axis=0)
    s[..., 100:103] = savgol_filter(m[..., 100:103], window_length=9, polyorder=3, axis=0)
    return torch.tensor(s).type_as(motion)


def engine_inference(oracle, audio, style_motion=None, clip_length=750, fix_pose=False):
    """inference.py:47-57: model.inference -> savgol -> [:clip_length] -> pose/jaw zeroing."""
    pred = oracle.inference({"audio": audio[None], "style_motion": style_motion})[0]
    pred = smooth_motion_savgol(pred)[:clip_length]
    if fix_pose:
        pred[..., 100:103] *= 0.0
    pred[..., 104:] *= 0.0
    return pred
