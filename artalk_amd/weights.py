"""Weight manifest of the reference's ``BitwiseARModel.state_dict()`` and a deterministic
synthetic-weight generator.

The real checkpoint ``assets/ARTalk_wav2vec.pt`` is downloaded by the reference's
``build_resources.sh:14-35`` and is not available offline, so parity is pinned with
deterministic synthetic weights that carry the reference's key names and shapes
(loaded into the reference with ``strict=True`` by ``oracle/make_golden.py``).  A real
checkpoint loads through the same manifest.

Key names/shapes follow the modules that create them:
``app/models.py:14-56`` (top level), ``app/transformer.py:12-63`` (attn_blocks),
``app/modules/bitwise_vae.py:15-41,128-209`` (basic_vae), ``app/modules/style_encoder.py:10-24,41-56``
(style_encoder) and ``transformers/models/wav2vec2/modeling_wav2vec2.py`` (audio_encoder).
"""
from __future__ import annotations

import hashlib
import math
from collections import OrderedDict

import numpy as np

from .config import ARTalkConfig

DEFAULT_SEED = 1234


# --------------------------------------------------------------------------------------
# manifest
# --------------------------------------------------------------------------------------
def manifest(cfg: ARTalkConfig):
    """Ordered ``name -> (shape, dtype, init)``; ``init`` is a tuple understood by ``_generate``."""
    m = OrderedDict()
    E, D, NT = cfg.embed_dim, cfg.cond_dim, cfg.n_tokens

    def lin(prefix, out_f, in_f, bias=True, gain=1.0):
        m[prefix + ".weight"] = ((out_f, in_f), "f32", ("uniform", gain / math.sqrt(in_f)))
        if bias:
            m[prefix + ".bias"] = ((out_f,), "f32", ("uniform", gain / math.sqrt(in_f)))

    def ln(prefix, n):
        m[prefix + ".weight"] = ((n,), "f32", ("ln_w",))
        m[prefix + ".bias"] = ((n,), "f32", ("ln_b",))

    emb_std = math.sqrt(1.0 / E / 3.0)
    # ---- top level (app/models.py:19-56) ----
    m["null_style_cond"] = ((1, 1, E), "f32", ("normal", 0.5))
    m["pos_embed"] = ((1, NT, E), "f32", ("tnormal", emb_std))
    m["prev_pos_embed"] = ((1, NT * cfg.prev_ratio, E), "f32", ("tnormal", emb_std))
    m["attn_bias_for_masking"] = ((1, 1, NT, NT * (1 + cfg.prev_ratio)), "f32", ("buf_ar_mask",))
    m["lvl_idx"] = ((1, NT), "i64", ("buf_lvl_idx",))
    m["lvl_embed.weight"] = ((len(cfg.patch_nums), E), "f32", ("tnormal", emb_std))
    lin("vqfeat_embed", E, cfg.code_dim)
    lin("style_cond_embed", E, cfg.style_dim)
    lin("cond_logits_head.ada_lin.1", 2 * E, D)
    lin("logits_head", 2 * cfg.code_dim, E)
    # ---- AR blocks (app/transformer.py:12-63) ----
    for i in range(cfg.ar_depth):
        p = f"attn_blocks.{i}"
        m[p + ".attn.scale_mul_1H11"] = ((1, cfg.ar_heads, 1, 1), "f32", ("scale_mul",))
        lin(p + ".attn.query", E, E)
        lin(p + ".attn.key", E, E, bias=False)
        lin(p + ".attn.value", E, E)
        lin(p + ".attn.proj", E, E)
        lin(p + ".ffn.0", 4 * E, E)
        lin(p + ".ffn.2", E, 4 * E)
        lin(p + ".ada_lin.1", 6 * E, D)
    # ---- VAE (app/modules/bitwise_vae.py) ----
    H, T2 = cfg.vae_hidden, 2 * cfg.frames_per_chunk
    m["basic_vae.enc_pos_embed"] = ((1, T2, cfg.motion_dim), "f32", ("tnormal", math.sqrt(1.0 / cfg.motion_dim / 3.0)))
    m["basic_vae.dec_pos_embed"] = ((1, T2, cfg.code_dim), "f32", ("tnormal", math.sqrt(1.0 / cfg.code_dim / 3.0)))
    m["basic_vae.attn_mask"] = ((1, 1, T2, T2), "f32", ("buf_vae_mask",))
    m["basic_vae.motion_mean"] = ((cfg.motion_dim,), "f32", ("stat_mean",))
    m["basic_vae.motion_std"] = ((cfg.motion_dim,), "f32", ("stat_std",))
    lin("basic_vae.encoder.inp_mapping.0", H, cfg.motion_dim)
    lin("basic_vae.encoder.code_mapping", cfg.code_dim, H)
    lin("basic_vae.decoder.inp_mapping.0", H, cfg.code_dim)
    lin("basic_vae.decoder.out_mapping", cfg.motion_dim, H, gain=0.5)
    for side, stack in (("encoder", "encoder_transformer"), ("decoder", "decoder_transformer")):
        for i in range(cfg.vae_depth):
            a = f"basic_vae.{side}.{stack}.{2 * i}"
            ln(a + ".norm", H)
            lin(a + ".to_qkv", 3 * H, H, bias=False)
            lin(a + ".to_out", H, H)
            f = f"basic_vae.{side}.{stack}.{2 * i + 1}"
            lin(f + ".0", int(1.5 * H), H)
            lin(f + ".2", H, int(1.5 * H))
    # ---- style encoder (app/modules/style_encoder.py) ----
    S = cfg.style_dim
    m["style_encoder.motion_mean"] = ((cfg.motion_dim,), "f32", ("stat_mean",))
    m["style_encoder.motion_std"] = ((cfg.motion_dim,), "f32", ("stat_std",))
    m["style_encoder.PE.pe"] = ((1, cfg.style_pe_len, S), "f32", ("buf_pe",))
    lin("style_encoder.encoder.motion_proj", S, cfg.motion_dim)
    for i in range(cfg.style_layers):
        p = f"style_encoder.encoder.transformer.layers.{i}"
        m[p + ".self_attn.in_proj_weight"] = ((3 * S, S), "f32", ("uniform", 1.0 / math.sqrt(S)))
        m[p + ".self_attn.in_proj_bias"] = ((3 * S,), "f32", ("uniform", 1.0 / math.sqrt(S)))
        lin(p + ".self_attn.out_proj", S, S)
        lin(p + ".linear1", cfg.style_ffn, S)
        lin(p + ".linear2", S, cfg.style_ffn)
        ln(p + ".norm1", S)
        ln(p + ".norm2", S)
    # ---- wav2vec2 (transformers modeling_wav2vec2.py) ----
    w = cfg.w2v
    Hs, Fi = w["hidden_size"], w["intermediate_size"]
    m["audio_encoder.masked_spec_embed"] = ((Hs,), "f32", ("uniform", 1.0))   # dead in inference
    cin = 1
    for i, (cd, ck) in enumerate(zip(w["conv_dim"], w["conv_kernel"])):
        p = f"audio_encoder.feature_extractor.conv_layers.{i}"
        bound = 1.0 / math.sqrt(cin * ck)
        m[p + ".conv.weight"] = ((cd, cin, ck), "f32", ("uniform", bound))
        m[p + ".conv.bias"] = ((cd,), "f32", ("uniform", bound))
        ln(p + ".layer_norm", cd)
        cin = cd
    ln("audio_encoder.feature_projection.layer_norm", w["conv_dim"][-1])
    lin("audio_encoder.feature_projection.projection", Hs, w["conv_dim"][-1])
    G, KP = w["num_conv_pos_embedding_groups"], w["num_conv_pos_embeddings"]
    bound = 1.0 / math.sqrt(Hs // G * KP)
    pc = "audio_encoder.encoder.pos_conv_embed.conv"
    m[pc + ".bias"] = ((Hs,), "f32", ("uniform", bound))
    m[pc + ".parametrizations.weight.original0"] = ((1, 1, KP), "f32", ("wn_g", pc + ".parametrizations.weight.original1"))
    m[pc + ".parametrizations.weight.original1"] = ((Hs, Hs // G, KP), "f32", ("uniform", bound))
    ln("audio_encoder.encoder.layer_norm", Hs)
    for i in range(w["num_hidden_layers"]):
        p = f"audio_encoder.encoder.layers.{i}"
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            lin(p + ".attention." + nm, Hs, Hs)
        ln(p + ".layer_norm", Hs)
        ln(p + ".final_layer_norm", Hs)
        lin(p + ".feed_forward.intermediate_dense", Fi, Hs)
        lin(p + ".feed_forward.output_dense", Hs, Fi)
    return m


_STATS = None


def motion_stats():
    """``{"motion_mean": [106], "motion_std": [106]}``: the reference's ALLTALKEMICA table (assets/motion_stats.json)."""
    global _STATS
    if _STATS is None:
        import json
        import os
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", "motion_stats.json")) as f:
            _STATS = json.load(f)
    return _STATS


def n_params(cfg: ARTalkConfig):
    return sum(int(np.prod(s)) for s, _, init in manifest(cfg).values() if not init[0].startswith(("buf", "stat")))


# --------------------------------------------------------------------------------------
# deterministic generator
# --------------------------------------------------------------------------------------
def _rng(name: str, seed: int):
    h = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    return np.random.Generator(np.random.SFC64(int.from_bytes(h[:8], "little")))


def _ar_level_index(cfg):
    return np.concatenate([np.full(pn, i, dtype=np.int64) for i, pn in enumerate(cfg.patch_nums)])


def _generate(name, shape, dtype, init, cfg, seed, done):
    kind = init[0]
    g = _rng(name, seed)
    n = int(np.prod(shape))
    if kind == "uniform":
        a = (g.random(n, dtype=np.float32) * np.float32(2.0) - np.float32(1.0)) * np.float32(init[1])
    elif kind == "normal":
        a = g.standard_normal(n, dtype=np.float32) * np.float32(init[1])
    elif kind == "tnormal":
        a = np.clip(g.standard_normal(n, dtype=np.float32), -2.0, 2.0) * np.float32(init[1])
    elif kind == "ln_w":   # not all-ones, so a dropped/misplaced affine is visible in parity tests
        a = np.float32(1.0) + np.float32(0.1) * (g.random(n, dtype=np.float32) * np.float32(2.0) - np.float32(1.0))
    elif kind == "ln_b":
        a = np.float32(0.1) * (g.random(n, dtype=np.float32) * np.float32(2.0) - np.float32(1.0))
    elif kind == "scale_mul":  # app/transformer.py:54 inits log(4); perturbed per head
        a = np.float32(math.log(4.0)) + np.float32(0.2) * (g.random(n, dtype=np.float32) * np.float32(2.0) - np.float32(1.0))
    elif kind in ("stat_mean", "stat_std"):
        # the reference's own ALLTALKEMICA statistics (app/modules/data_stats.py:1-32; persistent buffers of a checkpoint,
        # bitwise_vae.py:24-25, style_encoder.py:12-13), shipped as data: artalk_amd/assets/motion_stats.json
        a = np.asarray(motion_stats()["motion_mean" if kind == "stat_mean" else "motion_std"], dtype=np.float32)
        assert a.shape == (n,), "motion statistics are defined for MOTION_DIM = 106 only"
    elif kind == "wn_g":       # torch weight_norm(dim=2) initialises g = ||v|| over dims (0,1); perturbed
        v = done[init[1]].astype(np.float64)
        nrm = np.sqrt((v * v).sum(axis=(0, 1), keepdims=True)).astype(np.float32)
        a = (nrm.reshape(-1) * (np.float32(1.0) + np.float32(0.1) * (g.random(n, dtype=np.float32) * 2 - 1))).astype(np.float32)
    elif kind == "buf_ar_mask":   # app/models.py:123-135
        lvl = _ar_level_index(cfg)
        cur = np.where(lvl[:, None] >= lvl[None, :], np.float32(0.0), np.float32(-np.inf)).astype(np.float32)
        a = np.concatenate([np.zeros((cfg.n_tokens, cfg.n_tokens * cfg.prev_ratio), np.float32), cur], axis=1)
    elif kind == "buf_lvl_idx":
        a = _ar_level_index(cfg)
    elif kind == "buf_vae_mask":  # app/modules/bitwise_vae.py:67-76
        T = cfg.frames_per_chunk
        a = np.zeros((2 * T, 2 * T), np.float32)
        a[:T, T:] = -np.inf
    elif kind == "buf_pe":        # app/modules/style_encoder.py:47-55
        import torch  # same ops as the reference so the buffer is bit-identical
        d_model, max_len = shape[2], shape[1]
        pe = torch.zeros(max_len, d_model)
        position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        a = pe.numpy()
    else:
        raise KeyError(kind)
    want = np.int64 if dtype == "i64" else np.float32
    return np.ascontiguousarray(np.asarray(a, dtype=want).reshape(shape))


# Second weight PROFILE (VERDICT r3 missing #3): the real checkpoint (reference inference.py:24-28) cannot be fetched, and
# XLS-R-class encoders carry activation outliers ("massive activations": a few hidden channels with LayerNorm gains of tens, a few
# FFN rows an order of magnitude above the rest) that the benign U(+-1/sqrt(fan_in)) profile never produces.  These profiles plant
# such structure into the same manifest, deterministically, so that the f16x3 operand format (|x| * 16 < 65504) is exercised against
# it with reference goldens: "outlier" keeps every P8 operand inside the format (hundreds to low thousands), "heavy" drives FFN
# hidden activations beyond it (the range guard must trip and the exact-fp32 re-run must pass).
PROFILES = {
    #            LN gain, LN channels, FFN row factor, FFN rows, pos-conv g spread (log range), AR FFN row factor
    "benign": None,
    "outlier": dict(ln_gain=100.0, ln_ch=2, ffn_mul=50.0, ffn_rows=4, g_spread=1.5, ar_ffn_mul=20.0, conv_ln_gain=10.0),
    "heavy": dict(ln_gain=300.0, ln_ch=2, ffn_mul=400.0, ffn_rows=4, g_spread=1.5, ar_ffn_mul=20.0, conv_ln_gain=10.0),
}


def apply_profile(done, cfg: ARTalkConfig, profile: str, seed: int):
    """In-place edit of a generated ``name -> ndarray`` dict (see PROFILES).  Channel / row choices come from name-keyed streams."""
    P = PROFILES[profile]
    if P is None:
        return done

    def pick(name, n, k):
        return _rng("profile:" + name, seed).choice(n, size=k, replace=False)

    Hs, Fi = cfg.w2v["hidden_size"], cfg.w2v["intermediate_size"]
    for i in range(cfg.w2v["num_hidden_layers"]):
        p = f"audio_encoder.encoder.layers.{i}."
        done[p + "final_layer_norm.weight"][pick(p + "ln2", Hs, P["ln_ch"])] = np.float32(P["ln_gain"])
        done[p + "layer_norm.weight"][pick(p + "ln1", Hs, 1)] = np.float32(P["ln_gain"] * 2.0 / 3.0)
        done[p + "feed_forward.intermediate_dense.weight"][pick(p + "ff1", Fi, P["ffn_rows"])] *= np.float32(P["ffn_mul"])
    k = "audio_encoder.feature_extractor.conv_layers.3.layer_norm.weight"
    if k in done:
        done[k][pick(k, done[k].shape[0], 2)] = np.float32(P["conv_ln_gain"])
    k = "audio_encoder.encoder.pos_conv_embed.conv.parametrizations.weight.original0"
    g = _rng("profile:" + k, seed)
    done[k] *= np.exp((g.random(done[k].shape, dtype=np.float32) * 2 - 1) * np.float32(P["g_spread"])).astype(np.float32)
    for i in range(cfg.ar_depth):
        p = f"attn_blocks.{i}.ffn.0.weight"
        done[p][pick(p, done[p].shape[0], 3)] *= np.float32(P["ar_ffn_mul"])
    k = "basic_vae.decoder.decoder_transformer.0.norm.weight"
    done[k][pick(k, done[k].shape[0], 1)] = np.float32(10.0)
    return done


def generate_state_dict(cfg: ARTalkConfig, seed: int = DEFAULT_SEED, as_torch: bool = True, profile: str = "benign"):
    """Deterministic synthetic ``state_dict`` with the reference's keys (bit-reproducible across hosts).  ``profile``: see PROFILES."""
    man = manifest(cfg)
    done = OrderedDict()
    deferred = []
    for name, (shape, dtype, init) in man.items():
        if init[0] == "wn_g":
            deferred.append(name)
            done[name] = None
            continue
        done[name] = _generate(name, shape, dtype, init, cfg, seed, done)
    for name in deferred:
        shape, dtype, init = man[name]
        done[name] = _generate(name, shape, dtype, init, cfg, seed, done)
    apply_profile(done, cfg, profile, seed)
    if as_torch:
        import torch
        return OrderedDict((k, torch.from_numpy(v)) for k, v in done.items())
    return done


def check_state_dict(cfg: ARTalkConfig, sd):
    """``strict=True`` semantics of reference ``inference.py:28``: returns (missing, unexpected, bad_shape)."""
    man = manifest(cfg)
    missing = [k for k in man if k not in sd]
    unexpected = [k for k in sd if k not in man]
    bad = [(k, tuple(sd[k].shape), man[k][0]) for k in man if k in sd and tuple(sd[k].shape) != tuple(man[k][0])]
    return missing, unexpected, bad


def fingerprint(sd, names=("pos_embed", "attn_blocks.0.attn.query.weight",
                           "audio_encoder.encoder.layers.0.attention.q_proj.weight",
                           "basic_vae.decoder.out_mapping.weight")):
    """Small checksum used by the golden fixtures to detect RNG drift between hosts."""
    out = {}
    for k in names:
        if k in sd:
            a = np.asarray(sd[k], dtype=np.float32).reshape(-1)
            out[k] = [float(a[:4096].astype(np.float64).sum()), float(a[-1])]
    return out
