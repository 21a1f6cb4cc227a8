#!/usr/bin/env python
"""Reference-captured INTERMEDIATES (build container only): ``tests/golden/taps_<case>.npz``.

Runs the reference's ``BitwiseARModel`` itself (loaded as in ``oracle/make_golden.py``) on two existing golden cases and
records, with forward hooks, the tensors that the device exposes through ``artalk_set_tap`` (include/artalk_hip.h):

  blk0_in   (chunks, 181, 768)  ``attn_feat`` entering ``attn_blocks[0]``                    app/models.py:100
  blk0_out  (chunks, 181, 768)  output of ``attn_blocks[0]``                                 app/transformer.py:30-43
  blkL_out  (chunks, 181, 768)  output of the last block                                     app/models.py:101-102
  prev_in   (chunks, 181, 768)  ``prev_attn_feat + prev_lvl_pos_embed`` as chunk j uses it   app/models.py:101,111-114
  logits    (chunks, 181, 64)   ``pred_motion_logits`` (fp32)                                app/models.py:103
  dec_out   (chunks, 200, 106)  decoder output before ``unnorm_with_stats``                  app/modules/bitwise_vae.py:110-111
  style_cond (768,)             ``motion_style_cond``                                        app/models.py:67-73

The reference re-runs all tokens of levels <= current in every scale step; row t of the block / logit tensors is taken from the
step that introduces token t (the values the bits of that token are decided from).  ``recompute_max_abs`` records how much the
rows of earlier tokens move when the reference recomputes them in later steps (its BLAS blocks by row count: ~1e-6, no decision
of these cases changes; with a KV cache a token is computed once).

Only these arrays are committed (data, no reference source).  Usage:  python oracle/make_golden_taps.py
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))
from artalk_amd.config import ARTalkConfig            # noqa: E402
from artalk_amd.synth import synth_audio, synth_style  # noqa: E402
from artalk_amd.weights import generate_state_dict, DEFAULT_SEED  # noqa: E402
from make_golden import load_reference_model           # noqa: E402

# (case of tests/golden, config, seed, seconds, styled): two chunks or more, so that prev_in after chunk 0 is covered
CASES = [("tiny_10s_s1_style", "tiny", 1, 10.0, True), ("full_5p5s_s3_style", "full", 3, 5.5, True)]
MAX_CHUNKS, COL_STRIDE = 2, 4
PN = (1, 5, 25, 50, 100)
OFF = (0, 1, 6, 31, 81, 181)


def capture(model, sd, seed, seconds, with_style):
    audio = torch.from_numpy(synth_audio(seed, seconds))[None]
    style = torch.from_numpy(synth_style(seed, sd["basic_vae.motion_mean"].numpy(), sd["basic_vae.motion_std"].numpy()))[None] if with_style else None
    rec = dict(b0_in=[], b0_prev=[], b0_out=[], bl_out=[], logits=[], dec=[], sc=[])
    first, last = model.attn_blocks[0], model.attn_blocks[len(model.attn_blocks) - 1]
    hooks = [
        first.register_forward_pre_hook(lambda m, a: (rec["b0_in"].append(a[0].detach().clone()), rec["b0_prev"].append(a[1].detach().clone()))[0] and None),
        first.register_forward_hook(lambda m, i, o: rec["b0_out"].append(o.detach().clone())),
        last.register_forward_hook(lambda m, i, o: rec["bl_out"].append(o.detach().clone())),
        model.logits_head.register_forward_hook(lambda m, i, o: rec["logits"].append(o.detach().clone())),
        model.basic_vae.decoder.register_forward_hook(lambda m, i, o: rec["dec"].append(o.detach().clone())),
        model.style_cond_embed.register_forward_hook(lambda m, i, o: rec["sc"].append(o.detach().clone())),
    ]
    with torch.no_grad():
        out = model.inference({"audio": audio, "style_motion": style})[0]
    for h in hooks:
        h.remove()
    n_chunks = len(rec["dec"])
    assert len(rec["b0_in"]) == len(rec["b0_out"]) == len(rec["bl_out"]) == len(rec["logits"]) == 5 * n_chunks
    g = dict(out=out.numpy())
    recompute = 0.0
    for name, src in (("blk0_in", rec["b0_in"]), ("blk0_out", rec["b0_out"]), ("blkL_out", rec["bl_out"]), ("logits", rec["logits"])):
        rows = []
        for c in range(n_chunks):
            full = torch.zeros(181, src[0].shape[-1])
            for p in range(5):
                t = src[5 * c + p][0]                       # (L_p, C): all tokens of levels <= p
                assert t.shape[0] == OFF[p + 1]
                full[OFF[p]:OFF[p + 1]] = t[OFF[p]:OFF[p + 1]]
                if p:                                       # rows of earlier tokens, recomputed by the reference in this step
                    recompute = max(recompute, float((t[:OFF[p]] - full[:OFF[p]]).abs().max()))
            rows.append(full)
        g[name] = torch.stack(rows).numpy().astype(np.float32)
    g["prev_in"] = torch.stack([rec["b0_prev"][5 * c][0] for c in range(n_chunks)]).numpy().astype(np.float32)
    # fixture size: the first MAX_CHUNKS chunks, every COL_STRIDE-th column of the 768-wide fields (every tile and wave of every
    # producing kernel still has columns in the sample); logits and decoder output in full
    for name in ("blk0_in", "blk0_out", "blkL_out", "prev_in"):
        g[name] = np.ascontiguousarray(g[name][:MAX_CHUNKS, :, ::COL_STRIDE])
    g["logits"] = g["logits"][:MAX_CHUNKS]
    g["col_stride"] = np.int64(COL_STRIDE)
    g["dec_out"] = torch.stack([d[0] for d in rec["dec"]]).numpy().astype(np.float32)[:MAX_CHUNKS]
    if with_style:
        sc = rec["sc"][0][:, None] * 1.1 - model.null_style_cond.detach() * 0.1      # app/models.py:69-70
        g["style_cond"] = sc.reshape(-1).numpy().astype(np.float32)
    else:
        g["style_cond"] = model.null_style_cond.detach().reshape(-1).numpy().astype(np.float32)
    g["recompute_max_abs"] = np.float64(recompute)
    g["seed"], g["seconds"], g["with_style"] = np.int64(seed), np.float64(seconds), np.bool_(with_style)
    g["weights_seed"] = np.int64(DEFAULT_SEED)
    return g


def main():
    out_dir = os.path.join(REPO, "tests", "golden")
    torch.manual_seed(0)
    models = {}
    for case, cfg_name, seed, seconds, with_style in CASES:
        if cfg_name not in models:
            cfg = ARTalkConfig.by_name(cfg_name)
            sd = generate_state_dict(cfg, DEFAULT_SEED)
            models[cfg_name] = (load_reference_model(cfg, sd), sd)
        model, sd = models[cfg_name]
        g = capture(model, sd, seed, seconds, with_style)
        ref = np.load(os.path.join(out_dir, case + ".npz"))
        assert np.array_equal(g["out"], ref["out"]), "this run of the reference does not reproduce the committed golden of the same case"
        del g["out"]
        path = os.path.join(out_dir, "taps_" + case + ".npz")
        np.savez_compressed(path, **g)
        print(f"  taps_{case}: {g['blk0_in'].shape[0]} chunks, recomputed rows move by {float(g['recompute_max_abs']):.1e} "
              f"-> {os.path.getsize(path) / 1024:.0f} KiB", flush=True)


if __name__ == "__main__":
    main()
