#!/usr/bin/env python
"""Generate ``tests/golden/*.npz`` by running the REFERENCE itself (build container only).

Imports ``BitwiseARModel`` from ``/root/reference`` (SURVEY.md Appendix A recipe: stub the absent,
unused ``torchvision``/``torchaudio`` imports and replace the by-name hub fetch of the XLS-R config
at ``app/models.py:25`` with the local JSON), loads the deterministic synthetic weights with
``strict=True`` and records, per case: the FLAME codes, the per-chunk bits, the history bits, the
decision margins, a slice of the wav2vec2 features and the Savitzky-Golay-smoothed engine output.

This script never runs on the GPU box (``/root/reference`` does not exist there); only its small
outputs are committed.  Usage:  python oracle/make_golden.py [--cases tiny,full] [--out tests/golden]
"""
from __future__ import annotations

import argparse
import os
import sys
import time
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from artalk_amd.config import ARTalkConfig            # noqa: E402
from artalk_amd.synth import synth_audio, synth_style  # noqa: E402
from artalk_amd.weights import generate_state_dict, fingerprint, DEFAULT_SEED  # noqa: E402

REFERENCE = "/root/reference"

# (case name, config, audio seed, seconds, with style)
CASES = [
    ("tiny_4s_s0", "tiny", 0, 4.0, False),
    ("tiny_10s_s1_style", "tiny", 1, 10.0, True),
    ("tiny_6p3s_s2", "tiny", 2, 6.3, False),          # ragged: 158 frames, last chunk zero padded
    ("full_10s_s0", "full", 0, 10.0, False),
    ("full_10s_s1_style", "full", 1, 10.0, True),
    ("full_4s_s2", "full", 2, 4.0, False),
    ("full_5p5s_s3_style", "full", 3, 5.5, True),
]


def load_reference_model(cfg: ARTalkConfig, sd):
    import transformers  # noqa: F401  (must precede the stubs: it probes torchvision.__spec__)
    from transformers import Wav2Vec2Config as HFConfig
    for name in ("torchvision", "torchaudio"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.dont_write_bytecode = True
    if REFERENCE not in sys.path:
        sys.path.insert(0, REFERENCE)
    import app.models as M

    w2v = dict(cfg.w2v)

    class LocalCfg:
        @staticmethod
        def from_pretrained(name, *a, **k):
            return HFConfig(**w2v)

    M.Wav2Vec2Config = LocalCfg
    model = M.BitwiseARModel(cfg.reference_dict()).eval()
    model.load_state_dict(sd, strict=True)
    return model


def run_case(model, sd, seed, seconds, with_style, audio_np=None):
    audio = torch.from_numpy(synth_audio(seed, seconds) if audio_np is None else audio_np)[None]
    style = None
    if with_style:
        style = torch.from_numpy(synth_style(seed, sd["basic_vae.motion_mean"].numpy(), sd["basic_vae.motion_std"].numpy()))[None]
    rec = dict(logits=[], quant_bits=[], bsq_in=[], w2v=[])
    hooks = [
        model.logits_head.register_forward_hook(lambda m, i, o: rec["logits"].append(o.detach().clone())),
        model.basic_vae.quantize.register_forward_hook(lambda m, i, o: rec["quant_bits"].append(o[1].detach().clone())),
        model.basic_vae.quantize.bsq_quant.register_forward_hook(lambda m, i, o: rec["bsq_in"].append(i[0].detach().clone())),
        model.audio_encoder.register_forward_hook(lambda m, i, o: rec["w2v"].append(o.detach().clone())),
    ]
    t0 = time.time()
    with torch.no_grad():
        out = model.inference({"audio": audio, "style_motion": style})[0]
    dt = time.time() - t0
    for h in hooks:
        h.remove()
    n_chunks = len(rec["w2v"])
    assert len(rec["logits"]) == 5 * n_chunks and len(rec["quant_bits"]) == n_chunks + 1
    last = rec["logits"][4::5]                                     # last scale step of each chunk: (1,181,64)
    pairs = torch.stack([l.view(181, 32, 2) for l in last])       # (chunks,181,32,2)
    bits = pairs.argmax(-1).to(torch.uint8)
    logit_margin = (pairs[..., 0] - pairs[..., 1]).abs()
    hist_bits = torch.stack([b[0] for b in rec["quant_bits"]]).to(torch.uint8)   # (chunks+1,181,32)
    zs = [torch.nn.functional.normalize(x[0], dim=-1).abs() for x in rec["bsq_in"]]   # 5 per quantize call
    hist_margin = torch.stack([torch.cat(zs[5 * i:5 * i + 5], dim=0) for i in range(n_chunks + 1)])
    w2v = torch.stack([x[0] for x in rec["w2v"]])                 # (chunks,199,1024)
    # engine-level post-processing, reference inference.py:52-56,89-95
    from scipy.signal import savgol_filter
    m = out.numpy()
    sm = savgol_filter(m, window_length=5, polyorder=2, axis=0)
    sm[..., 100:103] = savgol_filter(m[..., 100:103], window_length=9, polyorder=3, axis=0)
    eng = sm[:750].copy()
    eng[..., 104:] *= 0.0
    return dict(
        out=out.numpy().astype(np.float32),
        engine_out=eng.astype(np.float32),
        bits=np.packbits(bits.numpy(), axis=-1),
        hist_bits=np.packbits(hist_bits.numpy(), axis=-1),
        logit_margin=logit_margin.numpy().astype(np.float16),
        hist_margin=hist_margin.numpy().astype(np.float16),
        w2v_slice=w2v[:, :, :16].numpy().astype(np.float32),
        w2v_abs_mean=np.float64(w2v.abs().double().mean().item()),
        w2v_sum=np.float64(w2v.double().sum().item()),
        seed=np.int64(seed), seconds=np.float64(seconds), with_style=np.bool_(with_style),
        ref_seconds=np.float64(dt),
    )


# (demo/<name>.wav, style seed, with style): all six clips of the reference's demo/ directory (3.4 - 13.8 s, 1 - 4 chunks)
DEMO = [("eng1", 4, True), ("eng2", 5, False), ("cn1", 6, False), ("cn2", 7, True), ("jp1", 8, True), ("jp2", 9, False)]
DEMO_NAMES = ["cn1", "cn2", "eng1", "eng2", "jp1", "jp2"]      # sorted(demo/*.wav): the order config 5 cycles through


def demo_inputs():
    """demo/*.wav -> 16 kHz mono with this repo's restatement of the torchaudio resampler (oracle/audio_oracle.py), quantised
    to int16 so that the input itself is a small committed fixture (tests/golden/demo_16k_s16.npz)."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    from audio_oracle import load_mono_16k
    from artalk_amd.audio import read_wav
    inputs = {}
    for name in DEMO_NAMES:
        wav, sr = read_wav(os.path.join(REFERENCE, "demo", name + ".wav"))
        a = load_mono_16k(wav, sr).numpy()
        inputs[name] = np.clip(np.round(a * 32768.0), -32768, 32767).astype(np.int16)
    return inputs


def stamp(g, fp):
    g["weights_seed"] = np.int64(DEFAULT_SEED)
    g["weights_fingerprint_keys"] = np.array(list(fp.keys()))
    g["weights_fingerprint"] = np.array([fp[k] for k in fp], dtype=np.float64)
    return g


def demo_cases(model, sd, fp, out_dir, inputs):
    """Real speech (configs 1 and 5 of BASELINE.json): the reference runs on exactly the committed 16 kHz arrays."""
    for name, seed, with_style in DEMO:
        q = inputs[name]
        audio = (q.astype(np.float32) / np.float32(32768.0))
        g = stamp(run_case(model, sd, seed, len(q) / 16000.0, with_style, audio_np=audio), fp)
        g["demo"] = np.array(name)
        path = os.path.join(out_dir, f"full_demo_{name}.npz")
        np.savez_compressed(path, **g)
        print(f"  full_demo_{name}: frames={g['out'].shape[0]} min logit margin={g['logit_margin'].min():.2e} "
              f"min hist margin={g['hist_margin'].min():.2e} -> {os.path.getsize(path) / 1024:.0f} KiB", flush=True)


SPARSE_TAU = 1e-3     # clip sets keep the reference's decision margins only where they are below this


def run_set(model, sd, fp, clips, path):
    """A set of clips in one compact fixture.  clips: list of (kind, key, style_seed | None) with kind 'demo' (key = wav name,
    audio from `inputs`) or 'synth' (key = (seed, seconds)).  Per clip: FLAME codes, packed bits / history bits; the decision
    margins are kept sparsely (positions below SPARSE_TAU), which is all a parity test needs of them."""
    outs, bits, hist, nfr, nch = [], [], [], [], []
    lm_idx, lm_val, hm_idx, hm_val = [], [], [], []
    c_off, h_off, t_ref = 0, 0, 0.0
    for kind, audio_np, seed, seconds, style_seed in clips:
        g = run_case(model, sd, seed if style_seed is None else style_seed, seconds, style_seed is not None, audio_np=audio_np)
        t_ref += float(g["ref_seconds"])
        outs.append(g["out"]); bits.append(g["bits"]); hist.append(g["hist_bits"])
        nfr.append(g["out"].shape[0]); nch.append(g["bits"].shape[0])
        for marg, idxs, vals, off in ((g["logit_margin"], lm_idx, lm_val, c_off), (g["hist_margin"], hm_idx, hm_val, h_off)):
            w = np.argwhere(marg.astype(np.float32) < SPARSE_TAU)
            vals.append(marg[tuple(w.T)].astype(np.float32))
            w[:, 0] += off
            idxs.append(w.astype(np.int32))
        c_off += nch[-1]; h_off += nch[-1] + 1
    g = stamp(dict(
        out=np.concatenate(outs), bits=np.concatenate(bits), hist_bits=np.concatenate(hist),
        n_frames=np.array(nfr, np.int32), n_chunks=np.array(nch, np.int32),
        logit_margin_idx=np.concatenate(lm_idx), logit_margin_val=np.concatenate(lm_val),
        hist_margin_idx=np.concatenate(hm_idx), hist_margin_val=np.concatenate(hm_val),
        sparse_tau=np.float64(SPARSE_TAU),
        kind=np.array([c[0] for c in clips]), seed=np.array([c[2] for c in clips], np.int64),
        seconds=np.array([c[3] for c in clips], np.float64),
        style_seed=np.array([-1 if c[4] is None else c[4] for c in clips], np.int64),
        ref_seconds=np.float64(t_ref)), fp)
    np.savez_compressed(path, **g)
    print(f"  {os.path.basename(path)}: {len(clips)} clips, {sum(nfr)} frames, {sum(nch)} chunks, reference took {t_ref:.0f} s; "
          f"decisions with margin < {SPARSE_TAU}: {len(g['logit_margin_val'])} logit / {len(g['hist_margin_val'])} history "
          f"-> {os.path.getsize(path) / 1024:.0f} KiB", flush=True)


def clip_sets(model, sd, fp, out_dir, inputs):
    # BASELINE configs[4]: 32 clips cycling the six demo wavs, every clip with its own synthetic style clip (seed 200 + i)
    clips = []
    for i in range(32):
        q = inputs[DEMO_NAMES[i % 6]]
        clips.append((DEMO_NAMES[i % 6], q.astype(np.float32) / np.float32(32768.0), i, len(q) / 16000.0, 200 + i))
    run_set(model, sd, fp, clips, os.path.join(out_dir, "full_cfg4_demo32.npz"))
    # BASELINE configs[2]: the first 8 of the 32 synthetic 10 s clips of the throughput run (seeds 0..7; odd seeds styled)
    clips = [("synth", None, i, 10.0, (i if i % 2 else None)) for i in range(8)]
    run_set(model, sd, fp, clips, os.path.join(out_dir, "full_cfg2_synth8.npz"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="tiny,full")
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    ap.add_argument("--only-demo", action="store_true")
    ap.add_argument("--only-sets", action="store_true")
    ap.add_argument("--no-sets", action="store_true")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    torch.manual_seed(0)
    for cfg_name in args.cases.split(","):
        cfg = ARTalkConfig.by_name(cfg_name)
        t0 = time.time()
        sd = generate_state_dict(cfg, DEFAULT_SEED)
        print(f"[{cfg_name}] weights generated in {time.time() - t0:.1f}s", flush=True)
        t0 = time.time()
        model = load_reference_model(cfg, sd)
        print(f"[{cfg_name}] reference constructed + strict load in {time.time() - t0:.1f}s", flush=True)
        fp = fingerprint(sd)
        for name, c, seed, seconds, with_style in CASES:
            if c != cfg_name or args.only_demo or args.only_sets:
                continue
            g = stamp(run_case(model, sd, seed, seconds, with_style), fp)
            g["versions"] = np.array([torch.__version__, np.__version__])
            path = os.path.join(args.out, name + ".npz")
            np.savez_compressed(path, **g)
            print(f"  {name}: frames={g['out'].shape[0]} ref_time={float(g['ref_seconds']):.2f}s "
                  f"min logit margin={g['logit_margin'].min():.2e} min hist margin={g['hist_margin'].min():.2e} "
                  f"-> {os.path.getsize(path) / 1024:.0f} KiB", flush=True)
        if cfg_name == "full":
            inputs = demo_inputs()
            np.savez_compressed(os.path.join(args.out, "demo_16k_s16.npz"), **inputs)
            print("  demo inputs ->", os.path.getsize(os.path.join(args.out, "demo_16k_s16.npz")) // 1024, "KiB", flush=True)
            if not args.only_sets:
                demo_cases(model, sd, fp, args.out, inputs)
            if not args.no_sets:
                clip_sets(model, sd, fp, args.out, inputs)
        del model, sd


if __name__ == "__main__":
    main()
