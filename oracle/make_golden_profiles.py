#!/usr/bin/env python
"""Reference goldens for the OUTLIER weight profiles (build container only): ``tests/golden/<profile>_<case>.npz``.

The real checkpoint of reference ``inference.py:24-28`` cannot be fetched; ``artalk_amd.weights.PROFILES`` plants the structure
XLS-R-class checkpoints are known for (a few LayerNorm gains of 100, FFN rows x 50, spread pos-conv gains) into the same
manifest.  This script runs the REFERENCE itself on those weights, exactly as ``oracle/make_golden.py`` does for the benign
profile (same ``run_case``: FLAME codes, bits, history bits, decision margins, wav2vec2 feature slice).

  outlier: activations reach the hundreds to a thousand and every f16x3 operand stays inside its format (|x| < 4094)
  heavy:   FFN hidden activations of the encoder exceed the format: f16x3 mode must trip status bit 3, the f32 re-run must pass

Usage:  python oracle/make_golden_profiles.py
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
from artalk_amd.weights import generate_state_dict, fingerprint, DEFAULT_SEED  # noqa: E402
from make_golden import load_reference_model, run_case, stamp           # noqa: E402

# (profile, config, seed, seconds, styled)
CASES = [
    ("outlier", "tiny", 1, 10.0, True),
    ("outlier", "full", 2, 4.0, False),
    ("outlier", "full", 3, 5.5, True),
    ("heavy", "tiny", 2, 6.3, False),
    ("heavy", "full", 2, 4.0, False),
]


def case_name(profile, cfg_name, seed, seconds, styled):
    secs = ("%g" % seconds).replace(".", "p")
    return f"{profile}_{cfg_name}_{secs}s_s{seed}" + ("_style" if styled else "")


def main():
    out_dir = os.path.join(REPO, "tests", "golden")
    torch.manual_seed(0)
    cache = {}
    for profile, cfg_name, seed, seconds, styled in CASES:
        key = (profile, cfg_name)
        if key not in cache:
            cache.clear()
            cfg = ARTalkConfig.by_name(cfg_name)
            sd = generate_state_dict(cfg, DEFAULT_SEED, profile=profile)
            cache[key] = (load_reference_model(cfg, sd), sd, fingerprint(sd))
        model, sd, fp = cache[key]
        g = stamp(run_case(model, sd, seed, seconds, styled), fp)
        g["profile"] = np.array(profile)
        assert np.isfinite(g["out"]).all()
        name = case_name(profile, cfg_name, seed, seconds, styled)
        path = os.path.join(out_dir, name + ".npz")
        np.savez_compressed(path, **g)
        print(f"  {name}: frames={g['out'].shape[0]} |codes| max {np.abs(g['out']).max():.2f} min logit margin={g['logit_margin'].min():.2e} "
              f"min hist margin={g['hist_margin'].min():.2e} w2v |mean| {float(g['w2v_abs_mean']):.3f} -> {os.path.getsize(path) / 1024:.0f} KiB", flush=True)


if __name__ == "__main__":
    main()
