#!/usr/bin/env python
"""Headroom of the f16x3 operand format ("P8": f16(S x) + residual, S = the site's power-of-two scale) on every golden input
-> profiles/rNN_p8_headroom[_profile].json.

For each producer site of a P8 operand (artalk_set_audit) the largest |x| seen over all fixtures, the site's scale (16 unless the
calibration lowered it: artalk_calibrate) and the factor left to fp16's 65504.  The weights are synthetic (no checkpoint is
available offline), so this documents the mechanism and the margin on the synthetic model; a real checkpoint is audited by the same
command.  ``--calibrate`` first runs the calibration pass (exact-f32 audit over the same clips, headroom 4) - what the Python host
does by itself when a call trips the range guard - and then audits the f16x3 run with the calibrated scales.
"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from artalk_amd import capi                                                       # noqa: E402
from conftest import (clip_set_inputs, get_gpu_model, get_state_dict, golden_inputs, load_clip_set, load_golden)   # noqa: E402

CASES = ["full_10s_s0", "full_10s_s1_style", "full_4s_s2", "full_5p5s_s3_style", "full_demo_eng1", "full_demo_eng2", "full_demo_cn1",
         "full_demo_cn2", "full_demo_jp1", "full_demo_jp2"]


def read_audit(m):
    L = capi.lib()
    buf = C.create_string_buffer(1 << 16)
    vals = (C.c_float * 1024)()
    n = L.artalk_get_audit(m._h, buf, len(buf), vals, 1024)
    assert n >= 0
    names = buf.raw.split(b"\0")[:n]
    exps = (C.c_int * 1024)()
    ne = L.artalk_get_scales(m._h, exps, 1024)
    assert ne >= n
    return {nm.decode(): (float(vals[i]), int(exps[i])) for i, nm in enumerate(names)}


def main():
    # usage: p8_headroom.py [out.json] [--profile outlier|heavy]   (a weight profile of artalk_amd.weights.PROFILES: its own goldens'
    # inputs plus the demo / synthetic clip sets, on the FULL model)
    args = [a for a in sys.argv[1:]]
    profile = "benign"
    if "--profile" in args:
        i = args.index("--profile")
        profile = args[i + 1]
        del args[i:i + 2]
    calibrate = "--calibrate" in args
    if calibrate:
        args.remove("--calibrate")
    out = args[0] if args else os.path.join(REPO, "profiles", "r05_p8_headroom.json")
    cfg, sd = get_state_dict("full", profile)
    m = get_gpu_model("full", profile)
    m.check_finite = False          # the audit records what the format is asked to hold; no re-run here
    L = capi.lib()
    cases = CASES if profile == "benign" else [f"{profile}_full_4s_s2"] + ([f"{profile}_full_5p5s_s3_style"] if profile == "outlier" else []) + CASES

    def run_all():
        n = 0
        for case in cases:
            g = load_golden(case)
            audio, style = golden_inputs(g, sd)
            m.inference_batch([audio], [style])
            n += 1
        for name in ("full_cfg4_demo32", "full_cfg2_synth8"):
            clips = load_clip_set(name)
            audios, styles = clip_set_inputs(clips, sd)
            m.inference_batch(audios, styles)
            n += len(clips)
        return n

    changed = None
    if calibrate:
        m.set_precision("f32")
        assert L.artalk_set_audit(m._h, 1) == 0
        run_all()
        changed = L.artalk_calibrate(m._h, C.c_float(4.0))
        assert changed >= 0, L.artalk_last_error(m._h).decode()
        L.artalk_set_audit(m._h, 0)
    m.set_precision("f16x3")
    assert L.artalk_set_audit(m._h, 1) == 0
    n_clips = run_all()
    table = read_audit(m)
    status = m.status()
    L.artalk_set_audit(m._h, 0)
    rows = sorted(table.items(), key=lambda t: -(t[1][0] * 2.0 ** t[1][1]))
    worst = rows[0]
    wv = worst[1][0] * 2.0 ** worst[1][1]
    res = {
        "what": "per producer site of a P8 (f16x3) operand over all golden inputs: max |x|, the site's scale exponent e (scale 2^e, default 4) "
                "and max |x| * 2^e against the limit 65504 (fp16 max)",
        "weights": f"deterministic synthetic weights (seed 1234, profile '{profile}': artalk_amd.weights.PROFILES), reference motion statistics", "clips": n_clips,
        "calibrated": calibrate, "sites_changed_by_calibration": changed, "status_word": status, "limit": 65504.0,
        "worst_site": worst[0], "worst_scaled_value": wv, "min_headroom_factor": 65504.0 / max(wv, 1e-30),
        "lowered_sites": {k: v[1] for k, v in rows if v[1] != 4},
        "sites": {k: {"max_abs": v[0], "exp": v[1], "max_abs_scaled": v[0] * 2.0 ** v[1], "headroom_factor": (65504.0 / (v[0] * 2.0 ** v[1]) if v[0] > 0 else None)} for k, v in rows},
    }
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(f"{len(rows)} sites over {n_clips} clips; worst {worst[0]} = {wv:.1f} scaled (headroom x{res['min_headroom_factor']:.1f}); status {status}; "
          f"calibration changed {changed} sites: {len(res['lowered_sites'])} below the default scale")
    for k, v in rows[:12]:
        print(f"  {k:50s} max|x| {v[0]:12.3f}  2^{v[1]:<3d} scaled {v[0] * 2.0 ** v[1]:10.2f}")


if __name__ == "__main__":
    main()
