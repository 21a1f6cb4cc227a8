#!/usr/bin/env python
"""Stage-by-stage diff of the HIP path against a golden case (prints, never asserts). Usage: gpu_debug.py [case]"""
import os, sys, time
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import get_gpu_model, get_state_dict, golden_inputs, load_golden

case = sys.argv[1] if len(sys.argv) > 1 else "tiny_4s_s0"
g = load_golden(case)
name = case.split("_")[0]
t = time.time(); m = get_gpu_model(name); print("model load", time.time() - t, "s; weights", m.weight_bytes() / 1e9, "GB")
cfg, sd = get_state_dict(name)
audio, style = golden_inputs(g, sd)
if os.environ.get("NOGRAPH"): m.set_graphs(False)
t = time.time(); out = m.inference_batch([audio], [style], return_aux=True)[0].cpu().numpy(); print("infer", time.time() - t, "s")
aux = m.last_aux
w2v = aux["w2v"].cpu().numpy()
print("w2v finite", np.isfinite(w2v).all(), "slice err", np.abs(w2v[:, :, :16] - g["w2v_slice"]).max(), "absmean", np.abs(w2v).mean(), "gold", float(g["w2v_abs_mean"]))
bits = aux["bits"][0].cpu().numpy(); hist = aux["hist_bits"][0].cpu().numpy()
gb = np.unpackbits(g["bits"], axis=-1); gh = np.unpackbits(g["hist_bits"], axis=-1)
for c in range(gh.shape[0]):
    d = hist[c] != gh[c]
    print(f"hist[{c}] mismatches {d.sum()} / {d.size}", "max margin at mismatch", g["hist_margin"][c][d].max() if d.any() else 0)
    if c < gb.shape[0]:
        d = bits[c] != gb[c]
        lv = np.concatenate([np.full(p, i) for i, p in enumerate((1, 5, 25, 50, 100))])
        print(f"bits[{c}] mismatches {d.sum()} / {d.size} per level {[int(d[lv == i].sum()) for i in range(5)]}", "max margin at mismatch", g["logit_margin"][c][d].max() if d.any() else 0)
print("out finite", np.isfinite(out).all(), "FLAME max err", np.abs(out - g["out"]).max(), "per chunk", [float(np.abs(out[i:i+100] - g["out"][i:i+100]).max()) for i in range(0, out.shape[0], 100)])
