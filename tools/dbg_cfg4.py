"""Debug helper: where does the configs[4] batch differ from the reference goldens, and under which engine settings."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import clip_set_inputs, decision_parity, get_gpu_model, get_state_dict, load_clip_set

clips = load_clip_set("full_cfg4_demo32")
cfg, sd = get_state_dict("full")
audios, styles = clip_set_inputs(clips, sd)
m = get_gpu_model("full")

def report(tag, idx, outs, bits, hist):
    bad = []
    for k, i in enumerate(idx):
        c = clips[i]
        good, diff = decision_parity(bits[k], hist[k], c["bits"], c["hist_bits"], c["logit_margin"], c["hist_margin"])
        n = min(good, c["bits"].shape[0]) * 100
        n = min(n, c["out"].shape[0])
        err = float(np.abs(outs[k][:n] - c["out"][:n]).max()) if n else 0.0
        if diff is not None or err > 1e-3:
            bad.append((i, c["kind"], good, c["bits"].shape[0], f"{err:.2e}", diff))
    print(f"== {tag}: {len(bad)} bad of {len(idx)}", flush=True)
    for b in bad:
        print("   ", b, flush=True)

def run(idx, use_styles=True):
    outs = m.inference_batch([audios[i] for i in idx], [styles[i] for i in idx] if use_styles else None, return_aux=True)
    return [o.cpu().numpy() for o in outs], [b.cpu().numpy() for b in m.last_aux["bits"]], [h.cpu().numpy() for h in m.last_aux["hist_bits"]]

allidx = list(range(32))
for prec in ("f32", "f16x3"):
    m.set_precision(prec)
    report(f"{prec} batch32", allidx, *run(allidx))
    m.set_graphs(True, 1)
    report(f"{prec} batch32 one branch", allidx, *run(allidx))
    m.set_graphs(True, 0)
    report(f"{prec} single clip 5", [5], *run([5]))
m.set_precision("f16x3")
o, b, h = run([5])
c = clips[5]
for ci in range(c["hist_bits"].shape[0]):
    d = np.argwhere(h[0][ci] != c["hist_bits"][ci])
    print("hist", ci, "diff positions", d.tolist()[:12], len(d))
