import os, sys, numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import get_gpu_model
from artalk_amd.synth import synth_audio
m = get_gpu_model("tiny")
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 70
lens = [16000 + 5000 * (i % 31) for i in range(NB)]
audios = [torch.from_numpy(synth_audio(100 + i, 11.0))[:n].clone() for i, n in enumerate(lens)]
for prec in ("f32", "f16x3"):
    for graphs in (True, False):
        m.set_precision(prec); m.set_graphs(graphs)
        outs = m.inference_batch(audios, return_aux=True)
        w2v_b = m.last_aux["w2v"].clone(); idx = dict(m.last_aux["w2v_index"])
        bad = []
        for i in range(NB):
            single = m.inference_batch([audios[i]], return_aux=True)[0]
            w2v_s = m.last_aux["w2v"]
            e = (outs[i] - single).abs().max().item()
            ew = max((w2v_b[idx[(i, j)]] - w2v_s[j]).abs().max().item() for j in range(w2v_s.shape[0]))
            if e > 1e-4 or ew > 1e-3: bad.append((i, lens[i], round(e, 4), round(ew, 5)))
        print(prec, "graphs" if graphs else "eager", "bad clips:", bad[:12], "n_bad", len(bad), flush=True)
