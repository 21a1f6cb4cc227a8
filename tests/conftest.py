import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


_cache = {}


def get_state_dict(name, profile="benign"):
    """Deterministic synthetic weights of config `name` ('tiny' | 'full'), generated once per session.  `profile`: the benign
    U(+-1/sqrt(fan_in)) weights of every round, or one of the outlier profiles of artalk_amd.weights.PROFILES."""
    key = name if profile == "benign" else (name, profile)
    if key not in _cache:
        from artalk_amd.config import ARTalkConfig
        from artalk_amd.weights import generate_state_dict
        cfg = ARTalkConfig.by_name(name)
        _cache[key] = (cfg, generate_state_dict(cfg, profile=profile))
    return _cache[key]


_models = {}


def get_gpu_model(name, profile="benign"):
    """HIP model with the synthetic weights loaded (one per config and weight profile per session)."""
    key = name if profile == "benign" else (name, profile)
    if key not in _models:
        from artalk_amd.model import BitwiseARModel
        cfg, sd = get_state_dict(name, profile)
        m = BitwiseARModel(cfg).eval().to("cuda")
        m.load_state_dict(sd, strict=True)
        _models[key] = m
    return _models[key]


def drop_profile(name, profile):
    """Free the host and device copies of a non-default weight profile (2 GB + 4 GB each at the full size)."""
    key = (name, profile)
    _models.pop(key, None)
    _cache.pop(key, None)
    _cache.pop(("oracle",) + key, None)


def get_oracle(name, profile="benign"):
    from artalk_oracle import ARTalkOracle
    cfg, sd = get_state_dict(name, profile)
    key = ("oracle", name) if profile == "benign" else ("oracle", name, profile)
    if key not in _cache:
        _cache[key] = ARTalkOracle(cfg, sd)
    return _cache[key]


def load_golden(case):
    import numpy as np
    return np.load(os.path.join(GOLDEN, case + ".npz"))


def golden_inputs(g, sd):
    """Rebuild the inputs of a golden case from its seeds."""
    import numpy as np
    import torch
    from artalk_amd.synth import synth_audio, synth_style
    if "demo" in g.files:      # real speech: the committed 16 kHz int16 array is the input definition
        q = np.load(os.path.join(GOLDEN, "demo_16k_s16.npz"))[str(g["demo"])]
        audio = torch.from_numpy(q.astype(np.float32) / np.float32(32768.0))
    else:
        audio = torch.from_numpy(synth_audio(int(g["seed"]), float(g["seconds"])))
    style = None
    if bool(g["with_style"]):
        style = torch.from_numpy(synth_style(int(g["seed"]), sd["basic_vae.motion_mean"].numpy(), sd["basic_vae.motion_std"].numpy()))
    return audio, style


# ---------------------------------------------------------------------------------------------- decision parity (shared)
# Bar (BASELINE.json north_star): bit decisions EXACT, FLAME codes within 1e-3.  fp32 summation order differs from the
# reference's MKL kernels, so a decision can legitimately differ only where the reference's own margin is at rounding level;
# such a flip is tolerated only (a) at a reference margin below these thresholds AND (b) with the rest of the clip equal to the
# forced-decision continuation (assert_clip_parity: an alt fixture, or the pinned CPU oracle run in the test with that decision
# inverted).  ALLOWED_MARGINAL (a bare allowance without a continuation) is empty.
TAU_LOGIT = 2e-5     # |l0 - l1| of the reference at a flipped AR bit (the reference's own fixtures reach down to 1.1e-5)
TAU_HIST = 2e-6      # |z| (unit-normalised) of the reference at a flipped history bit
FLAME_TOL = 1e-3
# Regression guard beside the bar: every run of rounds 1-3 measured 3e-7 .. 8.3e-7 on the FLAME codes of decision-exact chunks, so a
# kernel that loses a decimal digit fails HERE, before it starts to re-roll which clips flip.
FLAME_GUARD = 5e-6
ALLOWED_MARGINAL = {}    # {(case, precision): first chunk allowed to differ}

# The PINNED set of rounding-level clips: {"<clip tag>|<precision>": stages}.  A clip that is not listed must equal the
# reference golden outright; a listed clip may follow at most the listed number of chained rounding-level flips.  A change of
# arithmetic that re-rolls the set therefore FAILS until the file is edited deliberately (ARTALK_ROUNDING_DISCOVER=1 records what
# a run observes in gpurun_out/rounding_level_observed.json without failing on the pin; the margin / continuation bar still holds).
ROUNDING_PIN_FILE = os.path.join(GOLDEN, "rounding_level_expected.json")
ROUNDING_OBSERVED = {}      # filled by assert_clip_parity, written at session end


def rounding_pin():
    import json
    if "pin" not in _cache:
        _cache["pin"] = json.load(open(ROUNDING_PIN_FILE))["clips"] if os.path.exists(ROUNDING_PIN_FILE) else {}
    return _cache["pin"]


def clip_key(tag, precision):
    """'cfg4 clip 5 (jp2, style 205)' -> 'cfg4 clip 5|f16x3': the clip and the batch it runs in (tile shapes, hence the fp32 summation
    order, depend on the batch), not the free-text suffix."""
    return tag.split(" (")[0] + "|" + precision


def pytest_sessionfinish(session, exitstatus):
    if not ROUNDING_OBSERVED:
        return
    import json
    try:      # (a record for the builder, never a reason for the session to fail: the checkout may be read-only)
        d = os.path.join(REPO, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "rounding_level_observed.json"), "w") as f:
            json.dump({"clips": {k: v for k, v in sorted(ROUNDING_OBSERVED.items()) if v["stages"]},
                       "checked": len(ROUNDING_OBSERVED)}, f, indent=1)
    except OSError:
        pass


_LEVEL_OF = None


def _level_masks():
    global _LEVEL_OF
    if _LEVEL_OF is None:
        import numpy as np
        lv = np.concatenate([np.full(p, i) for i, p in enumerate((1, 5, 25, 50, 100))])
        _LEVEL_OF = [lv == i for i in range(5)]
    return _LEVEL_OF


def decision_parity(bits, hist, gbits, ghist, logit_margin, hist_margin):
    """bits/gbits (chunks,181,32), hist/ghist (chunks+1,181,32) 0/1 arrays; margins: callables (chunk, mask[181,32]) -> margins
    of the reference at the masked decisions.  Walks the decision GROUPS in causal order - hist[0] levels 0..4 (each level of the
    multi-scale quantiser works on the residual the coarser levels left: bitwise_vae.py:227-242), bits[0] levels 0..4 (each scale
    step sees the bits of the earlier ones: app/models.py:97-107), hist[1], ... - and returns (chunks that are decision-exact
    before the first differing group, description of that group or None).  Everything after the first differing group depends
    on it, so only that group's margins say whether the difference is a rounding-level flip."""
    import numpy as np
    n_chunks = gbits.shape[0]
    levels = _level_masks()
    for c in range(n_chunks + 1):
        groups = [("history", hist, ghist, hist_margin, TAU_HIST)]
        if c < n_chunks:
            groups.append(("AR", bits, gbits, logit_margin, TAU_LOGIT))
        for name, mine, gold, marg, tau in groups:
            for lv, sel in enumerate(levels):
                d = (mine[c] != gold[c]) & sel[:, None]
                if d.any():
                    mg = np.asarray(marg(c, d), dtype=np.float64)
                    pos = [tuple(int(v) for v in x) for x in np.argwhere(d)]
                    return c, dict(kind=name, chunk=c, level=lv, count=int(d.sum()), max_margin=float(mg.max()),
                                   marginal=bool((mg < tau).all()), positions=pos[:4], all_positions=pos)
    return n_chunks, None


MAX_FLIP_STAGES = 3      # a clip may follow at most this many chained rounding-level flips (the known worst case has two)


def oracle_continuation(config_name, audio, style, force_hist, force_bits):
    """What the reference's arithmetic gives for this clip WITH the listed decisions inverted: the CPU oracle (pinned bit for bit
    to the reference on every fixture, tests/test_oracle_golden.py) run with those decisions forced.  In the format of a golden."""
    import numpy as np
    import torch
    rec = {}
    o = get_oracle(*config_name) if isinstance(config_name, tuple) else get_oracle(config_name)      # (name, weight profile) or name
    out = o.inference({"audio": audio[None], "style_motion": style[None] if style is not None else None}, record=rec,
                      force_hist=force_hist, force_bits=force_bits)[0].numpy()
    return dict(out=out, bits=torch.cat(rec["bits"]).numpy().astype(np.uint8), hist_bits=torch.cat(rec["hist_bits"]).numpy().astype(np.uint8),
                logit_margin=dense_margins(torch.cat(rec["logit_margin"]).numpy()), hist_margin=dense_margins(torch.cat(rec["hist_margin"]).numpy()))


def assert_clip_parity(tag, precision, out, bits, hist, gout, gbits, ghist, logit_margin, hist_margin, alt=None, inputs=None):
    """Full-clip parity of one clip against its reference golden: every chunk decision-exact, FLAME codes within FLAME_TOL over
    all frames.  The one tolerated deviation is a ROUNDING-LEVEL flip: the first differing decision group consists only of
    decisions whose reference margin is below TAU_LOGIT / TAU_HIST (the reference's own sign test there is decided by its fp32
    summation order; the goldens hold such margins down to < 3e-8).  Everything later in the clip depends on that decision, so
    the rest of the clip is then held, to the same bar, to the continuation the reference's arithmetic gives WITH that decision
    inverted: a pre-computed ``alt`` fixture (oracle/make_alt_golden.py; a list = a chain of stages) when one is listed, else -
    with ``inputs`` = (config name, audio, style) - the pinned CPU oracle run here with the decision forced
    (``oracle_continuation``), chained up to MAX_FLIP_STAGES times.  A difference at a margin above the thresholds always fails.
    A rounding-level difference in the trailing history (it feeds nothing that is returned) is tolerated as such.
    Sets ``assert_clip_parity.last_rounding_level`` (stages followed) for the callers' summaries.
    Returns (decision-exact chunks, n_chunks, err)."""
    import numpy as np
    n_chunks = gbits.shape[0]
    assert out.shape == gout.shape, f"{tag} [{precision}]: shape {out.shape} vs {gout.shape}"
    good, diff = decision_parity(bits, hist, gbits, ghist, logit_margin, hist_margin)
    note, stages = "", 0
    alts = list(alt or [])
    force_hist, force_bits = {}, {}
    while diff is not None and diff["chunk"] < n_chunks and diff["marginal"] and stages < MAX_FLIP_STAGES:
        a = None
        if alts and diff["kind"] == "history" and diff["chunk"] == alts[0]["forced_hist"] and diff["positions"][0] == alts[0]["forced_pos"] \
                and diff["count"] == 1:
            a = alts.pop(0)
        (force_hist if diff["kind"] == "history" else force_bits).setdefault(diff["chunk"], []).extend(
            (t, b) for (t, b) in diff["all_positions"])
        if a is None:
            if inputs is None:
                break
            alts = []
            a = oracle_continuation(inputs[0], inputs[1], inputs[2], force_hist, force_bits)
        stages += 1
        note += (f" (after the rounding-level flip of {diff['count']} {diff['kind']} decision(s) at chunk {diff['chunk']} level {diff['level']} "
                 f"{diff['positions'][0]}, reference margin {diff['max_margin']:.1e})")
        gout, gbits, ghist, logit_margin, hist_margin = a["out"], a["bits"], a["hist_bits"], a["logit_margin"], a["hist_margin"]
        good, diff = decision_parity(bits, hist, gbits, ghist, logit_margin, hist_margin)
    assert_clip_parity.last_rounding_level = stages
    assert_clip_parity.last_expected_out = gout      # the reference golden's codes, or the forced-decision continuation's the clip was held to
    good = min(good, n_chunks)
    n = min(good * 100, gout.shape[0])
    err = float(np.abs(out[:n] - gout[:n]).max()) if n else 0.0
    msg = (f"{tag} [{precision}]: chunks decision-exact {good}/{n_chunks}, FLAME max-abs err {err:.3e} over {n} frames, "
           f"first difference: {diff}{note}")
    if stages:
        print("NOTE " + msg)
    assert err < FLAME_TOL, msg
    assert err < FLAME_GUARD, "precision regression (bar 1e-3 still met, guard = measured 8e-7 + margin): " + msg
    key = clip_key(tag, precision)
    ROUNDING_OBSERVED[key] = dict(stages=stages, note=note.strip())
    pinned = int(rounding_pin().get(key, 0))
    if stages < pinned:
        print(f"NOTE {key}: pinned at {pinned} rounding-level stage(s), this run followed {stages} - the pin can be tightened")
    if not os.environ.get("ARTALK_ROUNDING_DISCOVER"):
        assert stages <= pinned, (f"{key} followed {stages} rounding-level flip(s) but tests/golden/rounding_level_expected.json pins "
                                  f"{pinned}: a change of arithmetic re-rolled the set - check the cause, then edit the pin deliberately. " + msg)
    if diff is not None and diff["chunk"] < n_chunks:
        allowed = ALLOWED_MARGINAL.get((tag, precision), ALLOWED_MARGINAL.get((tag, "*")))
        assert diff["marginal"] and allowed is not None and diff["chunk"] >= allowed, msg
    elif diff is not None:
        assert diff["marginal"], msg      # trailing history: must still be a rounding-level flip
    return good, n_chunks, err


assert_clip_parity.last_rounding_level = 0
assert_clip_parity.last_expected_out = None


def load_alt(name):
    """A forced-decision continuation written by oracle/make_alt_golden.py."""
    import numpy as np
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    tau = float(g["sparse_tau"])
    return dict(out=g["out"], bits=np.unpackbits(g["bits"], axis=-1), hist_bits=np.unpackbits(g["hist_bits"], axis=-1),
                logit_margin=sparse_margins(g["logit_margin_idx"], g["logit_margin_val"], 0, tau),
                hist_margin=sparse_margins(g["hist_margin_idx"], g["hist_margin_val"], 0, tau),
                forced_hist=int(g["forced_hist"]), forced_pos=tuple(int(v) for v in g["forced_pos"][-1]), clip=int(g["clip"]))


def dense_margins(margin):
    """margin (chunks,181,32) array -> callable for decision_parity."""
    return lambda c, mask: margin[c][mask]


def sparse_margins(idx, val, first_row, tau):
    """Sparse margins of a clip-set fixture (rows with margin < tau are listed, everything else is >= tau)."""
    import numpy as np
    table = {(int(r), int(t), int(b)): float(v) for (r, t, b), v in zip(idx, val)}

    def f(c, mask):
        tb = np.argwhere(mask)
        return np.array([table.get((first_row + c, int(t), int(b)), tau) for t, b in tb])
    return f


def load_clip_set(name):
    """A clip-set fixture (oracle/make_golden.py::run_set) as a list of per-clip dicts."""
    import numpy as np
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    clips, f0, c0, h0 = [], 0, 0, 0
    gb, gh = np.unpackbits(g["bits"], axis=-1), np.unpackbits(g["hist_bits"], axis=-1)
    tau = float(g["sparse_tau"])
    for i in range(len(g["n_frames"])):
        nf, nc = int(g["n_frames"][i]), int(g["n_chunks"][i])
        clips.append(dict(
            out=g["out"][f0:f0 + nf], bits=gb[c0:c0 + nc], hist_bits=gh[h0:h0 + nc + 1],
            logit_margin=sparse_margins(g["logit_margin_idx"], g["logit_margin_val"], c0, tau),
            hist_margin=sparse_margins(g["hist_margin_idx"], g["hist_margin_val"], h0, tau),
            kind=str(g["kind"][i]), seed=int(g["seed"][i]), seconds=float(g["seconds"][i]), style_seed=int(g["style_seed"][i])))
        f0 += nf; c0 += nc; h0 += nc + 1
    return clips


def clip_set_inputs(clips, sd):
    """Audio / style tensors of a clip set, rebuilt from seeds and the committed demo arrays."""
    import numpy as np
    import torch
    from artalk_amd.synth import synth_audio, synth_style
    demo = None
    audios, styles = [], []
    mean, std = sd["basic_vae.motion_mean"].numpy(), sd["basic_vae.motion_std"].numpy()
    for c in clips:
        if c["kind"] == "synth":
            audios.append(torch.from_numpy(synth_audio(c["seed"], c["seconds"])))
        else:
            if demo is None:
                demo = np.load(os.path.join(GOLDEN, "demo_16k_s16.npz"))
            audios.append(torch.from_numpy(demo[c["kind"]].astype(np.float32) / np.float32(32768.0)))
        styles.append(torch.from_numpy(synth_style(c["style_seed"], mean, std)) if c["style_seed"] >= 0 else None)
    return audios, styles
