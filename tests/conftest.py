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


def get_state_dict(name):
    """Deterministic synthetic weights of config `name` ('tiny' | 'full'), generated once per session."""
    if name not in _cache:
        from artalk_amd.config import ARTalkConfig
        from artalk_amd.weights import generate_state_dict
        cfg = ARTalkConfig.by_name(name)
        _cache[name] = (cfg, generate_state_dict(cfg))
    return _cache[name]


_models = {}


def get_gpu_model(name):
    """HIP model with the synthetic weights loaded (one per config per session)."""
    if name not in _models:
        from artalk_amd.model import BitwiseARModel
        cfg, sd = get_state_dict(name)
        m = BitwiseARModel(cfg).eval().to("cuda")
        m.load_state_dict(sd, strict=True)
        _models[name] = m
    return _models[name]


def get_oracle(name):
    from artalk_oracle import ARTalkOracle
    cfg, sd = get_state_dict(name)
    key = ("oracle", name)
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
