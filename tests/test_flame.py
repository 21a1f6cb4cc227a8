"""FLAME vertex path.  The golden comes from the reference's own lbs() on a deterministic synthetic
asset (oracle/make_golden_flame.py); the licensed FLAME_with_eye.pt is absent, so the asset values - not the arithmetic - are synthetic."""
import os

import numpy as np
import pytest
import torch

from artalk_amd.flame import synthetic_flame_asset
from conftest import GOLDEN
from flame_oracle import flame_forward


def _golden():
    g = np.load(os.path.join(GOLDEN, "flame_lbs.npz"))
    m = torch.from_numpy(g["motion"])
    return g, torch.from_numpy(g["shape"]), m[:, :100], m[:, 100:]


def test_oracle_matches_reference_lbs():
    g, shape, exp, pose = _golden()
    v = flame_forward(synthetic_flame_asset(), shape, exp, pose)
    assert v.shape == (6, 5023, 3)
    assert np.abs(v[:, ::37].numpy() - g["verts_sub"]).max() < 1e-6
    assert abs(v.double().sum().item() - float(g["verts_sum"])) < 1e-3


@pytest.mark.gpu
def test_hip_flame_matches_reference_lbs():
    from artalk_amd.flame import FLAMEModel
    g, shape, exp, pose = _golden()
    asset = synthetic_flame_asset()
    fm = FLAMEModel(n_shape=300, n_exp=100, scale=1.0, no_lmks=True, flame_ckpt=asset)
    v = fm(shape_params=shape, expression_params=exp, pose_params=pose).cpu()
    assert v.shape == (6, 5023, 3)
    assert np.abs(v[:, ::37].numpy() - g["verts_sub"]).max() < 2e-6
    want = flame_forward(asset, shape, exp, pose)
    assert (v - want).abs().max().item() < 2e-6
    # scale and the 3-component pose form of FLAME.py:127-128
    fm5 = FLAMEModel(n_shape=300, n_exp=100, scale=5.0, no_lmks=True, flame_ckpt=asset)
    v5 = fm5(shape_params=shape, expression_params=exp, pose_params=pose[:, 3:]).cpu()
    want5 = flame_forward(asset, shape, exp, torch.cat([torch.zeros(6, 3), pose[:, 3:]], 1), scale=5.0)
    assert (v5 - want5).abs().max().item() < 1e-5
    with pytest.raises(NotImplementedError):
        FLAMEModel(n_shape=300, n_exp=100, no_lmks=False, flame_ckpt=asset)
    with pytest.raises(FileNotFoundError):
        FLAMEModel(n_shape=300, n_exp=100, no_lmks=True, flame_ckpt="assets/FLAME_with_eye.pt")


@pytest.mark.gpu
def test_engine_rendering_returns_vertices_with_a_flame_model():
    """inference.py:62-69: mesh branch up to the vertices (rasterising is outside the package)."""
    from artalk_amd.engine import ARTAvatarInferEngine
    from artalk_amd.flame import FLAMEModel
    from artalk_amd.synth import synth_audio
    from conftest import get_gpu_model
    eng = ARTAvatarInferEngine.__new__(ARTAvatarInferEngine)
    eng.ARTalk, eng.device, eng.style_motion, eng.clip_length, eng.fix_pose = get_gpu_model("tiny"), "cuda", None, 40, False
    eng.flame_model = None
    audio = torch.from_numpy(synth_audio(3, 4.0))
    pred = eng.inference(audio)
    with pytest.raises(NotImplementedError):
        eng.rendering(audio, pred)
    eng.flame_model = FLAMEModel(n_shape=300, n_exp=100, scale=1.0, no_lmks=True, flame_ckpt=synthetic_flame_asset())
    verts = eng.rendering(audio, pred, shape_id="mesh")
    assert verts.shape == (40, 5023, 3) and torch.isfinite(verts).all()
    want = flame_forward(synthetic_flame_asset(), torch.zeros(40, 300), pred.cpu()[:, :100], pred.cpu()[:, 100:])
    assert (verts.cpu() - want).abs().max().item() < 1e-5
