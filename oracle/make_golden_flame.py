#!/usr/bin/env python
"""tests/golden/flame_lbs.npz from the REFERENCE's own lbs() (build container only): imports
/root/reference/app/flame_model/lbs.py by path (its package __init__ pulls pytorch3d, which is absent) and runs it on the
deterministic synthetic FLAME asset with inputs shaped like the engine's (inference.py:62-69: zero shape code, codes (T,106))."""
import importlib.util
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from artalk_amd.flame import synthetic_flame_asset  # noqa: E402

spec = importlib.util.spec_from_file_location("ref_lbs", "/root/reference/app/flame_model/lbs.py")
ref = importlib.util.module_from_spec(spec)
sys.dont_write_bytecode = True
spec.loader.exec_module(ref)

asset = synthetic_flame_asset()
fm = asset["flame_model"]
g = torch.Generator().manual_seed(3)
T = 6
shape = 0.5 * torch.randn(T, 300, generator=g)
shape[0] = 0.0                                      # the engine's default shape code (inference.py:63)
motion = torch.cat([torch.randn(T, 100, generator=g), 0.3 * torch.randn(T, 6, generator=g)], dim=1)   # (T,106)
exp, pose = motion[:, :100], motion[:, 100:]
sd = fm["shapedirs"]
shapedirs = torch.cat([sd[:, :, :300], sd[:, :, 300:400]], 2)
posedirs = fm["posedirs"].reshape(-1, 36).T
parents = fm["kintree_table"][0].clone(); parents[0] = -1
betas = torch.cat([shape, exp], dim=1)
full_pose = torch.cat([pose[:, :3], torch.zeros(T, 3), pose[:, 3:], torch.zeros(T, 6)], dim=1)
verts, _ = ref.lbs(betas, full_pose, fm["v_template"][None].expand(T, -1, -1), shapedirs, posedirs, fm["J_regressor"], parents,
                   fm["weights"], dtype=torch.float32, detach_pose_correctives=False)
out = os.path.join(REPO, "tests", "golden", "flame_lbs.npz")
np.savez_compressed(out, shape=shape.numpy(), motion=motion.numpy(), verts_sub=verts[:, ::37].numpy().astype(np.float32),
                    verts_sum=np.float64(verts.double().sum().item()), verts_abs_mean=np.float64(verts.abs().double().mean().item()))
print("wrote", out, os.path.getsize(out) // 1024, "KiB", verts.shape, float(verts.abs().max()))
