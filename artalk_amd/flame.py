"""Host-side mirror of the reference's ``FLAMEModel`` (``app/flame_model/FLAME.py:15-142``) for the vertex path the mesh
renderer uses (``inference.py:69`` -> ``BITWISE_VAE.get_flame_verts`` -> ``FLAMEModel.forward`` with ``no_lmks=True``).

The reference constructor reads the licence-gated ``assets/FLAME_with_eye.pt`` (``build_resources.sh:1-11``), which is not
available offline; this class takes the same dictionary (or a path to it) and runs linear blend skinning on the GPU through
``artalk_flame_*`` (``include/artalk_hip.h``).  Landmarks (``no_lmks=False``) are renderer/tracker features and not built.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import capi


def synthetic_flame_asset(seed: int = 7, n_verts: int = 5023):
    """A deterministic stand-in with the real asset's keys and shapes (V=5023, 400 shape+expression directions after the
    reference's slicing, 36 pose directions, 5 joints) so the skinning path can be tested without the licensed file."""
    g = np.random.default_rng(seed)
    V = n_verts
    jr = np.abs(g.standard_normal((5, V)).astype(np.float32)) ** 4
    jr /= jr.sum(axis=1, keepdims=True)
    w = np.exp(2.0 * g.standard_normal((V, 5)).astype(np.float32))
    w /= w.sum(axis=1, keepdims=True)
    model = {
        "f": torch.from_numpy(g.integers(0, V, size=(9976, 3)).astype(np.int64)),
        "v_template": torch.from_numpy((0.1 * g.standard_normal((V, 3))).astype(np.float32)),
        "shapedirs": torch.from_numpy((0.01 * g.standard_normal((V, 3, 400))).astype(np.float32)),
        "posedirs": torch.from_numpy((0.01 * g.standard_normal((V, 3, 36))).astype(np.float32)),
        "J_regressor": torch.from_numpy(jr.astype(np.float32)),
        "kintree_table": torch.tensor([[-1, 0, 1, 1, 1], [0, 1, 2, 3, 4]], dtype=torch.int64),
        "weights": torch.from_numpy(w.astype(np.float32)),
    }
    return {"flame_model": model}


class FLAMEModel:
    def __init__(self, n_shape, n_exp, scale=1.0, no_lmks=False, lmks_type="lmks70", flame_ckpt=None, device="cuda"):
        if isinstance(flame_ckpt, str):
            flame_ckpt = torch.load(flame_ckpt, map_location="cpu", weights_only=True)     # FileNotFoundError like FLAME.py:26
        if flame_ckpt is None:
            flame_ckpt = torch.load("assets/FLAME_with_eye.pt", map_location="cpu", weights_only=True)
        if not no_lmks:
            raise NotImplementedError("landmark embeddings are renderer/tracker features; construct with no_lmks=True")
        self.scale, self.no_lmks, self.lmks_type = scale, no_lmks, lmks_type
        fm = flame_ckpt["flame_model"]
        self.device = torch.device(device)
        self.faces_tensor = fm["f"]
        self.v_template = fm["v_template"].float().contiguous()
        sd = fm["shapedirs"].float()
        self.shapedirs = torch.cat([sd[:, :, :n_shape], sd[:, :, 300:300 + n_exp]], 2).contiguous()      # FLAME.py:38
        pd = fm["posedirs"].float()
        self.posedirs = pd.reshape(-1, pd.shape[-1]).T.contiguous()                                        # FLAME.py:39-40: [36, V*3]
        self.J_regressor = fm["J_regressor"].float().contiguous()
        parents = fm["kintree_table"][0].clone()
        parents[0] = -1                                                                                     # FLAME.py:42-43
        self.parents = parents
        self.lbs_weights = fm["weights"].float().contiguous()
        self.eye_pose = torch.zeros(1, 6)
        self.neck_pose = torch.zeros(1, 3)
        V, NB = self.v_template.shape[0], self.shapedirs.shape[2]
        self.n_verts, self.n_betas = V, NB
        pt = self.posedirs.T.contiguous()                       # [V*3][36] = GEMM weight layout
        par32 = parents.to(torch.int32).contiguous()
        h = C.c_void_p()
        L = capi.lib()
        rc = L.artalk_flame_create(self.device.index or 0, V, NB, pt.shape[1], capi.ptr(self.v_template), capi.ptr(self.shapedirs), capi.ptr(pt),
                                   capi.ptr(self.J_regressor), capi.ptr(par32), capi.ptr(self.lbs_weights), float(scale), C.byref(h))
        if rc != capi.OK:
            raise RuntimeError("artalk_flame_create failed: " + L.artalk_flame_last_error(None).decode())
        self._h = h

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None:
                capi.lib().artalk_flame_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def to(self, device):
        return self

    def get_faces(self):
        return self.faces_tensor.long()

    def __call__(self, *a, **k):
        return self.forward(*a, **k)

    @torch.no_grad()
    def forward(self, shape_params=None, expression_params=None, pose_params=None, eye_pose_params=None, verts_sclae=None):
        """FLAME.py:117-142: (N,n_shape), (N,n_exp), (N,6 | 3) -> vertices (N,V,3) * scale."""
        dev = self.device
        n = shape_params.shape[0]
        if pose_params is None:
            pose_params = self.eye_pose.expand(n, -1)
        if eye_pose_params is None:
            eye_pose_params = self.eye_pose.expand(n, -1)
        if expression_params is None:
            expression_params = torch.zeros(n, self.n_betas - shape_params.shape[1])
        shape_params, expression_params = shape_params.to(dev).float(), expression_params.to(dev).float()
        pose_params, eye_pose_params = pose_params.to(dev).float(), eye_pose_params.to(dev).float()
        if pose_params.shape[-1] == 3:
            pose_params = torch.cat([torch.zeros(n, 3, device=dev), pose_params], dim=-1)
        betas = torch.cat([shape_params, expression_params], dim=1).contiguous()
        full_pose = torch.cat([pose_params[:, :3], self.neck_pose.to(dev).expand(n, -1), pose_params[:, 3:], eye_pose_params], dim=1).contiguous()
        assert betas.shape[1] == self.n_betas and full_pose.shape[1] == 15
        out = torch.empty(n, self.n_verts, 3, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = capi.lib().artalk_flame_verts(self._h, capi.ptr(betas), capi.ptr(full_pose), n, capi.ptr(out), capi.current_stream_ptr())
        if rc != capi.OK:
            raise RuntimeError("artalk_flame_verts failed: " + capi.lib().artalk_flame_last_error(self._h).decode())
        if verts_sclae is not None:
            return out * (verts_sclae / self.scale)
        return out
