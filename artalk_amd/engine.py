"""Host-side mirror of the reference's ``ARTAvatarInferEngine`` (``inference.py:18-95``) for the audio->motion path.

Same constructor arguments, attributes and methods as the reference class, so code written against it (the CLI
at ``inference.py:225-237``, the Gradio handler at ``:99-125``) keeps working for everything up to the FLAME
codes.  Rendering (``inference.py:59-87``: FLAME mesh / GAGAvatar, PyAV muxing) is downstream of the drop-in
boundary and out of scope: ``rendering`` only forwards to a renderer the caller plugs in.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from typing import List, Optional, Sequence

import torch

from . import capi
from .config import ARTalkConfig
from .model import BitwiseARModel


class ARTAvatarInferEngine:
    def __init__(self, load_gaga=False, fix_pose=False, clip_length=750, device="cuda",
                 ckpt_path="./assets/ARTalk_wav2vec.pt", config_path=None, state_dict=None, config=None, model=None):
        self.device = device
        self.fix_pose = fix_pose
        self.clip_length = clip_length
        audio_encoder = "wav2vec"                                           # inference.py:23
        if model is not None:          # an already loaded BitwiseARModel (several engines may share the 2 GB of weights)
            state_dict, config = {}, model.cfg
        if state_dict is None:
            # FileNotFoundError for a missing checkpoint, like torch.load at inference.py:24
            state_dict = torch.load(ckpt_path, map_location="cpu", weights_only=True)
        if config is None:
            if config_path is not None:
                configs = json.load(open(config_path))
            else:
                configs = ARTalkConfig.full().reference_dict()            # == assets/config.json
            configs["AR_CONFIG"]["AUDIO_ENCODER"] = audio_encoder
            config = ARTalkConfig.from_reference_dict(configs)
        if model is not None:
            self.ARTalk = model
        else:
            self.ARTalk = BitwiseARModel(config).eval().to(device)
            self.ARTalk.load_state_dict(state_dict, strict=True)
        # renderer-side attributes of the reference object; filled by whoever owns the (out of scope) renderers
        self.flame_model = None
        self.mesh_renderer = None
        self.output_dir = "render_results/ARTAvatar_{}".format(audio_encoder)
        self.style_motion = None
        if load_gaga:
            raise NotImplementedError("GAGAvatar rendering is outside the audio->motion path (SURVEY.md section 2, #14)")

    def set_style_motion(self, style_motion):
        if isinstance(style_motion, str):
            style_motion = torch.load("assets/style_motion/{}.pt".format(style_motion), map_location="cpu", weights_only=True)
        assert style_motion.shape == (50, 106), f"Invalid style_motion shape: {style_motion.shape}."
        self.style_motion = style_motion[None].to(self.device)

    # ------------------------------------------------------------------ inference.py:47-57
    def inference(self, audio, clip_length=None):
        audio_batch = {"audio": audio[None].to(self.device), "style_motion": self.style_motion}
        pred_motions = self.ARTalk.inference(audio_batch, with_gtmotion=False)[0]
        return self._postprocess(pred_motions, clip_length)

    def inference_batch(self, audios: Sequence[torch.Tensor], style_motions: Optional[Sequence[Optional[torch.Tensor]]] = None,
                        clip_length=None) -> List[torch.Tensor]:
        """Data-parallel form of ``inference``: B clips in one pass, each post-processed like a single clip."""
        if style_motions is None and self.style_motion is not None:
            style_motions = [self.style_motion[0]] * len(audios)
        preds = self.ARTalk.inference_batch(audios, style_motions)
        return [self._postprocess(p, clip_length) for p in preds]

    def _postprocess(self, pred_motions, clip_length=None):
        clip_length = clip_length if clip_length is not None else self.clip_length
        pred_motions = self.smooth_motion_savgol_device(pred_motions)[:clip_length]
        if self.fix_pose:
            pred_motions[..., 100:103] *= 0.0
        pred_motions[..., 104:] *= 0.0
        return pred_motions

    def smooth_motion_savgol(self, motion_codes):
        """``smooth_motion_savgol`` (inference.py:89-95) without the host round trip: Savitzky-Golay (5, 2) on all dims,
        (9, 3) on dims 100:103 of the unsmoothed signal, scipy's mode='interp' edges, as a device kernel (artalk_savgol)."""
        T = motion_codes.shape[0]
        if T < 9:
            # scipy raises for mode='interp' when window_length (9 for the pose dims) exceeds the signal length
            raise ValueError("If mode is 'interp', window_length must be less than or equal to the size of x.")
        x = motion_codes.contiguous()
        out = torch.empty_like(x)
        with torch.cuda.device(x.device):
            rc = capi.lib().artalk_savgol(self.ARTalk._h, capi.ptr(x), capi.ptr(out), int(T), capi.current_stream_ptr())
        if rc != capi.OK:
            raise RuntimeError("artalk_savgol failed: " + self.ARTalk._err())
        return out

    smooth_motion_savgol_device = smooth_motion_savgol

    def rendering(self, audio, pred_motions, shape_id="mesh", shape_code=None, save_name="ARTAvatar.mp4"):
        """Downstream of the boundary (inference.py:59-87).  Produces the vertices the reference's mesh branch would
        feed its renderer when a FLAME model has been plugged in; rasterising and muxing are not part of this package."""
        if shape_id != "mesh" or self.flame_model is None:
            raise NotImplementedError("rendering is outside the audio->motion path; plug a FLAME model into "
                                      "engine.flame_model to get vertices, or pass pred_motions to the reference renderer")
        if shape_code is None:
            shape_code = audio.new_zeros(1, 300).to(self.device).expand(pred_motions.shape[0], -1)
        else:
            assert shape_code.dim() == 2, f"Invalid shape_code dim: {shape_code.dim()}."
            assert shape_code.shape[0] == 1, f"Invalid shape_code shape: {shape_code.shape}."
            shape_code = shape_code.to(self.device).expand(pred_motions.shape[0], -1)
        return self.ARTalk.basic_vae.get_flame_verts(self.flame_model, shape_code, pred_motions, with_global=True)
