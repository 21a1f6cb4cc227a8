"""Model configuration for the audio->motion path.

Mirrors what the reference reads as data: ``assets/config.json`` (reference
``inference.py:25-26``) plus the XLS-R-300M hyper-parameters the reference fetches by
name (``app/models.py:25``) and the constants it hard-codes (768 / 1024 / 128,
``app/models.py:19-27``; style encoder sizes, ``app/modules/style_encoder.py:15-21``).
"""
from __future__ import annotations

import copy
import json
import os
from dataclasses import dataclass, field

_ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")


def _load(name):
    with open(os.path.join(_ASSETS, name)) as f:
        d = json.load(f)
    d.pop("_comment", None)
    return d


@dataclass
class ARTalkConfig:
    # AR_CONFIG
    ar_depth: int = 12
    ar_heads: int = 12
    prev_ratio: int = 1
    audio_encoder: str = "wav2vec"
    # VAE_CONFIG
    motion_dim: int = 106
    code_dim: int = 32
    vae_depth: int = 8
    vae_heads: int = 8
    vae_hidden: int = 512
    patch_nums: tuple = (1, 5, 25, 50, 100)
    # hard-coded in the reference
    embed_dim: int = 768          # app/models.py:19
    cond_dim: int = 1024          # app/models.py:27
    style_dim: int = 128          # app/modules/style_encoder.py:16
    style_heads: int = 4
    style_layers: int = 4
    style_ffn: int = 512
    style_len: int = 50           # inference.py:44
    style_pe_len: int = 600       # style_encoder.py:45
    # wav2vec2 (XLS-R-300M unless overridden)
    w2v: dict = field(default_factory=lambda: _load("xlsr_300m.json"))

    # ---- derived sizes ----
    @property
    def n_tokens(self):           # 181
        return sum(self.patch_nums)

    @property
    def frames_per_chunk(self):   # 100
        return self.patch_nums[-1]

    @property
    def samples_per_chunk(self):  # 64000 (app/models.py:80)
        return int(self.patch_nums[-1] / 25.0 * 16000)

    @property
    def head_dim(self):
        return self.embed_dim // self.ar_heads

    def w2v_lengths(self, n=None):
        """Conv feature-extractor lengths: 64000 -> 12799 ... -> 199."""
        n = self.samples_per_chunk if n is None else n
        out = []
        for k, s in zip(self.w2v["conv_kernel"], self.w2v["conv_stride"]):
            n = (n - k) // s + 1
            out.append(n)
        return out

    def reference_dict(self):
        """The dict the reference's ``BitwiseARModel(configs)`` takes (inference.py:25-27)."""
        return {
            "AR_CONFIG": {"T_DEPTH": self.ar_depth, "T_NUM_HEADS": self.ar_heads,
                          "PREV_RATIO": self.prev_ratio, "AUDIO_ENCODER": self.audio_encoder},
            "VAE_CONFIG": {"MOTION_DIM": self.motion_dim, "V_CODE_DIM": self.code_dim,
                           "T_DEPTH": self.vae_depth, "T_NUM_HEADS": self.vae_heads,
                           "T_HIDDEN_DIM": self.vae_hidden, "V_PATCH_NUMS": list(self.patch_nums)},
        }

    @staticmethod
    def from_reference_dict(d, w2v=None):
        ar, vae = d["AR_CONFIG"], d["VAE_CONFIG"]
        enc = ar.get("AUDIO_ENCODER", "wav2vec")
        if enc != "wav2vec":
            # same error type as app/models.py:32 (mimi has no released checkpoint; out of scope)
            raise ValueError("Invalid audio encoder: {}".format(enc))
        assert ar["PREV_RATIO"] == 1, "only PREV_RATIO=1 (assets/config.json:5) is supported"
        c = ARTalkConfig(ar_depth=ar["T_DEPTH"], ar_heads=ar["T_NUM_HEADS"], prev_ratio=ar["PREV_RATIO"],
                         motion_dim=vae["MOTION_DIM"], code_dim=vae["V_CODE_DIM"], vae_depth=vae["T_DEPTH"],
                         vae_heads=vae["T_NUM_HEADS"], vae_hidden=vae["T_HIDDEN_DIM"],
                         patch_nums=tuple(vae["V_PATCH_NUMS"]))
        if w2v is not None:
            c.w2v = copy.deepcopy(w2v)
        return c

    @staticmethod
    def full():
        return ARTalkConfig.from_reference_dict(_load("config.json"))

    @staticmethod
    def tiny():
        """Same widths, fewer layers: for fast CPU tests of the oracle against the reference."""
        c = ARTalkConfig.full()
        c.ar_depth, c.vae_depth = 2, 2
        c.w2v = copy.deepcopy(c.w2v)
        c.w2v["num_hidden_layers"] = 2
        return c

    @staticmethod
    def by_name(name):
        return {"full": ARTalkConfig.full, "tiny": ARTalkConfig.tiny}[name]()
