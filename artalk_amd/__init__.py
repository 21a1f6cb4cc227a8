"""artalk_amd: MI355X-native (gfx950) implementation of ARTalk's audio->motion path.

Only the hot path named by BASELINE.json's north_star lives here: wav2vec2 (XLS-R-300M) ->
BitwiseARModel multi-scale AR decode -> BITWISE_VAE decode -> 106-d FLAME codes, as hand-written
HIP kernels behind a C-ABI library (``include/artalk_hip.h``), driven from a Python engine that
keeps the reference's ``ARTAvatarInferEngine`` / ``BitwiseARModel`` call surface.
"""
from .config import ARTalkConfig  # noqa: F401

__all__ = ["ARTalkConfig"]
