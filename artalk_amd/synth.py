"""Synthetic inputs shared by the golden generator, the tests and ``bench.py``."""
from __future__ import annotations

import numpy as np


def synth_audio(seed: int, seconds: float = 10.0, sr: int = 16000) -> np.ndarray:
    """``0.1 * N(0,1)`` float32 mono at 16 kHz, seed = clip index."""
    n = int(round(seconds * sr))
    return (np.random.default_rng(seed).standard_normal(n).astype(np.float32) * np.float32(0.1))


def synth_style(seed: int, motion_mean: np.ndarray, motion_std: np.ndarray) -> np.ndarray:
    """A (50,106) style clip ``mean + std * N(0,1)`` (real ``assets/style_motion/*.pt`` are not available offline)."""
    z = np.random.default_rng(1000 + seed).standard_normal((50, motion_mean.shape[0])).astype(np.float32)
    return (np.asarray(motion_mean, np.float32) + np.asarray(motion_std, np.float32) * z).astype(np.float32)
