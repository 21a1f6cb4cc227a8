"""CPU restatement of the reference's audio loading lines (inference.py:230-231).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference calls ``torchaudio.transforms.Resample(sr, 16000)`` (torchaudio 2.4.1 pinned in its
environment.yml, not installed here, not vendored) and holds no fixture of resampled audio.  This file restates the
published algorithm of ``torchaudio.functional.resample`` (``_get_sinc_resample_kernel`` + ``_apply_sinc_resample_kernel``,
defaults sinc_interp_hann / lowpass_filter_width 6 / rolloff 0.99) with torch CPU ops; the HIP kernel is checked against it.
"""
import math

import torch
import torch.nn.functional as F


def get_sinc_resample_kernel(orig_freq, new_freq, gcd, lowpass_filter_width=6, rolloff=0.99):
    orig_freq, new_freq = int(orig_freq) // gcd, int(new_freq) // gcd
    base_freq = min(orig_freq, new_freq) * rolloff
    width = math.ceil(lowpass_filter_width * orig_freq / base_freq)
    idx = torch.arange(-width, width + orig_freq, dtype=torch.float64)[None, None] / orig_freq
    t = torch.arange(0, -new_freq, -1, dtype=torch.float64)[:, None, None] / new_freq + idx
    t *= base_freq
    t = t.clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t *= math.pi
    scale = base_freq / orig_freq
    kernels = torch.where(t == 0, torch.tensor(1.0, dtype=torch.float64), t.sin() / t)
    kernels *= window * scale
    return kernels.to(torch.float32), width


def resample(waveform, orig_freq, new_freq):
    """waveform (..., N) float32 -> (..., ceil(N*new/orig))."""
    if orig_freq == new_freq:
        return waveform
    gcd = math.gcd(int(orig_freq), int(new_freq))
    kernel, width = get_sinc_resample_kernel(orig_freq, new_freq, gcd)
    o, n = int(orig_freq) // gcd, int(new_freq) // gcd
    shape = waveform.size()
    w = waveform.reshape(-1, shape[-1])
    num, length = w.shape
    w = F.pad(w, (width, width + o))
    r = F.conv1d(w[:, None], kernel, stride=o)
    r = r.transpose(1, 2).reshape(num, -1)
    target = int(math.ceil(n * length / o))
    r = r[..., :target]
    return r.view(shape[:-1] + r.shape[-1:])


def load_mono_16k(waveform, sr):
    """inference.py:231: Resample(sr, 16000)(audio).mean(dim=0)"""
    return resample(waveform, sr, 16000).mean(dim=0)
