"""Audio front-end of the reference CLI (``inference.py:230-231``):

    audio, sr = torchaudio.load(path); audio = torchaudio.transforms.Resample(sr, 16000)(audio).mean(dim=0)

torchaudio is not installed here, so this module restates what those two calls do: PCM WAV decode to float32 in
[-1, 1) (``torchaudio.load`` normalises integer PCM by 2^(bits-1)) and ``torchaudio.functional.resample`` with its
defaults (``sinc_interp_hann``, ``lowpass_filter_width=6``, ``rolloff=0.99``): a polyphase FIR built from a
Hann-windowed sinc.  The FIR runs on the GPU (``artalk_op_resample_mean``); the filter taps are built on the host
in float64 and cast to float32 exactly as torchaudio does for float32 input.

PARITY UNPINNED for this stage: no torchaudio, no reference fixture of resampled audio exists offline; the CPU
restatement in ``oracle/audio_oracle.py`` follows the same published algorithm (torchaudio 2.4.1 pin in the reference's
``environment.yml``) and the tests check the kernel against it plus filter properties.
"""
from __future__ import annotations

import ctypes as C
import math
import wave

import numpy as np
import torch

from . import capi


def read_wav(path: str):
    """PCM WAV -> (float32 tensor (channels, frames) in [-1, 1), sample_rate), like ``torchaudio.load(path)``."""
    with wave.open(path, "rb") as w:
        nch, sw, sr, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        raw = w.readframes(n)
    if sw == 2:
        a = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif sw == 4:
        a = (np.frombuffer(raw, dtype="<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
    elif sw == 3:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v >= 1 << 23, v - (1 << 24), v)
        a = (v.astype(np.float64) / 8388608.0).astype(np.float32)
    elif sw == 1:
        a = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    else:
        raise ValueError(f"unsupported sample width {sw}")
    return torch.from_numpy(a.reshape(-1, nch).T.copy()), sr


def sinc_resample_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """torchaudio ``_get_sinc_resample_kernel`` (sinc_interp_hann): returns (taps [new, 2*width+orig] f32, width, orig, new)
    with orig/new already divided by their gcd."""
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
    t *= base
    t = t.clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t *= math.pi
    scale = base / orig
    kernels = torch.where(t == 0, torch.tensor(1.0, dtype=torch.float64), t.sin() / t)
    kernels *= window * scale
    return kernels.to(torch.float32).reshape(new, -1).contiguous(), width, orig, new


def resample_mean_16k(waveform: torch.Tensor, sr: int, device="cuda", target: int = 16000) -> torch.Tensor:
    """``Resample(sr, 16000)(waveform).mean(dim=0)`` on the GPU: (channels, N) -> (ceil(N*16000/sr),) float32."""
    assert waveform.dim() == 2
    x = waveform.to(device=device, dtype=torch.float32).contiguous()
    nch, n = x.shape
    if sr == target:
        return x.mean(dim=0)
    taps, width, orig, new = sinc_resample_kernel(sr, target)
    n_out = int(math.ceil(new * n / orig))
    out = torch.empty(n_out, dtype=torch.float32, device=x.device)
    tp = taps.to(x.device)
    with torch.cuda.device(x.device):
        rc = capi.lib().artalk_op_resample_mean(capi.ptr(x), nch, n, capi.ptr(tp), orig, new, width, capi.ptr(out), n_out,
                                                capi.current_stream_ptr())
    if rc != capi.OK:
        raise RuntimeError(f"artalk_op_resample_mean failed ({rc})")
    return out


def load_audio_16k(path: str, device="cuda") -> torch.Tensor:
    """The two lines of ``inference.py:230-231`` for a PCM WAV file."""
    wav, sr = read_wav(path)
    return resample_mean_16k(wav, sr, device)
