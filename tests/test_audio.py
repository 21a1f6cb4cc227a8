"""Audio front-end (reference inference.py:230-231): WAV decode + torchaudio-style sinc resampler + channel mean.
PARITY UNPINNED against torchaudio itself (absent here); see oracle/audio_oracle.py."""
import math
import os
import wave

import numpy as np
import pytest
import torch

from artalk_amd.audio import read_wav, sinc_resample_kernel
from audio_oracle import get_sinc_resample_kernel, load_mono_16k, resample
from conftest import GOLDEN

DEMO_WAV = "/root/reference/demo/eng1.wav"


def _write_wav(path, data_i16, sr):
    with wave.open(str(path), "wb") as w:
        w.setnchannels(data_i16.shape[1]); w.setsampwidth(2); w.setframerate(sr)
        w.writeframes(data_i16.astype("<i2").tobytes())


def test_read_wav_matches_pcm(tmp_path):
    rng = np.random.default_rng(0)
    pcm = rng.integers(-32768, 32767, size=(4800, 2), dtype=np.int16)
    _write_wav(tmp_path / "a.wav", pcm, 48000)
    wav, sr = read_wav(str(tmp_path / "a.wav"))
    assert sr == 48000 and wav.shape == (2, 4800) and wav.dtype == torch.float32
    assert np.array_equal(wav.numpy(), pcm.T.astype(np.float32) / 32768.0)


def test_filter_taps_and_geometry():
    taps, width, orig, new = sinc_resample_kernel(48000, 16000)
    assert (orig, new, width) == (3, 1, math.ceil(6 * 3 / 0.99)) and taps.shape == (1, 2 * width + 3)
    k, w = get_sinc_resample_kernel(48000, 16000, 16000)
    assert w == width and torch.equal(k.reshape(1, -1), taps)
    assert abs(float(taps.sum()) - 1.0) < 2e-3                       # unit DC gain
    t2, w2, o2, n2 = sinc_resample_kernel(44100, 16000)
    assert (o2, n2) == (441, 160) and t2.shape == (160, 2 * w2 + 441)


def test_resampler_properties():
    sr, n = 48000, 48000
    t = torch.arange(n, dtype=torch.float64) / sr
    low = torch.sin(2 * math.pi * 1000 * t).float()[None]             # passband tone keeps its amplitude
    high = torch.sin(2 * math.pi * 12000 * t).float()[None]           # above the new Nyquist (8 kHz): removed
    y_low, y_high = resample(low, sr, 16000)[0], resample(high, sr, 16000)[0]
    assert y_low.shape[0] == math.ceil(n * 16000 / sr)
    mid = slice(200, -200)
    assert abs(float(y_low[mid].abs().max()) - 1.0) < 2e-2
    assert float(y_high[mid].abs().max()) < 2e-2
    stereo = torch.cat([low, 0.5 * low])
    assert torch.allclose(load_mono_16k(stereo, sr), 0.75 * y_low, atol=1e-6)
    assert torch.equal(resample(low, 16000, 16000), low)


@pytest.mark.skipif(not os.path.exists(DEMO_WAV), reason="reference demo wav only exists in the build container")
def test_demo_fixture_is_reproducible():
    """tests/golden/demo_16k_s16.npz is exactly read_wav -> resampler restatement -> int16 of the reference's demo/eng1.wav."""
    wav, sr = read_wav(DEMO_WAV)
    a = load_mono_16k(wav, sr).numpy()
    q = np.clip(np.round(a * 32768.0), -32768, 32767).astype(np.int16)
    fx = np.load(os.path.join(GOLDEN, "demo_16k_s16.npz"))["eng1"]
    assert q.shape == fx.shape == (217088,) and np.array_equal(q, fx)


def test_config1_plumbing_on_cpu_oracle():
    """BASELINE configs[0]: demo/eng1.wav, clip_length=50 through the CPU path (oracle engine restatement), reduced depth."""
    from artalk_oracle import engine_inference
    from conftest import get_oracle
    q = np.load(os.path.join(GOLDEN, "demo_16k_s16.npz"))["eng1"]
    audio = torch.from_numpy(q.astype(np.float32) / np.float32(32768.0))[: 16000 * 5]
    out = engine_inference(get_oracle("tiny"), audio, None, clip_length=50)
    assert out.shape == (50, 106) and torch.isfinite(out).all() and float(out[:, 104:].abs().max()) == 0.0


@pytest.mark.gpu
def test_resample_kernel_matches_oracle():
    from artalk_amd.audio import resample_mean_16k
    g = torch.Generator().manual_seed(0)
    for sr, n, nch in ((48000, 100001, 2), (44100, 50000, 1), (16000, 3000, 2), (22050, 40000, 2)):
        x = torch.randn(nch, n, generator=g) * 0.3
        want = load_mono_16k(x, sr)
        got = resample_mean_16k(x, sr, "cuda").cpu()
        assert got.shape == want.shape
        assert (got - want).abs().max().item() < 2e-6
