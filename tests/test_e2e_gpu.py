"""End-to-end parity of the HIP path (through the C ABI and the Python mirror of the reference interface) against
the golden vectors produced by the reference itself (oracle/make_golden.py), on the same seeded inputs.

Bar (BASELINE.json north_star): FLAME codes within 1e-3 max-abs; bit decisions exact.  fp32 summation order differs
from the reference's MKL kernels, so a decision may legitimately differ only where the reference's own margin is at
rounding level (< TAU); every decision after such a flip depends on it, so codes are compared up to that chunk.
"""
import numpy as np
import pytest
import torch

from conftest import get_gpu_model, get_state_dict, golden_inputs, load_golden

pytestmark = pytest.mark.gpu

TAU_LOGIT = 2e-4     # |l0 - l1| of the reference at a legitimately flipped AR bit
TAU_HIST = 2e-5      # |z| (unit-normalised) of the reference at a legitimately flipped history bit
FLAME_TOL = 1e-3
LEVEL_OF = np.concatenate([np.full(p, i) for i, p in enumerate((1, 5, 25, 50, 100))])

CASES = ["tiny_4s_s0", "tiny_10s_s1_style", "tiny_6p3s_s2", "full_10s_s0", "full_10s_s1_style", "full_4s_s2", "full_5p5s_s3_style",
         "full_demo_eng1", "full_demo_eng2"]     # the last two: real speech (reference demo/*.wav), 4 chunks / 1 chunk


def first_flip(mine, gold, margin, tau):
    """mine/gold: (chunks,181,32) 0/1.  Returns (chunk, level) of the first differing decision group or None;
    asserts that every differing decision in that first group has a reference margin below tau."""
    for c in range(gold.shape[0]):
        for lv in range(5):
            sel = LEVEL_OF == lv
            d = mine[c, sel] != gold[c, sel]
            if d.any():
                mg = margin[c, sel][d].astype(np.float64)
                assert (mg < tau).all(), f"decision differs at chunk {c} level {lv} with reference margin {mg.max():.3e} >= {tau}"
                return c, lv
    return None


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
@pytest.mark.parametrize("case", CASES)
def test_against_reference_golden(case, precision):
    """precision 'f32': exact fp32 MFMA GEMMs; 'f16x3': operand-split fp16 MFMA GEMMs (fp32-class accuracy).  Same bar."""
    g = load_golden(case)
    name = case.split("_")[0]
    m = get_gpu_model(name)
    m.set_precision(precision)
    cfg, sd = get_state_dict(name)
    audio, style = golden_inputs(g, sd)
    out = m.inference_batch([audio], [style], return_aux=True)[0].cpu().numpy()
    aux = m.last_aux
    assert out.shape == g["out"].shape
    # wav2vec2 features (third-party arithmetic pinned by the golden slice)
    w2v = aux["w2v"].cpu().numpy()
    assert np.abs(w2v[:, :, :16] - g["w2v_slice"]).max() < 2e-3
    assert abs(np.abs(w2v).mean() - float(g["w2v_abs_mean"])) < 1e-4
    bits = aux["bits"][0].cpu().numpy()
    hist = aux["hist_bits"][0].cpu().numpy()
    gbits = np.unpackbits(g["bits"], axis=-1)
    ghist = np.unpackbits(g["hist_bits"], axis=-1)
    n_chunks = gbits.shape[0]
    # causal order of decisions: hist[0], bits[0], hist[1], bits[1], ...
    good_chunks = n_chunks
    for c in range(n_chunks):
        fh = first_flip(hist[c:c + 1], ghist[c:c + 1], g["hist_margin"][c:c + 1], TAU_HIST)
        if fh is not None:
            good_chunks = c
            break
        fb = first_flip(bits[c:c + 1], gbits[c:c + 1], g["logit_margin"][c:c + 1], TAU_LOGIT)
        if fb is not None:
            good_chunks = c
            break
    n = min(good_chunks * 100, out.shape[0])
    err = np.abs(out[:n] - g["out"][:n]).max() if n else 0.0
    m.set_precision("f32")
    print(f"{case} [{precision}]: chunks exact {good_chunks}/{n_chunks}, FLAME max-abs err {err:.3e}, "
          f"w2v err {np.abs(w2v[:, :, :16] - g['w2v_slice']).max():.3e}")
    assert err < FLAME_TOL
    # at least the first chunk must be decision-exact in every fixture (margins there are far above rounding)
    assert good_chunks >= 1


def test_batch_equals_single_runs():
    """Batch > 1 is defined as B independent batch-1 runs (the reference asserts B == 1, app/models.py:65): a ragged
    batch (different lengths, with/without style) must reproduce the single-clip results.  Tile shapes and split-K
    factors depend on the row count, so the fp32 summation order (not the arithmetic) differs between batch sizes,
    exactly as MKL's blocking does for the reference: equality is to rounding (1e-5), decisions identical."""
    from artalk_amd.synth import synth_audio, synth_style
    m = get_gpu_model("tiny")
    cfg, sd = get_state_dict("tiny")
    mean, std = sd["basic_vae.motion_mean"].numpy(), sd["basic_vae.motion_std"].numpy()
    secs = [4.0, 10.0, 6.3, 1.7, 8.0]
    audios = [torch.from_numpy(synth_audio(10 + i, s)) for i, s in enumerate(secs)]
    styles = [None, torch.from_numpy(synth_style(11, mean, std)), None, torch.from_numpy(synth_style(13, mean, std)), None]
    for precision in ("f32", "f16x3"):
        m.set_precision(precision)
        batch = m.inference_batch(audios, styles, return_aux=True)
        bbits = [b.clone() for b in m.last_aux["bits"]]
        for i in range(len(secs)):
            single = m.inference_batch([audios[i]], [styles[i]], return_aux=True)[0]
            assert batch[i].shape == single.shape == (m.seq_length(audios[i].shape[0]), 106)
            assert torch.equal(bbits[i], m.last_aux["bits"][0]), f"clip {i}: decisions differ between batch and single run"
            assert (batch[i] - single).abs().max().item() < 1e-5, f"clip {i}: batch result differs from single run"
        again = m.inference_batch(audios, styles)
        assert all(torch.equal(a, b) for a, b in zip(batch, again)), "same batch twice must be bit-identical (deterministic split-K)"
    m.set_precision("f32")


def test_overlapped_schedule_equals_sequential():
    """With artalk_set_overlap(1), from 8 clips up artalk_infer runs wav2vec2 chunk index by chunk index on its own stream beside
    the AR/VAE body of the previous index.  Same arithmetic, different grouping of the wav2vec2 rows: the ragged 9-clip batch must
    give the same decisions and the same codes to rounding as the sequential schedule, and must be repeatable bit for bit
    (a race between the two streams would show up here)."""
    from artalk_amd.synth import synth_audio, synth_style
    m = get_gpu_model("tiny")
    cfg, sd = get_state_dict("tiny")
    mean, std = sd["basic_vae.motion_mean"].numpy(), sd["basic_vae.motion_std"].numpy()
    secs = [12.0, 10.0, 9.1, 8.0, 6.3, 4.0, 4.0, 2.2, 1.7]
    audios = [torch.from_numpy(synth_audio(40 + i, s)) for i, s in enumerate(secs)]
    styles = [torch.from_numpy(synth_style(41 + i, mean, std)) if i % 3 == 0 else None for i in range(len(secs))]
    try:
        for precision in ("f16x3", "f32"):
            m.set_precision(precision)
            m.set_overlap(False)
            seq = m.inference_batch(audios, styles, return_aux=True)
            sbits = [b.clone() for b in m.last_aux["bits"]]
            sw2v = m.last_aux["w2v"].clone()
            m.set_overlap(True)
            ovl = m.inference_batch(audios, styles, return_aux=True)
            assert (m.last_aux["w2v"] - sw2v).abs().max().item() < 1e-4 * max(1.0, sw2v.abs().max().item())
            for i in range(len(secs)):
                assert torch.equal(sbits[i], m.last_aux["bits"][i]), f"clip {i}: decisions differ between the schedules"
                assert (seq[i] - ovl[i]).abs().max().item() < 1e-5
            for _ in range(3):
                again = m.inference_batch(audios, styles)
                assert all(torch.equal(a, b) for a, b in zip(ovl, again)), "overlapped schedule is not repeatable"
    finally:
        m.set_overlap(False)
        m.set_precision("f32")


def test_streaming_equals_batch_call():
    """Chunk-at-a-time streaming (history kept in the model) reproduces the one-shot call: same decisions, codes to 1e-5
    (the wav2vec2 GEMMs see 1 chunk instead of 3 per launch, so only the fp32 summation order can differ)."""
    from artalk_amd.synth import synth_audio, synth_style
    m = get_gpu_model("tiny")
    cfg, sd = get_state_dict("tiny")
    mean, std = sd["basic_vae.motion_mean"].numpy(), sd["basic_vae.motion_std"].numpy()
    audios = [torch.from_numpy(synth_audio(20 + i, 10.0)) for i in range(2)]
    styles = [None, torch.from_numpy(synth_style(21, mean, std))]
    want = m.inference_batch(audios, styles)
    m.stream_begin(2, styles)
    got = []
    for j in range(3):
        chunk = torch.zeros(2, 64000)
        for b in range(2):
            seg = audios[b][j * 64000:(j + 1) * 64000]
            chunk[b, :seg.shape[0]] = seg
        got.append(m.stream_chunk(chunk.cuda()))
    got = torch.cat(got, dim=1)[:, :250]
    for b in range(2):
        assert (got[b] - want[b]).abs().max().item() < 1e-5
    with pytest.raises(AssertionError):
        m.stream_chunk(torch.zeros(3, 64000))


def test_f16x3_overflow_falls_back_to_f32():
    """An activation beyond fp16's range makes the f16x3 result non-finite; the host re-runs the call in f32 mode instead of
    returning NaNs.  Forced here with weights scaled so that a GEMM input exceeds 65504."""
    import warnings
    from artalk_amd.model import BitwiseARModel
    from artalk_amd.synth import synth_audio
    cfg, sd = get_state_dict("tiny")
    big = dict(sd)
    k = "audio_encoder.feature_projection.layer_norm.weight"       # LN output feeds the projection GEMM: 1e6 > 65504
    big[k] = sd[k] * 1e6
    kk = "audio_encoder.feature_projection.projection.weight"
    big[kk] = sd[kk] * 1e-6                                          # keep the function the same in exact arithmetic
    m = BitwiseARModel(cfg).eval().to("cuda")
    m.load_state_dict(big, strict=True)
    audio = torch.from_numpy(synth_audio(3, 4.0))
    m.set_precision("f32")
    want = m.inference_batch([audio])[0]
    m.set_precision("f16x3")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        got = m.inference_batch([audio])[0]
    assert any("re-running" in str(x.message) for x in w)
    assert torch.isfinite(got).all() and torch.equal(got, want)
    assert m._precision == "f16x3"


def test_reference_call_surface():
    """BitwiseARModel.inference(batch) / ARTAvatarInferEngine.inference(audio) keep the reference's surface."""
    from artalk_amd.engine import ARTAvatarInferEngine
    from artalk_amd.synth import synth_audio
    cfg, sd = get_state_dict("tiny")
    eng = ARTAvatarInferEngine(load_gaga=False, fix_pose=False, clip_length=60, device="cuda", state_dict=sd, config=cfg)
    audio = torch.from_numpy(synth_audio(0, 4.0))
    pred = eng.inference(audio)
    assert pred.shape == (60, 106) and pred.is_cuda
    assert float(pred[:, 104:].abs().max()) == 0.0
    g = load_golden("tiny_4s_s0")
    assert np.abs(pred.cpu().numpy() - g["engine_out"][:60]).max() < FLAME_TOL
    with pytest.raises(AssertionError):
        eng.ARTalk.inference({"audio": torch.zeros(2, 16000), "style_motion": None})
    with pytest.raises(AssertionError):
        eng.set_style_motion(torch.zeros(49, 106))
    with pytest.raises(FileNotFoundError):
        eng.set_style_motion("no_such_style")


def test_savgol_device_matches_scipy():
    from artalk_amd.engine import ARTAvatarInferEngine
    cfg, sd = get_state_dict("tiny")
    m = get_gpu_model("tiny")
    eng = ARTAvatarInferEngine.__new__(ARTAvatarInferEngine)
    eng.ARTalk = m
    x = torch.randn(137, 106, generator=torch.Generator().manual_seed(1))
    ref = ARTAvatarInferEngine.smooth_motion_savgol(x)
    out = eng.smooth_motion_savgol_device(x.cuda()).cpu()
    assert (out - ref).abs().max().item() < 2e-6
    with pytest.raises(ValueError):
        eng.smooth_motion_savgol_device(torch.zeros(8, 106).cuda())
