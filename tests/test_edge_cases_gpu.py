"""Edge cases of the path on the GPU against the CPU oracle (reduced-depth config so the oracle takes about a second):
clip lengths around the chunk boundary, near-silent and silent audio (the per-chunk normalisation divides by std + 1e-6,
app/modules/wav2vec.py:24-26), very short clips, large ragged batches, and the strict state_dict errors."""
import numpy as np
import pytest
import torch

from conftest import get_gpu_model, get_oracle, get_state_dict

pytestmark = pytest.mark.gpu


TAU = 2e-5     # a decision may differ from the oracle's only where the oracle's own logit margin is at rounding level


def _oracle_run(audio, style=None):
    o = get_oracle("tiny")
    rec = {}
    out = o.inference({"audio": audio[None], "style_motion": style[None] if style is not None else None}, record=rec)[0]
    return out.numpy(), torch.cat(rec["bits"]).numpy(), torch.cat(rec["logit_margin"]).numpy()


def _compare(got, bits, audio):
    """Whole-clip parity with the oracle: every chunk decision-exact and all frames within 1e-3.  Returns (max-abs err, #chunks).
    (A difference at an oracle margin below TAU would be a legitimate rounding-level flip; none of these cases has one, so it
    fails too - with the margin in the message, so that it can be told from a real defect.)"""
    want, wbits, margin = _oracle_run(audio)
    assert got.shape == want.shape
    n_chunks = wbits.shape[0]
    for c in range(n_chunks):
        d = bits[c] != wbits[c]
        assert not d.any(), (f"decision differs at chunk {c}/{n_chunks} ({int(d.sum())} bits); oracle margins there "
                             f"{np.sort(margin[c][d])[:3]} ({'rounding level' if margin[c][d].max() < TAU else 'NOT rounding level'})")
    err = float(np.abs(got - want).max())
    assert err < 1e-3, f"FLAME max-abs err {err:.3e} over {got.shape[0]} frames ({n_chunks} chunks decision-exact)"
    return err, n_chunks


def _check(audio, precision):
    m = get_gpu_model("tiny")
    m.set_precision(precision)
    got = m.inference_batch([audio], None, return_aux=True)[0].cpu().numpy()
    bits = m.last_aux["bits"][0].cpu().numpy()
    m.set_precision("f32")
    return _compare(got, bits, audio)


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
@pytest.mark.parametrize("n_samples", [640 * 9, 16000, 63999, 64000, 64001, 64640, 128000 + 1])
def test_lengths_around_chunk_boundaries(n_samples, precision):
    """seq_length = ceil(N/640), chunks = ceil(seq/100), last chunk zero padded (app/models.py:66,78-85); N = 64001 leaves a
    second chunk with a single non-zero sample (std ~ 1e-4): the harshest case for the per-chunk normalisation."""
    from artalk_amd.synth import synth_audio
    audio = torch.from_numpy(synth_audio(31, 20.0))[:n_samples].clone()
    _check(audio, precision)


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_silence_and_constant_audio(precision):
    for audio in (torch.zeros(70000), torch.full((40000,), 0.25)):
        _check(audio, precision)


def test_too_short_for_savgol_raises_like_scipy():
    from artalk_amd.engine import ARTAvatarInferEngine
    cfg, sd = get_state_dict("tiny")
    eng = ARTAvatarInferEngine.__new__(ARTAvatarInferEngine)
    eng.ARTalk, eng.device, eng.style_motion, eng.clip_length, eng.fix_pose = get_gpu_model("tiny"), "cuda", None, 750, False
    short = torch.zeros(640 * 8)                       # 8 frames < window 9
    assert eng.ARTalk.inference({"audio": short[None], "style_motion": None}).shape == (1, 8, 106)
    with pytest.raises(ValueError):
        eng.inference(short)
    with pytest.raises(ValueError):
        eng.ARTalk.inference_batch([torch.zeros(0)])
    assert eng.ARTalk.inference_batch([]) == []


def test_large_ragged_batch():
    """70 clips of 1..3 chunks in one call, as the FIRST call of a fresh model (the workspace is allocated inside the call;
    this once exposed zero-fills racing the first kernels): equals the oracle per clip."""
    from artalk_amd.synth import synth_audio
    from artalk_amd.model import BitwiseARModel
    cfg, sd = get_state_dict("tiny")
    m = BitwiseARModel(cfg).eval().to("cuda")
    m.load_state_dict(sd, strict=True)
    m.set_precision("f16x3")
    lens = [16000 + 5000 * (i % 31) for i in range(70)]
    audios = [torch.from_numpy(synth_audio(100 + i, 11.0))[:n].clone() for i, n in enumerate(lens)]
    outs = m.inference_batch(audios, return_aux=True)
    bits = [b.cpu().numpy() for b in m.last_aux["bits"]]
    m.set_precision("f32")
    assert [o.shape[0] for o in outs] == [m.seq_length(n) for n in lens]
    for i in (0, 17, 30, 69):
        _compare(outs[i].cpu().numpy(), bits[i], audios[i])


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_long_clip_30s_eight_chunks(precision):
    """The reference's default clip_length is 750 frames = 30 s = 8 chunks (inference.py:19,52): the history hand-over
    (app/models.py:111-114) runs seven times in one call.  Reduced-depth model against the oracle, every chunk decision-exact."""
    from artalk_amd.synth import synth_audio
    audio = torch.from_numpy(synth_audio(77, 30.0))
    m = get_gpu_model("tiny")
    assert m.n_chunks(audio.shape[0]) == 8 and m.seq_length(audio.shape[0]) == 750
    err, n = _check(audio, precision)
    assert n == 8
    print(f"30 s clip [{precision}]: 8 chunks decision-exact, FLAME max-abs err {err:.3e}")


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_w2v_group_split_full_width(precision):
    """More than 96 chunks in one call at the FULL model: run_wav2vec (engine.hip) then walks the chunk list in two groups (96 + 6
    here) and the second chunk index straddles the group boundary.  34 synthetic 10 s clips; seeds 0..7 against the reference's
    own outputs (tests/golden/full_cfg2_synth8.npz), every chunk decision-exact."""
    from artalk_amd.synth import synth_audio, synth_style
    from conftest import assert_clip_parity, load_clip_set
    clips = load_clip_set("full_cfg2_synth8")
    cfg, sd = get_state_dict("full")
    mean, std = sd["basic_vae.motion_mean"].numpy(), sd["basic_vae.motion_std"].numpy()
    audios = [torch.from_numpy(synth_audio(s, 10.0)) for s in range(34)]
    styles = [torch.from_numpy(synth_style(s, mean, std)) if (s % 2 == 1 and s < 8) else None for s in range(34)]
    m = get_gpu_model("full")
    assert sum(m.n_chunks(a.shape[0]) for a in audios) == 102
    m.set_precision(precision)
    try:
        outs = m.inference_batch(audios, styles, return_aux=True)
        assert m._precision == precision and m.status() == 0
        bits = [b.cpu().numpy() for b in m.last_aux["bits"]]
        hist = [h.cpu().numpy() for h in m.last_aux["hist_bits"]]
    finally:
        m.set_precision("f32")
    worst = 0.0
    for i, c in enumerate(clips):
        good, n, err = assert_clip_parity(f"group split clip {i}", precision, outs[i].cpu().numpy(), bits[i], hist[i], c["out"], c["bits"],
                                          c["hist_bits"], c["logit_margin"], c["hist_margin"], inputs=("full", audios[i], styles[i]))
        assert good == n == 3
        worst = max(worst, err)
    print(f"102 chunks in one call [{precision}]: 8 golden clips decision-exact, worst FLAME max-abs err {worst:.3e}")


def test_packed_batch_non_contiguous_view():
    """A packed (B, N) device tensor that is a strided view (every other sample of a wider buffer) must be read as the view's
    values, not as dense memory (the fast path passed only stride(0))."""
    from artalk_amd.synth import synth_audio
    m = get_gpu_model("tiny")
    m.set_precision("f32")
    clips = [torch.from_numpy(synth_audio(400 + i, 8.0)) for i in range(2)]       # 128000 samples = 2 whole chunks
    wide = torch.zeros(2, 256000)
    wide[:, ::2] = torch.stack(clips)
    wide[:, 1::2] = 7.0                                                            # junk between the samples
    view = wide.cuda()[:, ::2]
    assert not view.is_contiguous() and view.shape == (2, 128000)
    want = m.inference_batch(clips)
    got = m.inference_batch(view)
    for g, w in zip(got, want):
        assert torch.equal(g, w)


def test_bad_style_flag_is_refused_before_anything_is_enqueued():
    """artalk_infer / artalk_stream_begin validate has_style (0, 1, 2) first: EINVAL, no work queued, and a streaming session that
    is open stays open and continues exactly."""
    import ctypes as C
    from artalk_amd import capi
    from artalk_amd.synth import synth_audio
    m = get_gpu_model("tiny")
    m.set_precision("f32")
    L = capi.lib()
    audio = torch.from_numpy(synth_audio(410, 8.0))
    want = m.inference_batch([audio])[0].cpu()
    m.stream_begin(1)
    first = m.stream_chunk(audio[None, :64000].cuda()).cpu()
    # a batch call with a bad flag in the middle of the session
    pad = audio[None].cuda().contiguous()
    out = torch.zeros(1, 200, 106, device="cuda")
    style = torch.zeros(1, 50, 106, device="cuda")
    bad = (C.c_uint8 * 1)(3)
    nch = (C.c_int64 * 1)(2)
    t0 = m.last_ticket()
    rc = L.artalk_infer(m._h, capi.ptr(pad), pad.stride(0), nch, 1, capi.ptr(style), C.cast(bad, C.c_void_p), capi.ptr(out), out.stride(0),
                        None, None, None, C.c_void_p(m._stream.cuda_stream))
    assert rc == capi.EINVAL and "has_style" in m._err()
    assert m.last_ticket() == t0                       # no call was published
    second = m.stream_chunk(audio[None, 64000:].cuda()).cpu()      # the session is still there and continues bit for bit
    assert torch.equal(torch.cat([first[0], second[0]]), want)
    m.stream_end()
    assert L.artalk_stream_begin(m._h, 1, capi.ptr(style), C.cast(bad, C.c_void_p), C.c_void_p(m._stream.cuda_stream)) == capi.EINVAL


def test_batches_in_flight_and_per_call_status():
    """A serving loop that enqueues batch i+1 before it looks at batch i (``check=False`` + ``status_of(ticket)``): six different
    batches queued back to back from pinned host memory give exactly what one-at-a-time calls give, every call has its own health
    word (tickets count up by one per call), and a ticket older than the four kept slots is refused (include/artalk_hip.h)."""
    from artalk_amd.synth import synth_audio
    m = get_gpu_model("tiny")
    m.set_precision("f16x3")
    batches = [[torch.from_numpy(synth_audio(300 + 10 * k + i, 4.0 + i)).pin_memory() for i in range(3)] for k in range(6)]
    want = [[o.cpu().clone() for o in m.inference_batch(b)] for b in batches]
    t0 = m.last_ticket()
    queued, tickets = [], []
    for b in batches:
        queued.append(m.inference_batch(b, check=False))
        tickets.append(m.last_ticket())
    assert tickets == list(range(t0 + 1, t0 + 7))
    for tk in tickets[-4:]:
        assert m.status_of(tk) == 0
    with pytest.raises(RuntimeError):
        m.status_of(tickets[0])            # its slot has been reused by a later call
    torch.cuda.synchronize()
    for got, ref in zip(queued, want):
        for g, r in zip(got, ref):
            assert torch.equal(g.cpu(), r)
    assert m.status() == 0
    m.set_precision("f32")


def test_strict_state_dict_errors():
    """load_state_dict(strict=True) semantics of reference inference.py:28 through the C ABI."""
    from artalk_amd.model import BitwiseARModel
    cfg, sd = get_state_dict("tiny")
    bad = dict(sd)
    bad["not.a.key"] = torch.zeros(3)
    with pytest.raises(RuntimeError, match="Unexpected key"):
        BitwiseARModel(cfg).to("cuda").load_state_dict(bad)
    bad = dict(sd)
    del bad["logits_head.bias"]
    with pytest.raises(RuntimeError, match="Missing key"):
        BitwiseARModel(cfg).to("cuda").load_state_dict(bad)
    bad = dict(sd)
    bad["vqfeat_embed.weight"] = sd["vqfeat_embed.weight"][:, :16].contiguous()
    with pytest.raises(RuntimeError, match="size mismatch"):
        BitwiseARModel(cfg).to("cuda").load_state_dict(bad)
    bad = dict(sd)
    bad["lvl_idx"] = sd["lvl_idx"].flip(-1).contiguous()
    with pytest.raises(RuntimeError, match="lvl_idx"):
        BitwiseARModel(cfg).to("cuda").load_state_dict(bad)
    with pytest.raises(RuntimeError, match="before inference"):
        BitwiseARModel(cfg).to("cuda").inference_batch([torch.zeros(16000)])


def test_cu_partition_gives_the_same_results():
    """A model restricted to half of the compute units (artalk_set_cu_mask: its own stream, the library's side streams and the grids
    of the persistent GEMM kernels follow the mask) makes the same decisions and - the persistent kernels' tile walk changes no
    summation order - the same codes as on the whole chip."""
    from artalk_amd.model import BitwiseARModel
    from artalk_amd.synth import synth_audio
    cfg, sd = get_state_dict("tiny")
    audios = [torch.from_numpy(synth_audio(500 + i, 9.0)) for i in range(12)]       # 12 clips x 3 chunks: large-grid GEMMs + two clip groups
    m = BitwiseARModel(cfg).eval().to("cuda")
    m.load_state_dict(sd, strict=True)
    m.set_precision("f16x3")
    want = [o.cpu() for o in m.inference_batch(audios, return_aux=True)]
    wbits = [b.cpu() for b in m.last_aux["bits"]]
    m.set_cu_mask([0xFFFFFFFF] * 4 + [0] * 4)
    got = [o.cpu() for o in m.inference_batch(audios, return_aux=True)]
    gbits = [b.cpu() for b in m.last_aux["bits"]]
    assert m.status() == 0
    for a, b in zip(gbits, wbits):
        assert torch.equal(a, b)
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    m.set_cu_mask(None)
    again = [o.cpu() for o in m.inference_batch(audios)]
    for a, b in zip(again, want):
        assert torch.equal(a, b)


_FUSE_CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from conftest import get_gpu_model
from artalk_amd.synth import synth_audio
m = get_gpu_model("tiny")
audios = [torch.from_numpy(synth_audio(300 + i, s)) for i, s in enumerate((10.0, 6.3, 4.0, 9.1, 2.2))]
res = {}
for prec in ("f16x3", "f32"):
    m.set_precision(prec)
    outs = m.inference_batch(audios, None, return_aux=True)
    res[prec + "_out"] = np.concatenate([o.cpu().numpy() for o in outs])
    res[prec + "_bits"] = np.concatenate([b.cpu().numpy().reshape(-1) for b in m.last_aux["bits"]])
    one = m.inference_batch(audios[:1], None)[0]
    res[prec + "_one"] = one.cpu().numpy()
np.savez(sys.argv[2], **res)
"""


@pytest.mark.parametrize("switch", ["ARTALK_FUSE_QKV_REDUCE"])
def test_fused_qkv_reduce_is_bit_identical(tmp_path, switch):
    """With ARTALK_FUSE_QKV_REDUCE=1 the 1- / 5-token scale steps leave the split-K slabs of their q|k|v GEMM to the short-query
    attention kernel, whose lanes sum the rows they need straight into their fragment registers (engine.hip run_chunk_body,
    attention.hip AttnArgs::slabs): one launch less per block (measured: not faster, so it is off by default - DESIGN.md section 3).
    Same order of additions as the separate reduce pass, so every code and bit must be IDENTICAL between the arms (switch read at
    model creation: each arm runs in a child process), for a ragged batch of 5 and for batch 1, in both precisions.
    (The overlapped AdaLN-table schedule that shared this test in round 4 was a measured loss and is gone: profiles/r04_ada_overlap_sweep.log.)"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    arms = []
    for fuse in ("1", "0"):
        f = str(tmp_path / f"fuse{fuse}.npz")
        subprocess.run([sys.executable, "-c", _FUSE_CHILD, root, f], check=True, env=dict(os.environ, **{switch: fuse}), timeout=600)
        arms.append(np.load(f))
    for k in arms[0].files:
        assert np.isfinite(arms[0][k]).all() or arms[0][k].dtype == np.uint8
        assert np.array_equal(arms[0][k], arms[1][k]), k


def test_graph_cache_is_bounded_for_ragged_batches():
    """the body graphs are keyed by (active clips, clip group, precision, re-encoded clips).  A serving loop with varied clip
    lengths used to add a key per distinct (B_j, B_j+1) pair - O(B^2) executables, never evicted.  Now a group's re-encode count is
    rounded up to a multiple of 8 (re-encoding a clip without a next chunk is what the reference does, app/models.py:111-114) and the cache
    is an LRU of at most 96 executables.  Here: 40 calls with random ragged batches of 9..16 tiny clips (1..3 chunks each); the cache
    must stay within its bound, the SAME ragged batch twice must capture nothing new, and results must equal the single-clip runs'
    decisions (the bar of test_ragged_batch: identical bits)."""
    from artalk_amd.synth import synth_audio
    m = get_gpu_model("tiny")
    m.set_precision("f16x3")
    rng = np.random.default_rng(5)
    lens = [2.0, 4.0, 5.5, 8.0, 9.0, 11.9]
    clips = {(i, s): torch.from_numpy(synth_audio(900 + i, s)) for i in range(16) for s in lens}
    held0, cap0 = m.graph_count()
    for it in range(40):
        B = int(rng.integers(9, 17))
        audios = [clips[(i, lens[int(rng.integers(0, len(lens)))])] for i in range(B)]
        m.inference_batch(audios)
        held, cap = m.graph_count()
        assert held <= 96, held
    # per active-batch size B (9..16 here, and every smaller B a ragged batch decays to) a group of g clips has at most 1 + ceil(g / 8) keys
    assert cap - cap0 <= 16 * 2 * 3, f"{cap - cap0} captures for 40 ragged calls: the key space is not bounded"
    audios = [clips[(i, lens[i % len(lens)])] for i in range(12)]
    outs = m.inference_batch(audios, None, return_aux=True)
    bits = [b.cpu() for b in m.last_aux["bits"]]
    _, cap1 = m.graph_count()
    outs2 = m.inference_batch(audios, None, return_aux=True)
    assert m.graph_count()[1] == cap1, "the same ragged batch captured new graphs the second time"
    for a, b in zip(outs, outs2):
        assert torch.equal(a, b)
    for i in (0, 5, 11):
        m.inference_batch([audios[i]], None, return_aux=True)
        assert torch.equal(m.last_aux["bits"][0].cpu(), bits[i]), f"clip {i}: ragged-batch decisions differ from the single run"
    assert m.status() == 0


def test_kernel_timer_budget_matches_the_graph_path():
    """artalk_set_profiling(3) (bench.py `budget_ms`): every launch goes through hipExtLaunchKernelGGL with its own start / stop events and
    the durations are summed per stage / scale step (artalk_get_kernel_sums).  The pass runs the SAME clip groups eagerly, one after the
    other: its results must be bit-identical to the graph replay's, every stage must have been timed, and the sums must be plausible
    (positive, the encoder's above the conv stack's on this model, kernels counted)."""
    from artalk_amd.synth import synth_audio
    m = get_gpu_model("tiny")
    m.set_precision("f16x3")
    try:
        audios = [torch.from_numpy(synth_audio(700 + i, 8.0)) for i in range(10)]
        want = [o.clone() for o in m.inference_batch(audios)]
        m.set_profiling(3)
        got = m.inference_batch(audios)
        ks = m.get_kernel_sums()
        m.set_profiling(0)
        for a, b in zip(want, got):
            assert torch.equal(a, b), "the timed eager pass differs from the graph replay"
        for k in ("w2v_conv", "w2v_encoder", "ada", "ar_history_kv", "level0", "level1", "level2", "level3", "level4", "vae_decode", "reencode"):
            assert ks[k] > 0.0, (k, ks)
        assert ks["kernels"] > 500 and ks["w2v_encoder"] < 1000.0 and ks["level4"] > ks["level0"] * 0.5
        again = m.inference_batch(audios)          # back on the graph path
        for a, b in zip(want, again):
            assert torch.equal(a, b)
        with pytest.raises(RuntimeError):
            m.set_profiling(0)
            mm = get_gpu_model("tiny")
            from artalk_amd import capi
            import ctypes as C
            out = (C.c_double * 4)()
            if capi.lib().artalk_get_kernel_sums(mm._h, out, 4) != capi.OK:      # n < 14
                raise RuntimeError("EINVAL")
    finally:
        m.set_profiling(0)
        m.set_precision("f32")
