#!/usr/bin/env python
"""Two replicas of the path side by side on disjoint halves of the chip (CU-masked streams) against one replica on the whole chip.
Under full-chip matrix load the MI355X drops its shader clock to ~1.6-1.8 GHz (profiles/r03_mfma_util.log); a half-chip partition runs
at a higher clock, and the latency-bound AR/VAE body does not need the whole chip.  Prints frames/s for: one model, all CUs; two
models on the low / high 128 mask bits (16 CUs of every XCD each); two models, no masks."""
import ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from artalk_amd import capi
from artalk_amd.config import ARTalkConfig
from artalk_amd.model import BitwiseARModel
from artalk_amd.synth import synth_audio
from artalk_amd.weights import generate_state_dict

B = int(os.environ.get("PROBE_BATCH", "32"))
STEPS = int(os.environ.get("PROBE_STEPS", "6"))
cfg = ARTalkConfig.by_name("full")
sd = generate_state_dict(cfg)
L = capi.lib()


def masked_stream(words):
    arr = (C.c_uint32 * 8)(*words)
    out = C.c_void_p()
    assert L.artalk_op_create_masked_stream(C.cast(arr, C.c_void_p), 8, C.byref(out)) == 0
    return torch.cuda.ExternalStream(out.value)


def make_model(mask=None):
    m = BitwiseARModel(cfg).eval().to("cuda")
    m.load_state_dict(sd, strict=True)
    if mask is not None:
        m.set_cu_mask(mask)
    m.reserve(B, B * 3)
    return m


audio = [torch.from_numpy(synth_audio(i, 10.0)).pin_memory() for i in range(B)]
frames = B * 250


def run(models, callers, steps):
    # every replica has its own caller stream; a batch is enqueued on each in turn, the host waits for a replica's previous batch
    pend = [None] * len(models)
    t0 = None
    done = 0
    for it in range(steps + 2):
        if it == 2:
            torch.cuda.synchronize(); t0 = time.perf_counter(); done = 0
        for k, (m, cs) in enumerate(zip(models, callers)):
            with torch.cuda.stream(cs):
                if pend[k] is not None:
                    pend[k][1].synchronize()
                    done += 1
                outs = m.inference_batch(audio, None, check=False)
                ev = torch.cuda.Event(); ev.record(cs)
                pend[k] = (outs, ev)
    for p in pend:
        p[1].synchronize(); done += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nb = steps * len(models)
    return nb * frames / dt, dt / nb * 1e3


lo, hi = [0xffffffff] * 4 + [0] * 4, [0] * 4 + [0xffffffff] * 4
one = make_model()
fps, ms = run([one], [torch.cuda.Stream()], STEPS)
print(f"one replica, whole chip:            {fps:9.0f} frames/s  {ms:6.2f} ms per batch", flush=True)
ma, mb = make_model(lo), make_model(hi)
fps, ms = run([ma, mb], [torch.cuda.Stream(), torch.cuda.Stream()], STEPS)
print(f"two replicas, half chip each:       {fps:9.0f} frames/s  {ms:6.2f} ms per batch", flush=True)
mc, md = make_model(), make_model()
fps, ms = run([mc, md], [torch.cuda.Stream(), torch.cuda.Stream()], STEPS)
print(f"two replicas, unmasked streams:     {fps:9.0f} frames/s  {ms:6.2f} ms per batch", flush=True)
