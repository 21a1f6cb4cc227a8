#!/usr/bin/env python
"""Soak of the persistent ping-pong encoder attention (attention_f16_pp_kernel): the same launch many times, every result bit-identical
to the first (a race in its LDS-DMA / barrier protocol would show as an occasional difference), at the model's shape and at a ragged one."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from artalk_amd import capi
L = capi.lib()
p = lambda t: C.c_void_p(t.data_ptr())
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for B, H, Lq, Lk in ((96, 16, 199, 199), (37, 16, 160, 150), (40, 16, 208, 224)):
    D = H * 64
    g = torch.Generator().manual_seed(B)
    ts = [torch.randn(B, n, D, generator=g).cuda() for n in (Lq, Lk, Lk)]
    pk = []
    for t in ts:
        o = torch.empty_like(t, dtype=torch.int32)
        assert L.artalk_op_pack_split(p(t), p(o), t.numel(), 0, None) == 0
        pk.append(o)
    outs = [torch.full((B, Lq, D), float("nan"), device="cuda") for _ in range(2)]
    first = None
    bad = 0
    n = int(os.environ.get("SOAK_N", "300"))
    for it in range(n):
        out = outs[it & 1]
        out.fill_(float("nan"))
        assert L.artalk_op_attention(p(pk[0]), p(pk[1]), p(pk[2]), p(out), B, H, 64, Lq, Lk, 0.125, 2 | 4, None, 0, s) == 0
        if first is None:
            first = out.clone()
            assert torch.isfinite(first).all()
        elif not torch.equal(out, first):
            bad += 1
    print(f"B={B} H={H} Lq={Lq} Lk={Lk}: {n} launches, {bad} differ from the first", flush=True)
    assert bad == 0
