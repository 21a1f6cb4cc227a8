#!/usr/bin/env python
"""Per-shape time of the attention kernel."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from artalk_amd import capi
L = capi.lib()
p = lambda t: C.c_void_p(t.data_ptr())
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, B, H, HD, Lq, Lk, l2, split in [("w2v", 96, 16, 64, 199, 199, 0, 0), ("ar p4", 16, 12, 64, 100, 362, 1, 0), ("ar p3", 16, 12, 64, 50, 262, 1, 0),
                                          ("ar p2", 16, 12, 64, 25, 212, 1, 0), ("ar p1", 16, 12, 64, 5, 187, 1, 0), ("ar p0", 16, 12, 64, 1, 182, 1, 0),
                                          ("vae dec", 16, 8, 64, 200, 200, 0, 100), ("vae enc", 16, 8, 64, 100, 100, 0, 0)]:
    if os.environ.get("ATTN_ONLY") and name not in os.environ["ATTN_ONLY"].split(","):
        continue
    D = H * HD
    Q = torch.randn(B, Lq, D, device="cuda"); K = torch.randn(B, Lk, D, device="cuda"); V = torch.randn(B, Lk, D, device="cuda")
    O = torch.empty(B, Lq, D, device="cuda"); qs = torch.ones(H, device="cuda")
    fl = 4.0 * B * H * Lq * Lk * HD
    line = f"{name:8s} B={B} H={H} Lq={Lq} Lk={Lk}:"
    for mode, tag in ((l2, "fp32 mfma"), (l2 | 2, "f16 split")) + (((2 | 4, "f16 split, P8 in (timing only)"),) if not l2 else ()):
        best = 1e9
        for _ in range(3):
            L.artalk_op_attention(p(Q), p(K), p(V), p(O), B, H, HD, Lq, Lk, 0.125, mode, p(qs), split, s)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                L.artalk_op_attention(p(Q), p(K), p(V), p(O), B, H, HD, Lq, Lk, 0.125, mode, p(qs), split, s)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        line += f"  {tag} {best*1e3:7.1f} us {fl/best/1e9:6.1f} TF"
    print(line, flush=True)
