#!/usr/bin/env python
"""Per-shape throughput of the fp32 MFMA GEMM for each tile configuration (interleaved rounds, one process)."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from artalk_amd import capi
L = capi.lib()
p = lambda t: C.c_void_p(t.data_ptr())
SHAPES = [  # (name, M, N, K)
    ("w2v qkv", 19200, 3072, 1024), ("w2v out", 19200, 1024, 1024), ("w2v ff1", 19200, 4096, 1024), ("w2v ff2", 19200, 1024, 4096),
    ("conv1", 614400, 512, 1536), ("conv2", 307200, 512, 1536), ("conv6", 19200, 512, 1024), ("ada", 5792, 56832, 1024),
    ("ar qkv p4", 3200, 2304, 768), ("ar proj p4", 3200, 768, 768), ("ar ffn1 p4", 3200, 3072, 768), ("ar ffn2 p4", 3200, 768, 3072),
    ("ar qkv p2", 800, 2304, 768), ("ar ffn2 p2", 800, 768, 3072), ("ar qkv p0", 32, 2304, 768), ("ar ffn2 p0", 32, 768, 3072),
    ("vae qkv", 6400, 1536, 512), ("vae out", 6400, 512, 512),
]
only = os.environ.get("GEMM_ONLY")
if only:
    SHAPES = [x for x in SHAPES if x[0] in only.split(",")]
cfgs = [int(c) for c in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,1,2,3".split(","))]
# correctness of every requested configuration on an awkward shape first
A = torch.randn(777, 96, device='cuda'); W = torch.randn(333, 96, device='cuda'); b = torch.randn(333, device='cuda')
ref = (A.double() @ W.double().t() + b.double()).float()
for cfg in cfgs:
    Cc = torch.zeros(777, 333, device='cuda')
    L.artalk_op_gemm_ex(p(A), 96, p(W), p(b), p(Cc), 777, 333, 96, 0, cfg, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    print(f'cfg{cfg} max err {float((Cc - ref).abs().max()):.2e}', flush=True)
for name, M, N, K in SHAPES:
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.03; b = torch.randn(N, device="cuda")
    Cc = torch.empty(M, N, device="cuda")
    res = {}
    for rnd in range(3):
        for cfg in cfgs:
            if cfg == 3 and M > 4096: continue
            if cfg == 0 and M * N < 128 * 128 * 8: continue
            s = torch.cuda.current_stream().cuda_stream
            n = 3 if M * N * K > 1e11 else 10
            L.artalk_op_gemm_ex(p(A), K, p(W), p(b), p(Cc), M, N, K, 0, cfg, C.c_void_p(s))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                L.artalk_op_gemm_ex(p(A), K, p(W), p(b), p(Cc), M, N, K, 0, cfg, C.c_void_p(s))
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            res.setdefault(cfg, []).append(ms)
    line = f"{name:12s} M={M:6d} N={N:5d} K={K:4d} "
    for cfg in cfgs:
        if cfg in res:
            ms = min(res[cfg]); line += f"| cfg{cfg}: {ms*1e3:8.1f} us {2*M*N*K/ms/1e9:6.1f} TF "
    print(line, flush=True)
