#!/usr/bin/env python
"""Where a tile of the LDS-DMA split GEMM spends its time: per-workgroup wall-clock stamps of the 256x256 kernel (timing build, force_cfg 17).
Prints, per shape: median prologue (entry -> first stage landed), main loop, epilogue, the gap between consecutive
workgroups on the same CU, and the kernel's span."""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from artalk_amd import capi
L = capi.lib()
p = lambda t: C.c_void_p(t.data_ptr())
SHAPES = [("w2v out", 19200, 1024, 1024), ("w2v ff1", 19200, 4096, 1024), ("w2v ff2", 19200, 1024, 4096), ("K=256", 19200, 1024, 256)]
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
CFG, T = 17, 256
for name, M, N, K in SHAPES:
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.03
    Ap = torch.empty(M, K, dtype=torch.int32, device="cuda"); Wp = torch.empty(N, K, dtype=torch.int32, device="cuda")
    L.artalk_op_pack_split(p(A), p(Ap), M * K, 0, s); L.artalk_op_pack_split(p(W), p(Wp), N * K, 1, s)
    Cc = torch.empty(M, N, device="cuda")
    tiles = ((M + T - 1) // T) * ((N + T - 1) // T)
    st = torch.zeros(tiles, 8, dtype=torch.int64, device="cuda")
    for _ in range(3):
        assert L.artalk_op_gemm_f16s_packed(p(Ap), 1, K, p(Wp), p(st), p(Cc), M, N, K, 0, CFG, s) == 0
    torch.cuda.synchronize()
    t = st.cpu().numpy()
    us = lambda x: x / 100.0                       # 100 MHz counter
    t0, t1, t2, t3 = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
    hw, xcc = t[:, 4], t[:, 5] & 0xf
    cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8)
    span = us(t3.max() - t0.min())
    gaps = []
    per_cu = {}
    for c in np.unique(cu):
        idx = np.where(cu == c)[0]
        o = idx[np.argsort(t0[idx])]
        per_cu[c] = len(o)
        gaps += list(us(t0[o[1:]] - t3[o[:-1]]))
    med = lambda a: float(np.median(a)) if len(a) else float("nan")
    print(f"{name:8s} M={M} N={N} K={K} tiles={tiles} cus={len(per_cu)} tiles/cu {min(per_cu.values())}-{max(per_cu.values())} span {span:7.1f} us | "
          f"prologue {med(us(t1 - t0)):5.2f} loop {med(us(t2 - t1)):6.2f} ({med(us(t2 - t1)) / (K // 32):.3f}/step) epilogue {med(us(t3 - t2)):5.2f} "
          f"gap {med(gaps):5.2f} (p90 {float(np.percentile(gaps, 90)) if gaps else 0:5.2f}) first-start spread {us(np.sort(t0)[min(255, tiles - 1)] - t0.min()):5.2f}", flush=True)
