#!/usr/bin/env python
"""Does v_mfma_f32_32x32x16_f16 honour subnormal f16 inputs on this device?  Uses the split GEMM with hand-made P8 operands:
A = (hi 0, lo = a subnormal), W = (hi 1, lo 0)  ->  C = K * lo / 4096 (the operand scales) if subnormals count, 0 if they are
flushed.  Result on MI355X: they count, which is what lets the P8 format keep its residual unscaled (common.h)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from artalk_amd import capi
L = capi.lib()
p = lambda t: C.c_void_p(t.data_ptr())
M, N, K = 64, 64, 32
for bits in (0x0001, 0x0010, 0x0200, 0x03ff, 0x0400):
    a = np.zeros((M, K // 8, 16), dtype=np.uint16); a[:, :, 8:] = bits
    w = np.zeros((N, K // 8, 16), dtype=np.uint16); w[:, :, :8] = 0x3C00
    A = torch.from_numpy(a.view(np.int32).reshape(M, K).copy()).cuda(); W = torch.from_numpy(w.view(np.int32).reshape(N, K).copy()).cuda()
    out = torch.empty(M, N, device="cuda")
    assert L.artalk_op_gemm_f16s_packed(p(A), 1, K, p(W), None, p(out), M, N, K, 0, 1, None) == 0
    torch.cuda.synchronize()
    val = float(np.array([bits], dtype=np.uint16).view(np.float16)[0])
    print(f"lo bits {bits:#06x} = {val:.3e}: C = {out[0,0].item():.6e}  expected {K * val / 4096:.6e}")
