#!/usr/bin/env python
"""Sustained fp16 MFMA rate and in-kernel clock (register-only loops on non-trivial data): the practical ceiling of the f16x3 split
GEMM is this rate / 3.  Shapes: v_mfma_f32_32x32x16_f16 (what the GEMMs issue) and v_mfma_f32_16x16x32_f16."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from artalk_amd import capi
L = capi.lib()
out = torch.empty(4096 * 256, device="cuda")
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
fl = C.c_double()
for nacc, name in ((16, "32x32x16_f16"), (17, "16x16x32_f16"), (18, "32x32x16_f16 changing operands"), (19, "16x16x32_f16 changing operands")):
    for blocks in (256, 512, 1024):
        iters = 40000
        L.artalk_op_mfma_f32_peak(C.c_void_p(out.data_ptr()), blocks, 2000, nacc, C.byref(fl), s)
        torch.cuda.synchronize()
        best = 1e9
        for rnd in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); L.artalk_op_mfma_f32_peak(C.c_void_p(out.data_ptr()), blocks, iters, nacc, C.byref(fl), s); e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        clk = out.view(-1, 256)[:blocks, 0].median().item()
        print(f"{name} blocks={blocks:5d} ({blocks / 256:.0f} waves/SIMD): {best:8.3f} ms  {fl.value / best / 1e9:7.1f} TFLOP/s  "
              f"in-kernel clock {clk:.2f} GHz  => f16x3 ceiling {fl.value / best / 1e9 / 3:6.1f} TF/s", flush=True)
