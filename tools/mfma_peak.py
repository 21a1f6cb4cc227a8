#!/usr/bin/env python
"""Sustained fp32-MFMA rate of this chip (register-only loop), the ceiling the GEMM is compared with."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from artalk_amd import capi
L = capi.lib()
out = torch.empty(4096 * 256, device="cuda")
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
fl = C.c_double()
for nacc in (4, 1):
    for blocks in (256, 1024, 2048, 4096):
        iters = 5000
        L.artalk_op_mfma_f32_peak(C.c_void_p(out.data_ptr()), blocks, 100, nacc, C.byref(fl), s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); L.artalk_op_mfma_f32_peak(C.c_void_p(out.data_ptr()), blocks, iters, nacc, C.byref(fl), s); e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        print(f"nacc={nacc} blocks={blocks:5d} ({blocks // 256} waves/SIMD) iters={iters:6d}: {ms:8.3f} ms  {fl.value / ms / 1e9:7.1f} TFLOP/s", flush=True)
