#!/usr/bin/env python
"""Would running the wav2vec2 encoder as two concurrent half-batches hide kernel tails?  Times the chain of a layer's four GEMMs
(engine's kernel choice, cfg 99) over 12 layers: one stream at M = 19200 against two streams at M = 9600 each."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from artalk_amd import capi
L = capi.lib()
p = lambda t: C.c_void_p(t.data_ptr())
SH = [(3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096)]
def mk(M):
    d = {}
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for i, (N, K) in enumerate(SH):
        A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.03; b = torch.randn(N, device="cuda")
        Ap = torch.empty(M, K, dtype=torch.int32, device="cuda"); Wp = torch.empty(N, K, dtype=torch.int32, device="cuda")
        L.artalk_op_pack_split(p(A), p(Ap), M * K, 0, s); L.artalk_op_pack_split(p(W), p(Wp), N * K, 1, s)
        d[i] = (Ap, Wp, b, torch.empty(M, N, device="cuda"), M, N, K)
    torch.cuda.synchronize()
    return d
def chain(d, stream, layers=12):
    s = C.c_void_p(stream.cuda_stream)
    for _ in range(layers):
        for i in range(4):
            Ap, Wp, b, Cc, M, N, K = d[i]
            L.artalk_op_gemm_f16s_packed(p(Ap), 1, K, p(Wp), p(b), p(Cc), M, N, K, 0, 99, s)
full = mk(19200); h0 = mk(9600); h1 = mk(9600)
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
def timeit(fn, n=3):
    best = 1e9
    for _ in range(n):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best
def one():
    chain(full, torch.cuda.current_stream())
def two():
    cur = torch.cuda.current_stream()
    s0.wait_stream(cur); s1.wait_stream(cur)
    chain(h0, s0); chain(h1, s1)
    cur.wait_stream(s0); cur.wait_stream(s1)
def seq_halves():
    chain(h0, torch.cuda.current_stream()); chain(h1, torch.cuda.current_stream())
one(); two(); seq_halves()
print(f"one stream, M=19200: {timeit(one):.2f} ms   two streams, M=9600 each: {timeit(two):.2f} ms   halves back to back on one stream: {timeit(seq_halves):.2f} ms", flush=True)
