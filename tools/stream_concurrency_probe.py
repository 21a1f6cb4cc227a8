#!/usr/bin/env python
"""How well do kernels of several HIP streams overlap on this stack?  Chains of small dependent GEMM launches (the shapes of the AR
decoder's first scale steps: M = 16 / 80 / 400 rows, K = 768), captured as hipGraphs and replayed on 1, 2, 3 and 4 streams at once.
Perfect overlap of latency-bound chains: the time of ONE chain whatever the number of streams."""
import ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from artalk_amd import capi
L = capi.lib()
p = lambda t: C.c_void_p(t.data_ptr())
dev = torch.device("cuda")
NCH, NL = 4, 96          # chains, launches per chain


def build_chain(M, seed):
    torch.manual_seed(seed)
    A = torch.randn(M, 768, device=dev); Ap = torch.empty(M, 768, dtype=torch.int32, device=dev)
    Ws = [torch.randn(2304, 768, device=dev) * 0.03 for _ in range(8)]
    Wps = [torch.empty(2304, 768, dtype=torch.int32, device=dev) for _ in range(8)]
    b = torch.randn(2304, device=dev); Cc = torch.empty(M, 2304, device=dev)
    s0 = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.artalk_op_pack_split(p(A), p(Ap), M * 768, 0, s0)
    for W, Wp in zip(Ws, Wps):
        L.artalk_op_pack_split(p(W), p(Wp), 2304 * 768, 1, s0)
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    L.artalk_op_gemm_f16s_packed(p(Ap), 1, 768, p(Wps[0]), p(b), p(Cc), M, 2304, 768, 0, 99, C.c_void_p(st.cuda_stream))   # warm-up: scratch
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=st):
        gs = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for i in range(NL):
            L.artalk_op_gemm_f16s_packed(p(Ap), 1, 768, p(Wps[i % 8]), p(b), p(Cc), M, 2304, 768, 0, 99, gs)
    return dict(g=g, st=st, keep=(A, Ap, Ws, Wps, b, Cc))


for M in (16, 80, 400, 1600):
    chains = [build_chain(M, i) for i in range(NCH)]
    res = []
    for n in (1, 2, 3, 4):
        ts = []
        for rep in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for c in chains[:n]:
                with torch.cuda.stream(c["st"]):
                    c["g"].replay()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        res.append(sorted(ts)[len(ts) // 2] * 1e6)
    print(f"M={M:5d}: {NL} dependent q|k|v GEMMs per chain; 1 stream {res[0]:8.0f} us | 2 streams {res[1]:8.0f} | 3 streams {res[2]:8.0f} | 4 streams {res[3]:8.0f}   "
          f"(per launch: {res[0] / NL:.1f} us alone)", flush=True)
    del chains
