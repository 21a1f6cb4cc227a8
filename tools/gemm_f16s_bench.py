#!/usr/bin/env python
"""Throughput of the f16x3 split GEMM kernels per shape (GEMM_ONLY="name,..." and GEMM_VARIANTS="cfg:packed,..." select)."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from artalk_amd import capi
L = capi.lib()
p = lambda t: C.c_void_p(t.data_ptr())
SHAPES = [("w2v qkv", 19200, 3072, 1024), ("w2v out", 19200, 1024, 1024), ("w2v ff1", 19200, 4096, 1024), ("w2v ff2", 19200, 1024, 4096),
          ("conv1", 614400, 512, 1536), ("ada", 5792, 56832, 1024), ("ar ffn1 p4", 3200, 3072, 768), ("ar ffn2 p4", 3200, 768, 3072),
          ("ar qkv p2", 800, 2304, 768), ("ar ffn2 p2", 800, 768, 3072),
          # per clip group of 16 (what one of the two concurrent AR graphs launches at batch 32)
          ("g qkv p4", 1600, 2304, 768), ("g proj p4", 1600, 768, 768), ("g ffn1 p4", 1600, 3072, 768), ("g ffn2 p4", 1600, 768, 3072),
          ("g qkv p3", 800, 2304, 768), ("g proj p3", 800, 768, 768), ("g ffn1 p3", 800, 3072, 768), ("g ffn2 p3", 800, 768, 3072),
          ("g qkv p2", 400, 2304, 768), ("g proj p2", 400, 768, 768), ("g ffn1 p2", 400, 3072, 768), ("g ffn2 p2", 400, 768, 3072),
          ("g qkv p1", 80, 2304, 768), ("g ffn2 p1", 80, 768, 3072), ("g hist kv", 2896, 1536, 768)]
# the four GEMMs of an AR block at every row count a scale step can have: "t<M> qkv|proj|ffn1|ffn2"
for _M in (16, 32, 80, 160, 400, 800, 1600, 3200):
    SHAPES += [(f"t{_M} qkv", _M, 2304, 768), (f"t{_M} proj", _M, 768, 768), (f"t{_M} ffn1", _M, 3072, 768), (f"t{_M} ffn2", _M, 768, 3072)]
# the VAE decoder stack of one clip group of 16 (200 tokens per clip, hidden 512, MLP 768)
SHAPES += [("v qkv", 3200, 1536, 512), ("v proj", 3200, 512, 512), ("v mlp1", 3200, 768, 512), ("v mlp2", 3200, 512, 768)]
# the same stack for a merged group of 32 clips, and the re-encode stack (100 tokens per clip) of 16 / 32 clips
SHAPES += [("w qkv", 6400, 1536, 512), ("w proj", 6400, 512, 512), ("w mlp1", 6400, 768, 512), ("w mlp2", 6400, 512, 768)]
SHAPES += [("e qkv", 1600, 1536, 512), ("e proj", 1600, 512, 512), ("e mlp1", 1600, 768, 512), ("e mlp2", 1600, 512, 768)]
# row counts whose 128x128 tile counts are whole multiples of the 512 persistent workgroups (round-quantisation check)
for _M in (8192, 16384):
    SHAPES += [(f"q{_M} qkv", _M, 3072, 1024), (f"q{_M} out", _M, 1024, 1024), (f"q{_M} ff1", _M, 4096, 1024), (f"q{_M} ff2", _M, 1024, 4096)]
variants = [(1, 1), (7, 1), (8, 1), (99, 1)]   # (cfg, A packed): 0/1 register-staged 128x128 / 64x64; LDS-DMA kernels: 7 256x256, 8 persistent
                                                # two-workgroup 128x128, 99 the dispatch's own choice; small-grid 64x64: 20 (4 stages), 23 (8), 24 (5); cfg | S << 8 = split-K S (20, 23, 24)
# one-round launches of the 320x256 kernel on 30 / 60 / 120 / 240 of the 256 CUs (same work per CU): is the per-CU rate load-dependent?
for _M in (2400, 4800, 9600, 19200):
    SHAPES += [(f"p{_M} ff2", _M, 1024, 4096)]
only = os.environ.get("GEMM_ONLY")
if only:
    SHAPES = [x for x in SHAPES if x[0] in only.split(",") or any(o.endswith("*") and x[0].startswith(o[:-1]) for o in only.split(","))]
# GEMM_ROTATE=n: cycle through n copies of the weight matrix, so that (as in the model, whose 12 blocks hold 340 MB of split
# weights) a launch does not find its weights in L2 / Infinity Cache from the previous launch
ROT = int(os.environ.get("GEMM_ROTATE", "1"))
ACT = int(os.environ.get("GEMM_ACT", "0"), 0)      # activation | 0x100 = P8 result (timing only: the error column is then meaningless)
if os.environ.get("GEMM_VARIANTS"):
    variants = [tuple(int(v) for v in x.split(":")) for x in os.environ["GEMM_VARIANTS"].split(",")]
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
# warm-up: the first kernels of a process run at ramping clocks (the first variant of the first shape measured 10 % slow without it)
_a = torch.randn(8192, 8192, device="cuda")
for _ in range(30):
    _a = (_a @ _a) * 1e-4
torch.cuda.synchronize()
del _a
for name, M, N, K in SHAPES:
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.03; b = torch.randn(N, device="cuda")
    Ap = torch.empty(M, K, dtype=torch.int32, device="cuda"); Wp = torch.empty(N, K, dtype=torch.int32, device="cuda")
    L.artalk_op_pack_split(p(A), p(Ap), M * K, 0, s); L.artalk_op_pack_split(p(W), p(Wp), N * K, 1, s)
    Wps = [Wp] + [Wp.clone() for _ in range(ROT - 1)]
    Cc = torch.empty(M, N, device="cuda")
    ref = A[:256].double() @ W.double().t() + b.double()
    line = f"{name:12s} M={M:6d} N={N:5d} K={K:4d} "
    # variants are timed in INTERLEAVED rounds (A B C A B C ...) and reported by their median: box temperature and clock state drift by
    # several per cent over a process, which a variant-after-variant order turns into a bias
    runs = []
    for cfg, apk in variants:
        n = 3 if M * N * K > 1e11 else (10 if ROT == 1 else ROT)
        a = Ap if apk else A
        L.artalk_op_gemm_f16s_packed(p(a), apk, K, p(Wp), p(b), p(Cc), M, N, K, ACT, cfg, s)      # warm-up (allocates the split-K scratch)
        torch.cuda.synchronize()
        # the error column is a CHECK only for an fp32 result without an in-place residual: a P8 result (ACT & 0x100) read as fp32, or a
        # residual accumulated over the repeats (ACT & 0x200), is timing only and printed as n/a
        err = None if (ACT & 0x300) else float((Cc[:256].double() - ref).abs().max() / ref.abs().max())
        graph = None
        if os.environ.get("GEMM_GRAPH", "1") == "1" and M * N * K < 1e11:
            # the launches replayed from a hipGraph, as the model runs them: eager launches are host-bound below ~3.5 us per kernel
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                gs = C.c_void_p(torch.cuda.current_stream().cuda_stream)
                for it in range(n):
                    L.artalk_op_gemm_f16s_packed(p(a), apk, K, p(Wps[it % ROT]), p(b), p(Cc), M, N, K, ACT, cfg, gs)
        runs.append(dict(cfg=cfg, apk=apk, a=a, n=n, graph=graph, err=err, ts=[]))
    for rnd in range(int(os.environ.get("GEMM_ROUNDS", "5"))):
        for v in runs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if v["graph"] is not None:
                v["graph"].replay()
            else:
                for it in range(v["n"]):
                    L.artalk_op_gemm_f16s_packed(p(v["a"]), v["apk"], K, p(Wps[it % ROT]), p(b), p(Cc), M, N, K, ACT, v["cfg"], s)
            e1.record(); torch.cuda.synchronize()
            v["ts"].append(e0.elapsed_time(e1) / v["n"])
    for v in runs:
        best = sorted(v["ts"])[len(v["ts"]) // 2]
        cfg = v["cfg"]
        line += f"| cfg{cfg & 0xff}{'/%d' % (cfg >> 8) if cfg >> 8 else ''}{'P' if v['apk'] else ' '}: {best*1e3:8.1f} us {2*M*N*K/best/1e9:6.1f} TF e={'n/a  ' if v['err'] is None else '%.0e' % v['err']} "
    print(line, flush=True)
    del runs      # the graphs that replay launches into the split-K scratch are gone: retired scratch buffers may be freed
    torch.cuda.synchronize()
    L.artalk_op_release_scratch()
