"""Each HIP kernel, called through the C ABI (artalk_op_*), against torch fp32/fp64 CPU math of the same op.

Tolerances: the kernels are exact-fp32 (MFMA f32 is an fmaf chain), so differences to a float64 reference are
accumulation-order rounding only: |err| <= 2e-5 * sum|a*b| scale for GEMMs, 2e-5 absolute for O(1) outputs.
"""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _lib():
    from artalk_amd import capi
    return capi, capi.lib()


def _dev(t):
    return t.contiguous().cuda()


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


@pytest.mark.parametrize("M,N,K,act,use_bias,use_gate,use_res", [
    (1000, 512, 1536, 0, True, False, False),     # conv-as-GEMM shape, 128x128 tiles
    (2000, 1024, 1024, 1, True, False, True),     # gelu(erf) + residual
    (333, 768, 3072, 2, True, True, True),        # ragged M, gelu(tanh), gate, residual; 64x64 tiles
    (32, 2304, 768, 0, True, False, False),       # tiny M: 32x128 tiles
    (5, 64, 768, 0, True, False, False),          # logits head shape
    (200, 106, 512, 0, True, False, False),       # N not a multiple of anything
    (6400, 512, 32, 3, True, False, False),       # K = 32, leaky relu
    (40000, 64, 1024, 0, False, False, False),    # 128x64 tiles
])
def test_gemm(M, N, K, act, use_bias, use_gate, use_res):
    capi, L = _lib()
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    bias = torch.randn(N, generator=g) if use_bias else None
    gate = torch.randn(M, N, generator=g) if use_gate else None
    R = torch.randn(M, N, generator=g) if use_res else None
    ref = A.double() @ W.double().t()
    if bias is not None:
        ref = ref + bias.double()
    ref = ref.float()
    if act == 1:
        ref = F.gelu(ref)
    elif act == 2:
        ref = F.gelu(ref, approximate="tanh")
    elif act == 3:
        ref = F.leaky_relu(ref, 0.2)
    if gate is not None:
        ref = ref * gate
    if R is not None:
        ref = ref + R
    dA, dW = _dev(A), _dev(W)
    db, dg, dR = (_dev(x) if x is not None else None for x in (bias, gate, R))
    out = torch.full((M, N), float("nan"), device="cuda")
    rc = L.artalk_op_gemm(_p(dA), K, _p(dW), _p(db), _p(dg), _p(dR), _p(out), M, N, K, act, None)
    assert rc == 0
    torch.cuda.synchronize()
    err = (out.cpu() - ref).abs().max().item()
    assert err < 3e-5 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("M,N,K,cfg", [(1000, 512, 1536, 0), (2000, 1024, 1024, 1), (333, 768, 3072, -1), (19200, 1024, 4096, 0), (200, 106, 512, 1)])
def test_gemm_f16x3_split_is_fp32_accurate(M, N, K, cfg):
    """The operand-split fp16 GEMM (artalk_set_precision mode 1) against float64: its error must be at the level of an
    exact-fp32 GEMM (accumulation rounding), far from fp16/bf16 level; also checked against the fp32-MFMA kernel."""
    capi, L = _lib()
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g) * 3.0
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    bias = torch.randn(N, generator=g)
    ref = (A.double() @ W.double().t() + bias.double())
    dA, dW, db = _dev(A), _dev(W), _dev(bias)
    out = torch.full((M, N), float("nan"), device="cuda")
    out32 = torch.empty(M, N, device="cuda")
    assert L.artalk_op_gemm_f16s(_p(dA), K, _p(dW), _p(db), _p(out), M, N, K, 0, cfg, None) == 0
    assert L.artalk_op_gemm(_p(dA), K, _p(dW), _p(db), None, None, _p(out32), M, N, K, 0, None) == 0
    torch.cuda.synchronize()
    scale = ref.abs().max().item()
    err = (out.cpu().double() - ref).abs().max().item() / scale
    err32 = (out32.cpu().double() - ref).abs().max().item() / scale
    print(f"f16x3 rel err {err:.2e}  fp32-mfma rel err {err32:.2e}")
    # bar: no worse than the exact-fp32 MFMA kernel's own accumulation rounding (one fp32 accumulator per output in both)
    assert err < 3e-6 and err < 1.25 * err32 + 2e-7, (err, err32)


def test_gemm_p8_dma_pipeline_and_producers():
    """The LDS-DMA pipelined split GEMM (both operands in the P8 split format) against float64, with A produced in P8 by
    the LayerNorm kernel's out_p8 mode (what the wav2vec2 stage does) and by the pack kernel."""
    capi, L = _lib()
    g = torch.Generator().manual_seed(11)
    M, N, K = 8192 + 77, 1024, 1024
    X = torch.randn(M, K, generator=g) * 2 + 0.1
    lw, lb = torch.randn(K, generator=g), torch.randn(K, generator=g)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    bias = torch.randn(N, generator=g)
    A = F.layer_norm(X.double(), (K,), lw.double(), lb.double(), 1e-5)
    ref = A @ W.double().t() + bias.double()
    dX, dlw, dlb, dW, db = _dev(X), _dev(lw), _dev(lb), _dev(W), _dev(bias)
    Ap = torch.empty(M, K, dtype=torch.int32, device="cuda")     # P8 has the fp32 pitch
    Wp = torch.empty(N, K, dtype=torch.int32, device="cuda")
    assert L.artalk_op_pack_split(_p(dW), _p(Wp), N * K, 1, None) == 0
    # LayerNorm -> P8 (flag in the high bits of `act`: see artalk_op_layernorm)
    assert L.artalk_op_layernorm(_p(dX), _p(Ap), _p(dlw), _p(dlb), None, None, M, K, 1e-5, 0x100, None) == 0
    out = torch.full((M, N), float("nan"), device="cuda")
    # LDS-DMA kernels: 7 / 12 = persistent 256x256 / 320x256 tiles (M = 8269 leaves an edge tile of 77 / 269 rows), 13 = the non-persistent
    # 256x256 kernel, 8 = persistent two-workgroup 128x128 (deferred epilogue), 99 = launch_gemm_p8's own choice; register-staged
    # 128x128 and 64x64 (0, 1); the small-grid LDS-DMA kernel (20) and its deep-ring split-K configurations (23, 24; cfg | S << 8 =
    # split-K S), all fed with the P8 activation
    for cfg in (7, 12, 13, 8, 99, 0, 1, 20, 20 | (3 << 8), 23 | (4 << 8), 24 | (6 << 8), 28, 28 | (3 << 8), 30, 31, 33, 31 | (3 << 8), 33 | (2 << 8)):      # 28: the mid-grid 128x128 kernel; 30 / 31 / 33: the ping-pong kernel (256x128 with one / two barriers per K step, 128x128)
        out.fill_(float("nan"))
        assert L.artalk_op_gemm_f16s_packed(_p(Ap), 1, K, _p(Wp), _p(db), _p(out), M, N, K, 0, cfg, None) == 0
        torch.cuda.synchronize()
        err = (out.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
        assert err < 2e-6, (cfg, err)
    # small ragged shapes of the AR scale steps through the small-grid kernel (M = 80 / 400 rows, K = 1024 here)
    for Ms in (80, 400):
        for cfg in (20, 20 | (4 << 8), 23 | (8 << 8), 24 | (2 << 8), 8, 28, 28 | (2 << 8), 31, 33, 31 | (4 << 8)):     # 8: the large-grid kernel on a grid smaller than the chip (one tile per workgroup)
            o2 = torch.full((Ms, N), float("nan"), device="cuda")
            assert L.artalk_op_gemm_f16s_packed(_p(Ap), 1, K, _p(Wp), _p(db), _p(o2), Ms, N, K, 0, cfg, None) == 0
            torch.cuda.synchronize()
            err = (o2.cpu().double() - ref[:Ms]).abs().max().item() / ref.abs().max().item()
            assert err < 2e-6, (Ms, cfg, err)


@pytest.mark.parametrize("M,N,K", [(19200, 3072, 1024),        # wav2vec2 q|k|v at batch 32
                                   (8192 + 77, 1024, 1024),     # a ragged M, one column round
                                   (5792, 2048, 1024)])         # AdaLN-table rows (32 x 181), not a multiple of any tile
@pytest.mark.parametrize("residual", [False, True])
def test_gemm_p8_auto_dispatch(M, N, K, residual):
    """launch_gemm_p8 with no forced configuration (what the model calls): the plan says which production kernel takes the shape
    (a big-tile kernel only without a residual), the result is bit-identical to that kernel forced - and, the accumulation order
    being the same in every split kernel, to the other production kernels - and correct against float64."""
    capi, L = _lib()
    want = L.artalk_op_gemm_p8_plan(M, N, K, int(residual))
    assert want in (7, 12, 8)
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    bias = torch.randn(N, generator=g)
    R = torch.randn(M, N, generator=g)
    dA, dW, db = _dev(A), _dev(W), _dev(bias)
    Ap = torch.empty(M, K, dtype=torch.int32, device="cuda")
    Wp = torch.empty(N, K, dtype=torch.int32, device="cuda")
    assert L.artalk_op_pack_split(_p(dA), _p(Ap), M * K, 0, None) == 0 and L.artalk_op_pack_split(_p(dW), _p(Wp), N * K, 1, None) == 0
    act = 0x200 if residual else 0            # 0x200: residual read from C, in place (as the encoder's out-projection runs)
    outs = []
    for cfg in (99, want, 7, 12, 13, 8):
        out = _dev(R.clone()) if residual else torch.full((M, N), float("nan"), device="cuda")
        assert L.artalk_op_gemm_f16s_packed(_p(Ap), 1, K, _p(Wp), _p(db), _p(out), M, N, K, act, cfg, None) == 0
        torch.cuda.synchronize()
        outs.append(out.cpu())
    for o in outs[1:]:
        assert torch.equal(outs[0], o)
    rows = torch.cat([torch.arange(0, 300), torch.arange(M - 300, M)])       # first and last tiles against float64
    ref = A[rows].double() @ W.double().t() + bias.double() + (R[rows].double() if residual else 0.0)
    err = (outs[0][rows].double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 2e-6, err


def _unpack_p8(t_i32, scale=16.0):
    """P8 split format (common.h) -> float64: every 8 elements are 32 bytes [8 x f16 hi][8 x f16 lo], value = (hi + lo) / scale."""
    M, K = t_i32.shape
    h = t_i32.cpu().view(torch.float16).reshape(M, K // 8, 2, 8).double()
    return ((h[:, :, 0, :] + h[:, :, 1, :]) / scale).reshape(M, K)


@pytest.mark.parametrize("M", [2560, 1000])
def test_gemm_p8_large_grid_kernels_p8_gelu_epilogue(M):
    """The epilogue the encoder's FFN-in GEMM uses - bias, GELU(erf), result written in the P8 split format - on every large-grid
    kernel (persistent 256x256 / 320x256 tiles, the non-persistent 256x256 kernel, persistent 128x128): unpacked against float64,
    and bit-identical across the kernels.  M = 1000 leaves edge tiles whose rows beyond M must not be stored."""
    capi, L = _lib()
    N, K = 1024, 512
    g = torch.Generator().manual_seed(M)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    bias = torch.randn(N, generator=g)
    ref = F.gelu(A.double() @ W.double().t() + bias.double())
    dA, dW, db = _dev(A), _dev(W), _dev(bias)
    Ap = torch.empty(M, K, dtype=torch.int32, device="cuda")
    Wp = torch.empty(N, K, dtype=torch.int32, device="cuda")
    assert L.artalk_op_pack_split(_p(dA), _p(Ap), M * K, 0, None) == 0 and L.artalk_op_pack_split(_p(dW), _p(Wp), N * K, 1, None) == 0
    outs = []
    for cfg in (7, 12, 13, 8):
        out = torch.full((M + 64, N), 0x7fc00000, dtype=torch.int32, device="cuda")      # canary rows behind the result
        assert L.artalk_op_gemm_f16s_packed(_p(Ap), 1, K, _p(Wp), _p(db), _p(out), M, N, K, 0x101, cfg, None) == 0
        torch.cuda.synchronize()
        assert bool((out[M:] == 0x7fc00000).all()), f"cfg {cfg} stored rows beyond M"
        got = _unpack_p8(out[:M])
        err = (got - ref).abs().max().item() / ref.abs().max().item()
        assert err < 2e-6, (cfg, err)
        outs.append(out[:M].cpu())
    for o in outs[1:]:
        assert torch.equal(outs[0], o)


@pytest.mark.parametrize("act,approx", [(1, "none"), (2, "tanh")])
def test_gelu_activations_accuracy(act, approx):
    """The branch-free GELUs of common.h (erf form: x * Phi(x) through log2(erfc) as one polynomial + v_exp_f32; tanh form through
    the sigmoid identity) against float64 over the range the headroom audit sees, through an exact GEMM with an identity weight
    (products by 1 and 0 are exact, so C = act(A)).  Reference: F.gelu as wav2vec2 / the AR FFN call it
    (modeling_wav2vec2 hidden_act "gelu", app/transformer.py:27 approximate='tanh')."""
    capi, L = _lib()
    M, K = 8192, 32
    x = torch.cat([torch.linspace(-8, 8, M * K - 4096), torch.randn(4096) * 1e-3]).reshape(M, K)
    W = torch.eye(K)
    out = torch.empty(M, K, device="cuda")
    dA, dW = _dev(x), _dev(W)
    assert L.artalk_op_gemm(_p(dA), K, _p(dW), None, None, None, _p(out), M, K, K, act, None) == 0
    torch.cuda.synchronize()
    got = out.cpu().double()
    ref = F.gelu(x.double(), approximate=approx)
    err = (got - ref).abs()
    assert err.max().item() < 6e-7, err.max().item()                       # absolute, anywhere in [-8, 8]
    pos = x > 0.01
    ulp = torch.from_numpy(np.spacing(ref.abs().float().numpy())).double()
    worst = (err / ulp)[pos].max().item()
    assert worst < 3.0, worst                                               # the libm formulas in fp32 reach 1.6 / 1.7 ulp
    # the fp32 formula the reference executes (torch CPU), against the same float64 truth: the two are equally far from it
    ref32 = F.gelu(x, approximate=approx).double()
    worst32 = ((ref32 - ref).abs() / ulp)[pos].max().item()
    print(f"gelu({approx}): kernel {worst:.2f} ulp, torch fp32 {worst32:.2f} ulp from float64 (x > 0.01); max abs err {err.max().item():.2e}")
    assert worst < max(3.0, 1.5 * worst32)


def test_gemm_exact_integers():
    """A = I-like and asymmetric small integers: the MFMA lane maps (no row/col swap) are exact in fp32."""
    capi, L = _lib()
    M, N, K = 160, 192, 64
    A = torch.randint(-3, 4, (M, K)).float()
    W = (torch.arange(N * K).reshape(N, K) % 7 - 3).float()
    ref = A @ W.t()
    out = torch.empty(M, N, device="cuda")
    dA, dW = _dev(A), _dev(W)
    assert L.artalk_op_gemm(_p(dA), K, _p(dW), None, None, None, _p(out), M, N, K, 0, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), ref)


@pytest.mark.parametrize("D,affine,mod,act,eps", [(512, True, False, 1, 1e-5), (1024, True, False, 0, 1e-5),
                                                    (768, False, True, 0, 1e-6), (128, True, False, 0, 1e-5)])
def test_layernorm(D, affine, mod, act, eps):
    capi, L = _lib()
    g = torch.Generator().manual_seed(D)
    M = 777
    X = torch.randn(M, D, generator=g) * 2 + 0.3
    w = torch.randn(D, generator=g) if affine else None
    b = torch.randn(D, generator=g) if affine else None
    sc = torch.randn(M, D, generator=g) if mod else None
    sh = torch.randn(M, D, generator=g) if mod else None
    ref = F.layer_norm(X.double(), (D,), w.double() if affine else None, b.double() if affine else None, eps).float()
    if mod:
        ref = ref * (sc + 1) + sh
    if act == 1:
        ref = F.gelu(ref)
    dX = _dev(X)
    dw, db, dsc, dsh = (_dev(x) if x is not None else None for x in (w, b, sc, sh))
    out = torch.empty(M, D, device="cuda")
    assert L.artalk_op_layernorm(_p(dX), _p(out), _p(dw), _p(db), _p(dsc), _p(dsh), M, D, eps, act, None) == 0
    torch.cuda.synchronize()
    assert (out.cpu() - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize("B,H,HD,Lq,Lk,l2norm,split", [
    (3, 16, 64, 199, 199, 0, 0),      # wav2vec2 encoder
    (2, 12, 64, 25, 212, 1, 0),       # AR step 2 with KV cache: 181 history + 6 + 25 keys
    (2, 12, 64, 1, 182, 1, 0),        # AR step 0 (short-query kernel: keys split over the waves)
    (2, 12, 64, 5, 187, 1, 0),        # AR step 1
    (2, 12, 64, 50, 262, 1, 0),       # AR step 3 (four query tiles)
    (2, 8, 64, 16, 70, 0, 0),         # short-query kernel without L2 norm, ragged last key tile
    (2, 12, 64, 100, 362, 1, 0),      # AR step 4
    (2, 8, 64, 200, 200, 0, 100),     # VAE decoder mask
    (2, 8, 64, 100, 100, 0, 0),       # VAE encoder
    (3, 4, 32, 50, 50, 0, 0),         # style encoder
])
def test_attention(B, H, HD, Lq, Lk, l2norm, split):
    capi, L = _lib()
    g = torch.Generator().manual_seed(Lq * 1000 + Lk)
    D = H * HD
    Q = torch.randn(B, Lq, D, generator=g)
    K = torch.randn(B, Lk, D, generator=g)
    V = torch.randn(B, Lk, D, generator=g)
    qs = (torch.rand(H, generator=g) * 4 + 1) if l2norm else None
    scale = 1.0 if l2norm else (512 ** -0.5 if split or (H == 8) else HD ** -0.5)
    q = Q.view(B, Lq, H, HD).transpose(1, 2).double()
    k = K.view(B, Lk, H, HD).transpose(1, 2).double()
    v = V.view(B, Lk, H, HD).transpose(1, 2).double()
    if l2norm:
        q = F.normalize(q, dim=-1) * qs.double().view(1, H, 1, 1)
        k = F.normalize(k, dim=-1)
    s = q @ k.transpose(-1, -2) * scale
    if split:
        mask = torch.zeros(Lq, Lk, dtype=torch.double)
        mask[:split, split:] = -float("inf")
        s = s + mask
    ref = (s.softmax(-1) @ v).transpose(1, 2).reshape(B, Lq, D).float()
    dQ, dK, dV = _dev(Q), _dev(K), _dev(V)
    dqs = _dev(qs) if qs is not None else None
    out = torch.full((B, Lq, D), float("nan"), device="cuda")
    assert L.artalk_op_attention(_p(dQ), _p(dK), _p(dV), _p(out), B, H, HD, Lq, Lk, scale, l2norm, _p(dqs), split, None) == 0
    torch.cuda.synchronize()
    err = (out.cpu() - ref).abs().max().item()
    assert err < 2e-5, err
    if HD == 64:     # the fp16 operand-split kernel of f16x3 mode (long-query shapes; short ones take the key-split fp32 kernel)
        out.fill_(float("nan"))
        assert L.artalk_op_attention(_p(dQ), _p(dK), _p(dV), _p(out), B, H, HD, Lq, Lk, scale, l2norm | 2, _p(dqs), split, None) == 0
        torch.cuda.synchronize()
        err16 = (out.cpu() - ref).abs().max().item()
        print(f"attention fp32-mfma err {err:.2e}  f16-split err {err16:.2e}")
        assert err16 < 2e-5, err16
        if not l2norm:      # the same kernel fed with P8 operands (what the wav2vec2 qkv GEMM hands over)
            pk = []
            for t in (dQ, dK, dV):
                o = torch.empty_like(t, dtype=torch.int32)
                assert L.artalk_op_pack_split(_p(t), _p(o), t.numel(), 0, None) == 0
                pk.append(o)
            out.fill_(float("nan"))
            assert L.artalk_op_attention(_p(pk[0]), _p(pk[1]), _p(pk[2]), _p(out), B, H, HD, Lq, Lk, scale, 2 | 4, None, split, None) == 0
            torch.cuda.synchronize()
            errp = (out.cpu() - ref).abs().max().item()
            assert errp < 2e-5, errp


_RES_CHILD = r"""
import ctypes as C, sys, math, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from artalk_amd import capi
L = capi.lib()
p = lambda t: C.c_void_p(t.data_ptr())
M, N, K, cfg = [int(x) for x in sys.argv[3:7]]
g = torch.Generator().manual_seed(5)
A = torch.randn(M, K, generator=g); W = torch.randn(N, K, generator=g) / math.sqrt(K); b = torch.randn(N, generator=g); R = torch.randn(M, N, generator=g)
dA, dW, db, out = A.cuda(), W.cuda(), b.cuda(), R.cuda()
Ap = torch.empty(M, K, dtype=torch.int32, device="cuda"); Wp = torch.empty(N, K, dtype=torch.int32, device="cuda")
assert L.artalk_op_pack_split(p(dA), p(Ap), M * K, 0, None) == 0 and L.artalk_op_pack_split(p(dW), p(Wp), N * K, 1, None) == 0
assert L.artalk_op_gemm_f16s_packed(p(Ap), 1, K, p(Wp), p(db), p(out), M, N, K, 0x200, cfg, None) == 0      # 0x200: residual read from C, in place
torch.cuda.synchronize()
ref = R.double() + A.double() @ W.double().t() + b.double()
err = float((out.cpu().double() - ref).abs().max() / ref.abs().max())
assert err < 2e-6, err
np.save(sys.argv[2], out.cpu().numpy())
"""


@pytest.mark.parametrize("M,N,K,cfg", [(19200 // 4 + 77, 1024, 1024, 8), (2000, 1024, 4096, 8), (8192 + 77, 1024, 1024, 99)])
def test_persistent_gemm_residual_deferred_equals_immediate(tmp_path, M, N, K, cfg):
    """Residual tiles of gemm_p8_2wgp_kernel (the encoder's out-projection / FFN-out, x += ... in place): correct against float64 and
    bit-identical whether the residual sub-tiles are deferred through the LDS-DMA slots (default) or finished at once
    (ARTALK_P8_RES_DEFER=0).  The switch is read once per process: one child per arm."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for defer in ("1", "0"):
        f = str(tmp_path / f"res{defer}.npy")
        env = dict(os.environ, ARTALK_P8_RES_DEFER=defer)
        subprocess.run([sys.executable, "-c", _RES_CHILD, root, f, str(M), str(N), str(K), str(cfg)], check=True, env=env, timeout=300)
        outs.append(np.load(f))
    assert np.array_equal(outs[0], outs[1])


_WIDE_CHILD = r"""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from artalk_amd import capi
L = capi.lib()
p = lambda t: C.c_void_p(t.data_ptr())
B, H, HD, Lq, Lk, split = [int(x) for x in sys.argv[3:9]]
ar = sys.argv[9] == "ar"
g = torch.Generator().manual_seed(77)
D = H * HD
ts = [torch.randn(B, n, D, generator=g).cuda() for n in (Lq, Lk, Lk)]
out = torch.full((B, Lq, D), float("nan"), device="cuda")
if ar:      # fp32 rows, L2-normalised q and k with a per-head scale: the AR decoder's attention (f16 operand-split kernels)
    qs = (torch.rand(H, generator=g) * 4 + 1).cuda()
    assert L.artalk_op_attention(p(ts[0]), p(ts[1]), p(ts[2]), p(out), B, H, HD, Lq, Lk, 1.0, 1 | 2, p(qs), 0, None) == 0
else:       # P8 rows as the qkv GEMM hands them over (wav2vec2 encoder, VAE stacks)
    pk = []
    for t in ts:
        o = torch.empty_like(t, dtype=torch.int32)
        assert L.artalk_op_pack_split(p(t), p(o), t.numel(), 0, None) == 0
        pk.append(o)
    assert L.artalk_op_attention(p(pk[0]), p(pk[1]), p(pk[2]), p(out), B, H, HD, Lq, Lk, 0.125, 2 | 4, None, split, None) == 0
torch.cuda.synchronize()
np.save(sys.argv[2], out.cpu().numpy())
"""


@pytest.mark.parametrize("B,H,Lq,Lk,split,mode", [(3, 16, 199, 199, 0, "p8"), (2, 8, 200, 200, 100, "p8"), (2, 8, 100, 100, 0, "p8"), (20, 16, 199, 199, 0, "p8"),
                                                  (3, 12, 100, 362, 0, "ar"), (2, 12, 97, 181, 0, "ar"),
                                                  # >= 2 heads per CU: the persistent ping-pong kernel (attention_f16_pp_kernel): the encoder's shape with a ragged
                                                  # last round, the largest shape it takes, few queries (idle waves), the shortest second buffer
                                                  (40, 16, 199, 199, 0, "p8"), (33, 16, 208, 224, 0, "p8"), (36, 16, 150, 170, 0, "p8"), (32, 16, 199, 129, 0, "p8")])
def test_attention_wide_kernel_is_bit_identical(tmp_path, B, H, Lq, Lk, split, mode):
    """attention_f16_wide_kernel / attention_f16_wide_ar_kernel (one workgroup per (clip, head), keys staged once / 192 at a time) and
    attention_f16_pp_kernel (persistent, keys in two LDS buffers filled by LDS-DMA under the arithmetic) must give bit for bit what
    the 64-query kernel gives: the switch is read once per process, so each arm runs in a child process (ARTALK_ATTN_WIDE=1 / 0)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for wide in ("1", "0"):
        f = str(tmp_path / f"wide{wide}.npy")
        env = dict(os.environ, ARTALK_ATTN_WIDE=wide)
        subprocess.run([sys.executable, "-c", _WIDE_CHILD, root, f, str(B), str(H), "64", str(Lq), str(Lk), str(split), mode], check=True, env=env,
                       timeout=300)
        outs.append(np.load(f))
    assert np.isfinite(outs[0]).all()
    assert np.array_equal(outs[0], outs[1])


def test_w2v_front():
    """audio normalisation + conv0 + LN + GELU vs torch (reference ops: wav2vec.py:22-27, hf:291-299)."""
    capi, L = _lib()
    g = torch.Generator().manual_seed(3)
    Cn, n = 3, 64000
    audio = torch.randn(Cn, n, generator=g) * 0.1
    audio[2] = 0.0                                     # an all-zero padding chunk (std = 0 -> divide by 1e-6)
    w = torch.randn(512, 1, 10, generator=g) / math.sqrt(10)
    b = torch.randn(512, generator=g) * 0.3
    lw = 1 + 0.1 * torch.randn(512, generator=g)
    lb = 0.1 * torch.randn(512, generator=g)
    xn = (audio - audio.mean(-1, keepdim=True)) / (audio.std(-1, keepdim=True) + 1e-6)
    h = F.conv1d(xn[:, None], w, b, stride=5).transpose(1, 2)
    ref = F.gelu(F.layer_norm(h, (512,), lw, lb, 1e-5))
    T = ref.shape[1]
    da, dw, db, dlw, dlb = _dev(audio), _dev(w.view(512, 10)), _dev(b), _dev(lw), _dev(lb)
    dxn = torch.empty(Cn, n, device="cuda")
    out = torch.empty(Cn, T, 512, device="cuda")
    assert L.artalk_op_w2v_front(_p(da), Cn, n, _p(dw), _p(db), _p(dlw), _p(dlb), _p(dxn), _p(out), None) == 0
    torch.cuda.synchronize()
    assert (dxn.cpu() - xn).abs().max().item() < 1e-5
    assert (out.cpu() - ref).abs().max().item() < 5e-5


def test_pool_silu():
    capi, L = _lib()
    g = torch.Generator().manual_seed(4)
    X = torch.randn(2, 199, 1024, generator=g)
    xt = X.permute(0, 2, 1)
    ref = F.silu(torch.cat([F.interpolate(xt, size=(p), mode="area").permute(0, 2, 1) for p in (1, 5, 25, 50, 100)], dim=1))
    dX = _dev(X)
    out = torch.empty(2, 181, 1024, device="cuda")
    assert L.artalk_op_pool_silu(_p(dX), 2, 199, 1024, _p(out), None) == 0
    torch.cuda.synchronize()
    assert (out.cpu() - ref).abs().max().item() < 2e-6


def test_bsq_history_matches_oracle():
    """Multi-scale BSQ + feature recurrences vs the oracle restatement of bitwise_vae.py:227-242,264-288,316-334."""
    from conftest import get_oracle
    capi, L = _lib()
    o = get_oracle("tiny")
    g = torch.Generator().manual_seed(5)
    B = 4
    enc = torch.randn(B, 100, 32, generator=g)
    bits_ref, margin = o.ms_bsq(enc)
    fdec_ref = o.vqidx_to_feat(bits_ref, False)
    ms_ref = o.vqidx_to_feat(bits_ref, True)
    denc = _dev(enc)
    bits = torch.zeros(B, 181, 32, dtype=torch.uint8, device="cuda")
    fdec = torch.empty(B, 100, 32, device="cuda")
    ms = torch.empty(B, 180, 32, device="cuda")
    assert L.artalk_op_bsq_history(_p(denc), _p(bits), _p(fdec), _p(ms), B, None) == 0
    torch.cuda.synchronize()
    mism = bits.cpu().int() != bits_ref.int()
    # a sign decision may only differ where the oracle's own margin is at rounding level
    assert (margin[mism] < 1e-5).all(), (int(mism.sum()), float(margin[mism].max()) if mism.any() else 0)
    if not mism.any():
        assert (fdec.cpu() - fdec_ref).abs().max().item() < 1e-6
        assert (ms.cpu() - ms_ref).abs().max().item() < 1e-6
