"""The C-ABI library loads without a GPU and exports every symbol include/artalk_hip.h declares."""
import os
import re

from artalk_amd import capi

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(REPO, "include", "artalk_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(artalk_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    L = capi.lib()
    declared = _declared()
    assert len(declared) >= 19
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/artalk_hip.h but not exported"
    assert sorted(capi.SYMBOLS) == declared


def test_config_struct_matches_reference_config():
    from artalk_amd.config import ARTalkConfig
    s = capi.config_struct(ARTalkConfig.full())
    assert (s.ar_depth, s.ar_heads, s.vae_depth, s.vae_hidden, s.code_dim, s.motion_dim) == (12, 12, 8, 512, 32, 106)
    assert list(s.patch_nums)[:5] == [1, 5, 25, 50, 100]
    assert list(s.w2v_conv_kernel)[:7] == [10, 3, 3, 3, 3, 2, 2] and s.w2v_layers == 24


def test_argument_errors_without_gpu():
    """Entry points validate arguments before touching the device."""
    L = capi.lib()
    assert L.artalk_create(0, None, None) == capi.EINVAL
    assert L.artalk_op_gemm(None, 0, None, None, None, None, None, 1, 1, 32, 0, None) == capi.EINVAL
    assert L.artalk_set_profiling(None, 1) == capi.EINVAL
    assert L.artalk_last_error(None) is not None


def test_product_path_has_no_cpu_fallback():
    """Nothing under artalk_amd/ imports the oracle; the model refuses a CPU device."""
    import pytest
    pkg = os.path.join(REPO, "artalk_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "artalk_oracle" not in src and "import oracle" not in src, f
    from artalk_amd.model import BitwiseARModel
    from artalk_amd.config import ARTalkConfig
    with pytest.raises(RuntimeError):
        BitwiseARModel(ARTalkConfig.tiny()).to("cpu")
