"""CPU-side checks of the C-ABI boundary: the header, the ctypes table and the built library agree."""
import ctypes
import os
import re

import pytest

from ecgmm.hip import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "ecgmm.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ecgmm_[a-z0-9_]+)\s*\(", src)))


def test_header_and_ctypes_table_agree():
    names = header_functions()
    assert len(names) >= 50
    assert sorted(L.SIGNATURES) == names


def test_library_exports_every_declared_symbol():
    assert os.path.exists(L.LIB_PATH), "libecgmm_hip.so not built (python __graft_entry__.py)"
    h = ctypes.CDLL(L.LIB_PATH)
    for name in header_functions():
        assert hasattr(h, name), f"{name} declared in include/ecgmm.h but not exported"
    assert L.lib().ecgmm_version() == 100


def test_shape_queries_run_without_a_gpu():
    lib = L.lib()
    d = L.ResNet18Desc(8, 224, 224, 256, L.BF16, 1, 0.1, 1e-5)
    fwd, bwd = lib.ecgmm_resnet18_fwd_workspace(ctypes.byref(d)), lib.ecgmm_resnet18_bwd_workspace(ctypes.byref(d))
    assert fwd > 8 * 112 * 112 * 64 * 2 and bwd > 0
    s = L.ResNet1DDesc(8, 1, 5000, 256, L.F32, 1, 0.1, 1e-5, 0.3, 1, 0)
    assert lib.ecgmm_resnet1d_fwd_workspace(ctypes.byref(s)) > 8 * 2500 * 64 * 4
    bad = L.ResNet18Desc(8, 224, 224, 256, 7, 1, 0.1, 1e-5)
    assert lib.ecgmm_resnet18_fwd_workspace(ctypes.byref(bad)) == 0
    assert b"dtype" in lib.ecgmm_last_error()
    assert lib.ecgmm_conv_stats_rows(1000) == 16
    c = L.ConvDesc(8, 56, 56, 64, 64, 3, 3, 1, 1, 1)
    assert lib.ecgmm_conv_bwd_weight_workspace(L.BF16, ctypes.byref(c)) >= 9 * 64 * 64 * 4


def test_product_path_refuses_cpu_tensors():
    import torch
    from ecgmm.hip import functional as HF
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        HF.linear(torch.zeros(2, 4), torch.zeros(3, 4))
