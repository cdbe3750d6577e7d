"""Helpers shared by the GPU parity tests: call the C ABI with torch tensors, convert layouts."""
import ctypes as C

import numpy as np
import torch

from ecgmm.hip import lib as L
from ecgmm.hip.functional import ptr, stream

DEV = "cuda:0"
TDT = {L.F32: torch.float32, L.BF16: torch.bfloat16}


def dev(t):
    return t.to(DEV).contiguous()


def to_nhwc(x_nchw_cpu, dt):
    """reference layout (NCHW fp32, CPU) -> channels-last compute dtype on the GPU, via the library's own kernel"""
    x = dev(x_nchw_cpu.float())
    N, Cn = x.shape[0], x.shape[1]
    hw = int(np.prod(x.shape[2:]))
    out = torch.empty(N * hw * Cn, device=DEV, dtype=TDT[dt])
    L.check(L.lib().ecgmm_nchw_to_nhwc(dt, ptr(x), ptr(out), N, Cn, hw, stream()))
    return out


def from_nhwc(t, dt, shape_nchw):
    N, Cn = shape_nchw[0], shape_nchw[1]
    hw = int(np.prod(shape_nchw[2:]))
    out = torch.empty(shape_nchw, device=DEV, dtype=torch.float32)
    L.check(L.lib().ecgmm_nhwc_to_nchw(dt, ptr(t), ptr(out), N, Cn, hw, stream()))
    torch.cuda.synchronize()
    return out.cpu()


def conv_desc(N, H, W, Cin, Cout, R, S, stride, ph, pw):
    return L.ConvDesc(N, H, W, Cin, Cout, R, S, stride, ph, pw)


def pack_weight(w_oihw_cpu, dt):
    w = dev(w_oihw_cpu.float())
    Cout, Cin = w.shape[0], w.shape[1]
    RS = int(np.prod(w.shape[2:]))
    f = torch.empty(w.numel(), device=DEV, dtype=TDT[dt])
    d = torch.empty(w.numel(), device=DEV, dtype=TDT[dt])
    L.check(L.lib().ecgmm_pack_conv_weight(dt, ptr(w), ptr(f), ptr(d), Cout, Cin, RS, stream()))
    return f, d


def rel_err(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def bf16_round(t):
    return t.to(torch.bfloat16).float()
