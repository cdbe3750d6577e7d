"""Stand-alone forward of the signal encoder's sub-modules on the per-op C ABI.

The reference's ``SEBlock`` / ``BasicBlock1D`` (multimodal_paper_modal_balance.py:49-64, 67-93) are ordinary modules a
caller may apply to a [B, C, L] tensor by hand (feature extraction, per-block inspection).  Training runs through the
encoder's launch plan (csrc/plan_resnet1d.hip), so these forwards build NO autograd graph: they are computed under
``torch.no_grad()`` with the same kernels the plan enqueues -- conv forward with BatchNorm partial sums in its epilogue,
finalize (batch statistics + running-statistics update in train mode, running statistics in eval mode), apply + ReLU,
global average pool through the BatchNorm affine, the SE MLP, and the gated residual tail.
"""
import ctypes as C

import torch

from . import functional as HF
from . import lib as L
from .functional import _require_cuda, f32c, ptr, stream

_DT = {"bf16": (L.BF16, torch.bfloat16), "bfloat16": (L.BF16, torch.bfloat16), "fp32": (L.F32, torch.float32),
       "float32": (L.F32, torch.float32), "f32": (L.F32, torch.float32)}


def _to_nlc(x, dt, tdt):
    """[B, C, L] fp32 -> channels-last [B, L, C] in the compute dtype"""
    B, Cn, Ln = x.shape
    out = torch.empty(B * Ln * Cn, device=x.device, dtype=tdt)
    L.check(L.lib().ecgmm_nchw_to_nhwc(dt, ptr(x), ptr(out), B, Cn, Ln, stream()), "nchw_to_nhwc")
    return out


def _to_ncl(t, dt, B, Cn, Ln):
    out = torch.empty(B, Cn, Ln, device=t.device, dtype=torch.float32)
    L.check(L.lib().ecgmm_nhwc_to_nchw(dt, ptr(t), ptr(out), B, Cn, Ln, stream()), "nhwc_to_nchw")
    return out


def _conv_bn(x_nlc, B, Lin, conv, bn, dt, tdt, training):
    """Conv1d (+bias) -> BatchNorm coefficients.  Returns (raw conv output [B, Lout, Cout], coef [4][Cout], Lout)."""
    lib = L.lib()
    Cout, Cin, K = conv.weight.shape
    d = L.ConvDesc(B, 1, Lin, Cin, Cout, 1, K, conv.stride, 0, conv.padding)
    Lout = (Lin + 2 * conv.padding - K) // conv.stride + 1
    M = B * Lout
    wf = torch.empty(conv.weight.numel(), device=x_nlc.device, dtype=tdt)
    L.check(lib.ecgmm_pack_conv_weight(dt, ptr(f32c(conv.weight.detach())), ptr(wf), None, Cout, Cin, K, stream()), "pack")
    y = torch.empty(M * Cout, device=x_nlc.device, dtype=tdt)
    coef = torch.empty(4, Cout, device=x_nlc.device, dtype=torch.float32)
    bias = None if conv.bias is None else ptr(f32c(conv.bias.detach()))
    if training:
        rows = lib.ecgmm_conv_stats_rows(M)
        stats = torch.empty((rows + 64) * 2 * Cout, device=x_nlc.device, dtype=torch.float32)
        L.check(lib.ecgmm_conv_fwd(dt, C.byref(d), ptr(x_nlc), ptr(wf), bias, ptr(y), ptr(stats), 0, stream()), "conv_fwd")
        L.check(lib.ecgmm_bn_finalize(ptr(stats), rows, Cout, float(M), ptr(bn.weight.detach()), ptr(bn.bias.detach()),
                                      ptr(bn.running_mean), ptr(bn.running_var), ptr(bn.num_batches_tracked), bn.momentum,
                                      bn.eps, ptr(coef), stream()), "bn_finalize")
    else:
        L.check(lib.ecgmm_conv_fwd(dt, C.byref(d), ptr(x_nlc), ptr(wf), bias, ptr(y), None, 0, stream()), "conv_fwd")
        L.check(lib.ecgmm_bn_eval_coef(Cout, ptr(bn.weight.detach()), ptr(bn.bias.detach()), ptr(bn.running_mean),
                                       ptr(bn.running_var), bn.eps, ptr(coef), stream()), "bn_eval_coef")
    return y, coef, Lout


def _se_gate(se, pooled):
    """[B, C] channel means -> sigmoid(W2 relu(W1 m + b1) + b2)"""
    h = HF.linear(pooled, se.fc[0].weight, se.fc[0].bias, L.ACT_RELU)
    return HF.linear(h, se.fc[2].weight, se.fc[2].bias, L.ACT_SIGMOID)


@torch.no_grad()
def se_block_forward(se, x):
    """SEBlock.forward (PMB:58-62): x * sigmoid(fc(mean_L x)), fp32."""
    _require_cuda(x, "SEBlock")
    x = f32c(x)
    B, Cn, Ln = x.shape
    lib = L.lib()
    xl = _to_nlc(x, L.F32, torch.float32)
    pooled = torch.empty(B, Cn, device=x.device, dtype=torch.float32)
    L.check(lib.ecgmm_avgpool(L.F32, ptr(xl), ptr(pooled), B, Ln, Cn, None, stream()), "avgpool")
    gate = _se_gate(se, pooled)
    ident = torch.zeros(4, Cn, device=x.device, dtype=torch.float32)
    ident[0] = 1.0                                        # scale 1, shift 0: the apply pass as a plain gated copy
    out = torch.empty_like(xl)
    L.check(lib.ecgmm_bn_act(L.F32, ptr(xl), ptr(ident), None, None, ptr(gate), Ln, 0, ptr(out), B * Ln, Cn, stream()), "bn_act")
    return _to_ncl(out, L.F32, B, Cn, Ln)


@torch.no_grad()
def basic_block1d_forward(blk, x, compute_dtype="fp32"):
    """BasicBlock1D.forward (PMB:86-93): relu(se(bn2(conv2(relu(bn1(conv1 x))))) + identity)."""
    _require_cuda(x, "BasicBlock1D")
    x = f32c(x)
    dt, tdt = _DT[str(compute_dtype).lower()]
    B, Cin, Lin = x.shape
    lib = L.lib()
    training = blk.training
    xl = _to_nlc(x, dt, tdt)
    y1, coef1, L1 = _conv_bn(xl, B, Lin, blk.conv1, blk.bn1, dt, tdt, training and blk.bn1.training)
    Cout = blk.conv1.weight.shape[0]
    M = B * L1
    a1 = torch.empty_like(y1)
    L.check(lib.ecgmm_bn_act(dt, ptr(y1), ptr(coef1), None, None, None, 1, 1, ptr(a1), M, Cout, stream()), "bn_act")
    y2, coef2, L2 = _conv_bn(a1, B, L1, blk.conv2, blk.bn2, dt, tdt, training and blk.bn2.training)
    pooled = torch.empty(B, Cout, device=x.device, dtype=torch.float32)
    L.check(lib.ecgmm_avgpool(dt, ptr(y2), ptr(pooled), B, L2, Cout, ptr(coef2), stream()), "avgpool")   # mean_L bn2(y2)
    gate = _se_gate(blk.se, pooled)
    if blk.downsample is not None:
        yd, coefd, Ld = _conv_bn(xl, B, Lin, blk.downsample[0], blk.downsample[1], dt, tdt,
                                 training and blk.downsample[1].training)
        res, rcoef = yd, coefd
    else:
        res, rcoef = xl, None
    out = torch.empty_like(y2)
    L.check(lib.ecgmm_bn_act(dt, ptr(y2), ptr(coef2), ptr(res), None if rcoef is None else ptr(rcoef), ptr(gate), L2, 1,
                             ptr(out), B * L2, Cout, stream()), "bn_act")
    return _to_ncl(out, dt, B, Cout, L2)
