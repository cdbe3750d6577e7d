"""Encoder-level autograd Functions: one C-ABI call runs a whole ResNet18 / ResNet1D_SE forward (or a
range of backward stages) as a native launch plan (csrc/plan_resnet18.hip, plan_resnet1d.hip)."""
import ctypes as C
import os

import torch

from . import lib as L
from .functional import _Scratch, _require_cuda, f32c, grad_sink, new_bytes, ptr, stream, vp, _PhiloxState

DTYPES = {"bf16": L.BF16, "bfloat16": L.BF16, "fp32": L.F32, "float32": L.F32, "f32": L.F32}


def dtype_code(name):
    try:
        return DTYPES[str(name).lower()]
    except KeyError:
        raise ValueError(f"compute dtype must be 'bf16' or 'fp32', got {name!r}")


def _table(tensors):
    return (vp * len(tensors))(*[None if t is None else vp(t.data_ptr()) for t in tensors])


class _PlanFn(torch.autograd.Function):
    """Shared forward/backward driver.  ``spec`` supplies the C entry points and the descriptor."""

    @staticmethod
    def forward(ctx, x, spec, *params):
        _require_cuda(x, spec.name)
        for p in params:
            _require_cuda(p, spec.name + " parameter")
        x = f32c(x)
        lib = L.lib()
        desc = spec.make_desc(x)
        ws_bytes = getattr(lib, spec.prefix + "_fwd_workspace")(C.byref(desc))
        if ws_bytes == 0:
            L.check(1, spec.name + " workspace query")
        ws = new_bytes(ws_bytes, x.device)
        feat = torch.empty(x.shape[0], spec.out_dim, device=x.device, dtype=torch.float32)
        ptab = _table(params)
        btab = _table(spec.buffers)
        L.check(getattr(lib, spec.prefix + "_forward")(C.byref(desc), ptr(x), ptab, btab, ptr(feat), ptr(ws),
                                                        ws.numel(), stream()), spec.name + " forward")
        ctx.spec, ctx.desc, ctx.ws, ctx.x, ctx.params = spec, desc, ws, x, params
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        spec, desc = ctx.spec, ctx.desc
        if not desc.training:
            raise RuntimeError(f"{spec.name}: backward through an eval-mode forward is not supported")
        lib = L.lib()
        dfeat = f32c(dfeat)
        ptab = _table(ctx.params)
        gtab = _table([grad_sink(p) if ctx.needs_input_grad[2 + i] else None for i, p in enumerate(ctx.params)])
        nb = getattr(lib, spec.prefix + "_bwd_workspace")(C.byref(desc))
        bws = _Scratch.get(spec.prefix + "_bwd", nb, dfeat.device)
        fn = getattr(lib, spec.prefix + "_backward")
        stages = spec.stage_groups or [(0, spec.n_stages)]
        # with a stage hook (data-parallel overlap) only the last group joins the weight-gradient side stream to
        # this stream; the hook orders its all-reduce after the side stream itself (ecgmm_side_wait)
        defer = spec.stage_hook is not None and len(stages) > 1 and hasattr(lib, "ecgmm_side_defer_join") \
            and spec.prefix == "ecgmm_resnet18" and os.environ.get("ECGMM_DDP_DEFER_JOIN", "1") != "0"
        try:
            for gi, (b, e) in enumerate(stages):
                if defer:
                    lib.ecgmm_side_defer_join(int(gi + 1 < len(stages)))
                L.check(fn(C.byref(desc), ptr(ctx.x), ptr(dfeat), ptab, gtab, ptr(ctx.ws), ptr(bws), bws.numel(), b, e,
                           stream()), spec.name + " backward")
                if spec.stage_hook is not None:
                    spec.stage_hook(spec, gi)
        finally:
            if defer:
                lib.ecgmm_side_defer_join(0)
        ctx.ws = None
        return (None, None) + (None,) * len(ctx.params)


class PlanSpec:
    """Mutable per-module launch description (rebuilt cheaply every forward)."""
    stage_groups = None   # list of (begin, end) backward stage ranges (set by parallel.py for overlap)
    stage_hook = None     # callable(spec, group_index) fired after each group

    def __init__(self, name, prefix, n_stages):
        self.name, self.prefix, self.n_stages = name, prefix, n_stages
        self.buffers = []
        self.out_dim = 0


class ResNet18Spec(PlanSpec):
    def __init__(self):
        super().__init__("resnet18", "ecgmm_resnet18", 10)
        self.dtype, self.training, self.momentum, self.eps = L.BF16, True, 0.1, 1e-5

    def make_desc(self, x):
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError(f"image encoder expects [B,3,H,W], got {tuple(x.shape)}")
        return L.ResNet18Desc(x.shape[0], x.shape[2], x.shape[3], self.out_dim, self.dtype, int(self.training),
                              self.momentum, self.eps)


class ResNet1DSpec(PlanSpec):
    def __init__(self):
        super().__init__("resnet1d_se", "ecgmm_resnet1d", 5)
        self.dtype, self.training, self.momentum, self.eps = L.BF16, True, 0.1, 1e-5
        self.dropout_p, self.cin = 0.3, 1

    def make_desc(self, x):
        if x.dim() != 3 or x.shape[1] != self.cin:
            raise ValueError(f"signal encoder expects [B,{self.cin},L], got {tuple(x.shape)}")
        p = self.dropout_p if self.training else 0.0
        seed, off = _PhiloxState.take(x.shape[0] * 64) if p > 0 else (0, 0)
        return L.ResNet1DDesc(x.shape[0], self.cin, x.shape[2], self.out_dim, self.dtype, int(self.training),
                              self.momentum, self.eps, p, seed, off)


def run_plan(x, spec, params):
    return _PlanFn.apply(x, spec, *params)


# --------------------------------------------------------------------------------------------
# The multimodal head (csrc/plan_head.hip): LayerNorms, branch heads, AttentionFusion, fusion classifier, var_loss
# as ONE native call per direction.      reference: PMB:326-354, multimodal.py:440-469
# --------------------------------------------------------------------------------------------
class HeadFn(torch.autograd.Function):
    """inputs: the three encoder outputs, a HeadSpec, the 19 head parameters (order of include/ecgmm.h).
    outputs: img_logits, signal_logits, clinical_logits, fusion_logits, var_loss, soft_weights"""

    @staticmethod
    def forward(ctx, img_raw, sig_raw, clin_raw, spec, *params):
        raws = tuple(f32c(t) for t in (img_raw, sig_raw, clin_raw))
        for t in raws + tuple(params):
            _require_cuda(t, "multimodal head")
        if len(params) != 19:
            raise RuntimeError(f"multimodal head expects 19 parameters, got {len(params)}")
        B = raws[0].shape[0]
        dims = [t.shape[1] for t in raws]
        drop = spec.training and spec.dropout_p > 0
        seed, off = _PhiloxState.take(B * spec.hidden) if drop else (0, 0)
        desc = L.HeadDesc(B, (C.c_int * 3)(*dims), spec.hidden, spec.num_classes, int(spec.training), spec.eps,
                          spec.dropout_p if drop else 0.0, seed, off)
        lib = L.lib()
        nb = lib.ecgmm_head_fwd_workspace(C.byref(desc))
        if nb == 0:
            L.check(1, "multimodal head workspace query")
        dev = raws[0].device
        ws = new_bytes(nb, dev)
        logits = [torch.empty(B, spec.num_classes, device=dev, dtype=torch.float32) for _ in range(4)]
        var = torch.empty((), device=dev, dtype=torch.float32)
        soft = torch.empty(3, device=dev, dtype=torch.float32)
        L.check(lib.ecgmm_head_forward(C.byref(desc), _table(raws), _table(params), _table(logits), ptr(var), ptr(soft),
                                       ptr(ws), ws.numel(), stream()), "multimodal head forward")
        ctx.desc, ctx.ws, ctx.raws, ctx.params = desc, ws, raws, params
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(soft)
        return (*logits, var, soft)

    @staticmethod
    def backward(ctx, g_img, g_sig, g_clin, g_fus, g_var, _g_soft):
        desc, params = ctx.desc, ctx.params
        lib = L.lib()
        dlog = [None if g is None else f32c(g) for g in (g_img, g_sig, g_clin, g_fus)]
        dvar = None if g_var is None else f32c(g_var).reshape(1)
        fus = dlog[3] is not None
        reach = [fus or dvar is not None or dlog[m] is not None for m in range(3)]   # any gradient reaches feat_m?
        draw = [torch.empty_like(ctx.raws[m]) if ctx.needs_input_grad[m] else None for m in range(3)]
        # a parameter gets a gradient only if its branch is part of the loss (train.py:78 leaves the three branch heads
        # out: torch then leaves their .grad None, and so does this -- the sinks of skipped branches are not touched)
        active = [reach[0], reach[0], reach[1], reach[1], reach[2], reach[2]]
        active += [dlog[0] is not None] * 2 + [dlog[1] is not None] * 2 + [dlog[2] is not None] * 2
        active += [fus] * 7
        sinks = [grad_sink(p) if (a and ctx.needs_input_grad[4 + i]) else None
                 for i, (p, a) in enumerate(zip(params, active))]
        nb = lib.ecgmm_head_bwd_workspace(C.byref(desc))
        bws = _Scratch.get("head_bwd", nb, ctx.raws[0].device)
        L.check(lib.ecgmm_head_backward(C.byref(desc), _table(ctx.raws), _table(params), _table(sinks), _table(dlog),
                                        ptr(dvar), _table(draw), ptr(ctx.ws), ptr(bws), bws.numel(), stream()),
                "multimodal head backward")
        ctx.ws = None
        return (*draw, None) + (None,) * len(params)


class HeadSpec:
    def __init__(self, hidden, num_classes, training, eps, dropout_p):
        self.hidden, self.num_classes, self.training, self.eps, self.dropout_p = hidden, num_classes, training, eps, dropout_p


def run_head(img_raw, sig_raw, clin_raw, spec, params):
    return HeadFn.apply(img_raw, sig_raw, clin_raw, spec, *params)
