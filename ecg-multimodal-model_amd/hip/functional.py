"""torch.autograd glue over the C ABI: every op here launches HIP kernels from libecgmm_hip.so on
torch's current stream.  PyTorch only supplies device memory, the stream and the autograd graph.

Parameter gradients are written by the kernels straight into ``param.grad`` ("gradient sinks", see
:func:`grad_sink`) and the autograd Functions return ``None`` for parameters: a backward pass
OVERWRITES ``.grad`` (like ``zero_grad()`` + ``backward()`` in the reference loop, train.py:65-80)
instead of accumulating into it.  That keeps gradients inside one flat buffer for the RCCL
all-reduce and the fused Adam.
"""
import ctypes as C

import torch

from . import lib as L

vp = C.c_void_p
# debug switch: validate class indices on the host before the loss kernels (costs a device sync per call; without it an
# out-of-range label makes the loss NaN instead of raising like torch's CrossEntropyLoss)
_CHECK_LABELS = __import__("os").environ.get("ECGMM_CHECK_LABELS", "0") == "1"


def _require_cuda(t, what):
    if not t.is_cuda:
        raise RuntimeError(f"{what}: tensor is on {t.device}; the HIP library is the only compute path "
                           "(no CPU fallback) -- move the model and inputs to a ROCm device")


def ptr(t):
    return None if t is None else vp(t.data_ptr())


def stream():
    return vp(torch.cuda.current_stream().cuda_stream)


def f32c(t):
    """contiguous fp32 view/copy of ``t`` (host-side plumbing only)."""
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def grad_sink(p, accumulate=False):
    """The tensor the kernels write ``p``'s gradient into (allocated on first use, or the view of a
    flat gradient buffer installed by ``parallel.flatten``).  None when ``p`` is frozen.

    A backward pass OVERWRITES the sink.  Writing the same sink twice before the optimizer has consumed
    it (``FusedAdam.step()`` / ``zero_grad()``, ``module.zero_grad()`` or :func:`release_grads`) would
    silently drop the first gradient -- gradient accumulation over several backward passes, or a
    parameter used twice in one forward -- so that raises instead.  ``accumulate=True`` is for callers
    that add into the sink themselves (TabNet's shared layers)."""
    if p is None or not p.requires_grad:
        return None
    if p.grad is None:
        view = getattr(p, "_ecg_grad_view", None)
        p.grad = view if view is not None else torch.zeros_like(p, memory_format=torch.contiguous_format)
        p._ecg_dirty = False
    if getattr(p, "_ecg_dirty", False) and not accumulate:
        raise RuntimeError(
            "gradient sink written twice: backward passes of this library OVERWRITE .grad (train.py:65-81 pattern: "
            "zero_grad -> backward -> step).  Call optimizer.zero_grad() / optimizer.step() (or "
            "ecgmm.hip.functional.release_grads(model)) between backward passes; gradient accumulation and "
            f"parameters shared by several layers are not supported (parameter shape {tuple(p.shape)})")
    p._ecg_dirty = True
    p._ecg_written = True
    return p.grad


def release_grads(params):
    """Mark the gradients of ``params`` (an iterable of parameters or a module) as consumed: the next backward
    may overwrite them.  What ``FusedAdam.zero_grad()`` / ``step()`` do for their own parameters."""
    if isinstance(params, torch.nn.Module):
        params = params.parameters()
    for p in params:
        p._ecg_dirty = False


def new_bytes(n, device):
    return torch.empty(max(int(n), 16), dtype=torch.uint8, device=device)


class _Scratch:
    """Grow-only per-device scratch buffers for transient workspaces (never freed mid-step, so the
    launches stay graph-capturable and allocator-quiet)."""
    _bufs = {}

    @classmethod
    def get(cls, key, nbytes, device):
        # one buffer per (use, device, STREAM): encoders may run concurrently on different streams
        k = (key, device.index, torch.cuda.current_stream(device).cuda_stream)
        b = cls._bufs.get(k)
        if b is None or b.numel() < nbytes:
            b = new_bytes(nbytes, device)
            cls._bufs[k] = b
        return b


# --------------------------------------------------------------------------------------------
# Linear (+bias, +activation)      reference: nn.Linear at PMB:221,257,261,269-271,284,288
# --------------------------------------------------------------------------------------------
class LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, act):
        _require_cuda(x, "linear")
        x = f32c(x)
        B, In = x.shape
        Out = weight.shape[0]
        y = torch.empty(B, Out, device=x.device, dtype=torch.float32)
        L.check(L.lib().ecgmm_linear_fwd(ptr(x), ptr(weight), ptr(bias), ptr(y), B, In, Out, act, stream()), "linear_fwd")
        ctx.act = act
        ctx.params = (weight, bias)
        ctx.save_for_backward(x, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        weight, bias = ctx.params
        B, In = x.shape
        Out = weight.shape[0]
        dy = f32c(dy)
        if ctx.act != L.ACT_NONE:
            dz = torch.empty_like(dy)
            L.check(L.lib().ecgmm_act_bwd(ptr(dy), ptr(y), ptr(dz), dy.numel(), ctx.act, stream()), "act_bwd")
        else:
            dz = dy
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        # sinks only for the gradients this backward was asked for (torch.autograd.grad w.r.t. inputs leaves .grad alone)
        dw = grad_sink(weight) if ctx.needs_input_grad[1] else None
        db = grad_sink(bias) if ctx.needs_input_grad[2] else None
        nb = L.lib().ecgmm_linear_bwd_scratch(B, In, Out)
        scratch = _Scratch.get("linear", nb, x.device)
        L.check(L.lib().ecgmm_linear_bwd(ptr(dz), ptr(x), ptr(weight), ptr(dx), ptr(dw), ptr(db), B, In, Out,
                                         ptr(scratch), scratch.numel(), stream()), "linear_bwd")
        return dx, None, None, None


def linear(x, weight, bias=None, act=L.ACT_NONE):
    return LinearFn.apply(x, weight, bias, act)


# --------------------------------------------------------------------------------------------
# LayerNorm / AttentionFusion        reference: PMB:31-46, 223, 239, 263
# --------------------------------------------------------------------------------------------
def _seg_arrays(segs):
    n = len(segs)
    arr = (vp * 3)(*[vp(s.data_ptr()) for s in segs] + [None] * (3 - n))
    dims = (C.c_int * 3)(*[s.shape[1] for s in segs] + [0] * (3 - n))
    return arr, dims


class LayerNormFn(torch.autograd.Function):
    """nseg == 1: plain LayerNorm.  nseg == 3 with fusion weights: softmax(w) * feats -> cat -> LN."""

    @staticmethod
    def forward(ctx, gamma, beta, fusion_w, eps, *segs):
        for s in segs:
            _require_cuda(s, "layernorm")
        segs = tuple(f32c(s) for s in segs)
        B = segs[0].shape[0]
        D = sum(s.shape[1] for s in segs)
        dev = segs[0].device
        out = torch.empty(B, D, device=dev, dtype=torch.float32)
        stat = torch.empty(B, 2, device=dev, dtype=torch.float32)
        soft = torch.empty(3, device=dev, dtype=torch.float32) if fusion_w is not None else None
        arr, dims = _seg_arrays(segs)
        L.check(L.lib().ecgmm_layernorm_fwd(arr, dims, len(segs), ptr(fusion_w), ptr(gamma), ptr(beta), ptr(out),
                                            ptr(stat), ptr(soft), B, eps, stream()), "layernorm_fwd")
        ctx.params = (gamma, beta, fusion_w)
        ctx.nseg = len(segs)
        ctx.save_for_backward(stat, *segs)
        if fusion_w is not None:
            ctx.mark_non_differentiable(soft)
            return out, soft
        return out

    @staticmethod
    def backward(ctx, dout, *unused):
        stat, *segs = ctx.saved_tensors
        gamma, beta, fusion_w = ctx.params
        B = segs[0].shape[0]
        D = sum(s.shape[1] for s in segs)
        dout = f32c(dout)
        dsegs = [torch.empty_like(s) if ctx.needs_input_grad[4 + i] else None for i, s in enumerate(segs)]
        arr, dims = _seg_arrays(segs)
        darr = (vp * 3)(*[ptr(d) for d in dsegs] + [None] * (3 - len(segs)))
        scratch = _Scratch.get("ln", L.lib().ecgmm_layernorm_bwd_scratch(B, D), dout.device)
        L.check(L.lib().ecgmm_layernorm_bwd(arr, dims, len(segs), ptr(fusion_w), ptr(gamma), ptr(stat), ptr(dout),
                                            darr, 0, ptr(grad_sink(gamma) if ctx.needs_input_grad[0] else None),
                                            ptr(grad_sink(beta) if ctx.needs_input_grad[1] else None),
                                            ptr(grad_sink(fusion_w) if ctx.needs_input_grad[2] else None), B, ptr(scratch),
                                            stream()), "layernorm_bwd")
        return (None, None, None, None, *dsegs)


def layer_norm(x, gamma, beta, eps=1e-5):
    return LayerNormFn.apply(gamma, beta, None, eps, x)


def attention_fusion(img, sig, clin, weights, gamma, beta, eps=1e-5):
    return LayerNormFn.apply(gamma, beta, weights, eps, img, sig, clin)


# --------------------------------------------------------------------------------------------
# var_loss                             reference: PMB:349-352
# --------------------------------------------------------------------------------------------
class VarLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, f0, f1, f2):
        fs = tuple(f32c(f) for f in (f0, f1, f2))
        _require_cuda(fs[0], "var_loss")
        B = fs[0].shape[0]
        dev = fs[0].device
        loss = torch.empty((), device=dev, dtype=torch.float32)
        scratch = torch.empty(3 * B + 4, device=dev, dtype=torch.float32)
        L.check(L.lib().ecgmm_varloss_fwd(ptr(fs[0]), ptr(fs[1]), ptr(fs[2]), B, fs[0].shape[1], fs[1].shape[1],
                                          fs[2].shape[1], ptr(loss), ptr(scratch), stream()), "varloss_fwd")
        ctx.save_for_backward(scratch, *fs)
        return loss

    @staticmethod
    def backward(ctx, gout):
        scratch, *fs = ctx.saved_tensors
        gout = f32c(gout).reshape(1)
        outs = []
        for m, f in enumerate(fs):
            if not ctx.needs_input_grad[m]:
                outs.append(None)
                continue
            df = torch.empty_like(f)
            L.check(L.lib().ecgmm_varloss_bwd(ptr(f), f.shape[0], f.shape[1], ptr(gout), ptr(scratch), m, ptr(df), 0,
                                              stream()), "varloss_bwd")
            outs.append(df)
        return tuple(outs)


def var_loss(f0, f1, f2):
    return VarLossFn.apply(f0, f1, f2)


# --------------------------------------------------------------------------------------------
# CrossEntropy / Focal                reference: train.py:31,69-72; signal_model.py:91-106
# --------------------------------------------------------------------------------------------
class CELossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, focal, alpha, gamma):
        _require_cuda(logits, "cross_entropy")
        logits = f32c(logits)
        labels = labels.to(torch.int64).contiguous()
        B, Cn = logits.shape
        if labels.shape != (B,):
            raise ValueError(f"cross_entropy: expected {B} class indices, got shape {tuple(labels.shape)}")
        if _CHECK_LABELS and B > 0 and (int(labels.min()) < 0 or int(labels.max()) >= Cn):
            raise IndexError(f"Target {int(labels.max() if labels.max() >= Cn else labels.min())} is out of bounds.")
        loss = torch.empty((), device=logits.device, dtype=torch.float32)
        dcoef = torch.empty(B, device=logits.device, dtype=torch.float32)
        L.check(L.lib().ecgmm_ce_fwd(ptr(logits), ptr(labels), B, Cn, int(focal), float(alpha), float(gamma),
                                     ptr(loss), ptr(dcoef), stream()), "ce_fwd")
        ctx.save_for_backward(logits, labels, dcoef)
        return loss

    @staticmethod
    def backward(ctx, gout):
        logits, labels, dcoef = ctx.saved_tensors
        B, Cn = logits.shape
        d = torch.empty_like(logits)
        gout = f32c(gout).reshape(1)
        L.check(L.lib().ecgmm_ce_bwd(ptr(logits), ptr(labels), B, Cn, ptr(dcoef), ptr(gout), ptr(d), stream()), "ce_bwd")
        return d, None, None, None, None


class CEPlusFn(torch.autograd.Function):
    """CrossEntropy(logits, labels) + w * extra  (the step loss of train.py:69-78: CE(fusion_logits) + 0.1 * var_loss)"""

    @staticmethod
    def forward(ctx, logits, labels, extra, w):
        _require_cuda(logits, "cross_entropy")
        logits, extra = f32c(logits), f32c(extra).reshape(1)
        labels = labels.to(torch.int64).contiguous()
        B, Cn = logits.shape
        if labels.shape != (B,):
            raise ValueError(f"cross_entropy: expected {B} class indices, got shape {tuple(labels.shape)}")
        if _CHECK_LABELS and B > 0 and (int(labels.min()) < 0 or int(labels.max()) >= Cn):
            raise IndexError("Target is out of bounds.")
        loss = torch.empty((), device=logits.device, dtype=torch.float32)
        dcoef = torch.empty(B, device=logits.device, dtype=torch.float32)
        L.check(L.lib().ecgmm_ce_plus_fwd(ptr(logits), ptr(labels), B, Cn, ptr(extra), float(w), ptr(loss), ptr(dcoef),
                                          stream()), "ce_plus_fwd")
        ctx.w = float(w)
        ctx.save_for_backward(logits, labels, dcoef)
        return loss

    @staticmethod
    def backward(ctx, gout):
        logits, labels, dcoef = ctx.saved_tensors
        B, Cn = logits.shape
        d = torch.empty_like(logits)
        dextra = torch.empty((), device=logits.device, dtype=torch.float32)
        gout = f32c(gout).reshape(1)
        L.check(L.lib().ecgmm_ce_plus_bwd(ptr(logits), ptr(labels), B, Cn, ptr(dcoef), ptr(gout), ptr(d), ptr(dextra),
                                          ctx.w, stream()), "ce_plus_bwd")
        return d, None, dextra, None


def cross_entropy(logits, labels):
    return CELossFn.apply(logits, labels, False, 1.0, 0.0)


def cross_entropy_plus(logits, labels, extra, weight):
    """CrossEntropy(logits, labels) + weight * extra in one kernel per direction"""
    return CEPlusFn.apply(logits, labels, extra, weight)


def focal_loss(logits, labels, alpha=1.0, gamma=2.0):
    return CELossFn.apply(logits, labels, True, alpha, gamma)


# --------------------------------------------------------------------------------------------
# Dropout                              reference: nn.Dropout(0.3) at PMB:115,260,287
# --------------------------------------------------------------------------------------------
class _PhiloxState:
    """Seed/offset bookkeeping for the device Philox stream (advanced per call, like torch's)."""
    seed = 42
    offset = 0

    @classmethod
    def manual_seed(cls, seed):
        cls.seed, cls.offset = int(seed), 0

    @classmethod
    def take(cls, n):
        off = cls.offset
        cls.offset += (int(n) + 3) // 4
        return cls.seed, off


def manual_seed(seed):
    _PhiloxState.manual_seed(seed)


class DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p):
        _require_cuda(x, "dropout")
        x = f32c(x)
        y = torch.empty_like(x)
        mask = torch.empty(x.numel(), dtype=torch.uint8, device=x.device)
        seed, off = _PhiloxState.take(x.numel())
        L.check(L.lib().ecgmm_dropout_fwd(ptr(x), ptr(y), ptr(mask), x.numel(), p, seed, off, stream()), "dropout_fwd")
        ctx.p = p
        ctx.save_for_backward(mask)
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        dy = f32c(dy)
        dx = torch.empty_like(dy)
        L.check(L.lib().ecgmm_dropout_bwd(ptr(dy), ptr(mask), ptr(dx), dy.numel(), ctx.p, stream()), "dropout_bwd")
        return dx, None


def dropout(x, p, training):
    if not training or p <= 0.0:
        return x
    return DropoutFn.apply(x, float(p))


# --------------------------------------------------------------------------------------------
# BatchNorm1d over [B, C] fp32 (+ReLU)   reference: clinical_encoder[1:3], PMB:258-259
# --------------------------------------------------------------------------------------------
class BatchNorm1dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, rm, rv, nbt, training, momentum, eps, relu):
        _require_cuda(x, "batchnorm1d")
        x = f32c(x)
        B, Cn = x.shape
        dev = x.device
        lib = L.lib()
        coef = torch.empty(4, Cn, device=dev, dtype=torch.float32)
        if training:
            if B < 2:
                raise ValueError("Expected more than 1 value per channel when training")  # torch's message
            rows = lib.ecgmm_col_stats_rows(L.F32, B, Cn)
            partial = torch.empty(rows + 64, 2, Cn, device=dev, dtype=torch.float32)  # + tail rows (ecgmm.h)
            L.check(lib.ecgmm_col_stats(L.F32, ptr(x), B, Cn, ptr(partial), stream()), "col_stats")
            L.check(lib.ecgmm_bn_finalize(ptr(partial), rows, Cn, float(B), ptr(gamma), ptr(beta), ptr(rm), ptr(rv),
                                          ptr(nbt), momentum, eps, ptr(coef), stream()), "bn_finalize")
        else:
            L.check(lib.ecgmm_bn_eval_coef(Cn, ptr(gamma), ptr(beta), ptr(rm), ptr(rv), eps, ptr(coef), stream()),
                    "bn_eval_coef")
        y = torch.empty_like(x)
        L.check(lib.ecgmm_bn_act(L.F32, ptr(x), ptr(coef), None, None, None, 1, int(relu), ptr(y), B, Cn, stream()),
                "bn_act")
        ctx.params = (gamma, beta)
        ctx.relu, ctx.training = relu, training
        ctx.save_for_backward(x, y, coef)
        return y

    @staticmethod
    def backward(ctx, dy):
        if not ctx.training:
            raise RuntimeError("BatchNorm1d backward needs the forward to have run in training mode")
        x, y, coef = ctx.saved_tensors
        gamma, beta = ctx.params
        B, Cn = x.shape
        dy = f32c(dy)
        lib = L.lib()
        dx = torch.empty_like(x)
        scratch = _Scratch.get("bn1d", lib.ecgmm_bn_bwd_scratch(L.F32, B, Cn), x.device)
        L.check(lib.ecgmm_bn_bwd(L.F32, ptr(dy), ptr(y) if ctx.relu else None, None, None, 1, ptr(x), ptr(coef),
                                 ptr(gamma), ptr(grad_sink(gamma) if ctx.needs_input_grad[1] else None),
                                 ptr(grad_sink(beta) if ctx.needs_input_grad[2] else None), ptr(dx), None, None, B, Cn,
                                 ptr(scratch), stream()), "bn_bwd")
        return dx, None, None, None, None, None, None, None, None, None


def batch_norm1d(x, gamma, beta, rm, rv, nbt, training, momentum=0.1, eps=1e-5, relu=False):
    return BatchNorm1dFn.apply(x, gamma, beta, rm, rv, nbt, training, momentum, eps, relu)
