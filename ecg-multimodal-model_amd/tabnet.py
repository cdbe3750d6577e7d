"""TabNet clinical encoder on HIP kernels (reference: multimodal.py:109-148, which wraps
``pytorch_tabnet.tab_network.TabNetNoEmbeddings`` -- a third-party package that is not vendored in the reference;
restated from its published algorithm, module / parameter names kept so that a checkpoint of the library loads).

    ClinicalTabNetEncoder(input_dim=2, latent_dim=32)  ->  forward(x [B, 2]) = (z [B, 32], M_loss)

Linear layers use the library's dense kernels, BatchNorm / Ghost-BatchNorm its column-stats + finalize + apply
kernels (one virtual batch = one row slice, no concatenation), and the TabNet-specific pieces are the row kernels of
csrc/tabnet.hip: GLU gate, sparsemax, the mask / prior recurrence, column split (+ReLU), mask entropy.
Every arithmetic step is a HIP kernel with a hand-written backward (torch.autograd.Function); there is no CPU path.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from .hip import functional as HF
from .hip import lib as L
from .hip import nn as hnn
from .hip.functional import _Scratch, _require_cuda, f32c, grad_sink, ptr, stream

EW = dict(MUL=0, ADD_SCALE=1, PRIOR=2, RELU=3, RELU_BWD=4, SCALE=5, NEG_MUL=6, ADD=7, RSUB=8)


def _ew(op, a, b=None, s=0.0):
    out = torch.empty_like(a)
    L.check(L.lib().ecgmm_ew(EW[op], ptr(a), ptr(b), ptr(out), a.numel(), float(s), stream()), "ew")
    return out


class _Mul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = f32c(a), f32c(b)
        ctx.save_for_backward(a, b)
        return _ew("MUL", a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = f32c(g)
        return _ew("MUL", g, b), _ew("MUL", g, a)


class _AddScale(torch.autograd.Function):
    """(a + b) * s"""

    @staticmethod
    def forward(ctx, a, b, s):
        ctx.s = s
        return _ew("ADD_SCALE", f32c(a), f32c(b), s)

    @staticmethod
    def backward(ctx, g):
        gs = _ew("SCALE", f32c(g), None, ctx.s) if ctx.s != 1.0 else g
        return gs, gs, None


class _Scale(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, s):
        ctx.s = s
        return _ew("SCALE", f32c(a), None, s)

    @staticmethod
    def backward(ctx, g):
        return _ew("SCALE", f32c(g), None, ctx.s), None


class _PriorUpdate(torch.autograd.Function):
    """prior * (gamma - M); prior = None stands for the all-ones prior of the first step"""

    @staticmethod
    def forward(ctx, M, prior, gamma):
        M = f32c(M)
        ctx.gamma, ctx.has_prior = gamma, prior is not None
        if prior is None:
            return _ew("RSUB", M, None, gamma)
        prior = f32c(prior)
        ctx.save_for_backward(M, prior)
        return _ew("PRIOR", M, prior, gamma)

    @staticmethod
    def backward(ctx, g):
        g = f32c(g)
        if not ctx.has_prior:
            return _ew("SCALE", g, None, -1.0), None, None
        M, prior = ctx.saved_tensors
        return _ew("NEG_MUL", g, prior), _ew("PRIOR", M, g, ctx.gamma), None


class _GLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z):
        _require_cuda(z, "glu")
        z = f32c(z)
        N, D2 = z.shape
        out = torch.empty(N, D2 // 2, device=z.device, dtype=torch.float32)
        L.check(L.lib().ecgmm_glu_fwd(ptr(z), ptr(out), N, D2 // 2, stream()), "glu_fwd")
        ctx.save_for_backward(z)
        return out

    @staticmethod
    def backward(ctx, g):
        z, = ctx.saved_tensors
        dz = torch.empty_like(z)
        L.check(L.lib().ecgmm_glu_bwd(ptr(z), ptr(f32c(g)), ptr(dz), z.shape[0], z.shape[1] // 2, stream()), "glu_bwd")
        return dz


class _Sparsemax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _require_cuda(x, "sparsemax")
        x = f32c(x)
        p = torch.empty_like(x)
        L.check(L.lib().ecgmm_sparsemax_fwd(ptr(x), ptr(p), x.shape[0], x.shape[1], stream()), "sparsemax_fwd")
        ctx.save_for_backward(p)
        return p

    @staticmethod
    def backward(ctx, g):
        p, = ctx.saved_tensors
        dx = torch.empty_like(p)
        L.check(L.lib().ecgmm_sparsemax_bwd(ptr(p), ptr(f32c(g)), ptr(dx), p.shape[0], p.shape[1], stream()), "sparsemax_bwd")
        return dx


class _Split(torch.autograd.Function):
    """x [N, D] -> (relu?(x[:, :nd]), x[:, nd:]) as two contiguous tensors"""

    @staticmethod
    def forward(ctx, x, nd, relu):
        x = f32c(x)
        N, D = x.shape
        d = torch.empty(N, nd, device=x.device, dtype=torch.float32)
        a = torch.empty(N, D - nd, device=x.device, dtype=torch.float32)
        L.check(L.lib().ecgmm_split_cols(ptr(x), ptr(d), ptr(a), N, D, nd, int(relu), stream()), "split_cols")
        ctx.nd, ctx.relu, ctx.D = nd, relu, D
        ctx.save_for_backward(d)
        return d, a

    @staticmethod
    def backward(ctx, gd, ga):
        d, = ctx.saved_tensors
        gx = torch.empty(d.shape[0], ctx.D, device=d.device, dtype=torch.float32)
        L.check(L.lib().ecgmm_split_cols_bwd(ptr(d), ptr(f32c(gd)) if gd is not None else None,
                                             ptr(f32c(ga)) if ga is not None else None, ptr(gx), d.shape[0], ctx.D, ctx.nd,
                                             int(ctx.relu), stream()), "split_cols_bwd")
        return gx, None, None


class _Entropy(torch.autograd.Function):
    """mean_n sum_d M log(M + eps) -> tensor [1]"""

    @staticmethod
    def forward(ctx, M, eps):
        M = f32c(M)
        out = torch.empty(1, device=M.device, dtype=torch.float32)
        L.check(L.lib().ecgmm_entropy_fwd(ptr(M), ptr(out), M.shape[0], M.shape[1], eps, stream()), "entropy_fwd")
        ctx.eps = eps
        ctx.save_for_backward(M)
        return out

    @staticmethod
    def backward(ctx, g):
        M, = ctx.saved_tensors
        dM = torch.empty_like(M)
        L.check(L.lib().ecgmm_entropy_bwd(ptr(M), ptr(f32c(g)), ptr(dM), M.shape[0], M.shape[1], ctx.eps, stream()),
                "entropy_bwd")
        return dM, None


class _GhostBN(torch.autograd.Function):
    """BatchNorm1d applied to consecutive row slices ("virtual batches", torch.chunk's split), running statistics
    updated once per slice in order; every slice is written straight into one output tensor.  vbs = None: plain
    BatchNorm1d (one slice).  One launch per slice forward, one backward (ecgmm_bn_small_*: any channel count)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, rm, rv, nbt, training, momentum, eps, vbs):
        _require_cuda(x, "batchnorm")
        x = f32c(x)
        B, Cn = x.shape
        lib = L.lib()
        nch = 1 if vbs is None else int(math.ceil(B / vbs))
        size = int(math.ceil(B / nch))                  # torch.chunk: equal slices of ceil(B / nch) rows
        bounds = [(i, min(B, i + size)) for i in range(0, B, size)]
        y = torch.empty_like(x)
        save = torch.empty(len(bounds), 2, Cn, device=x.device, dtype=torch.float32)
        for k, (i0, i1) in enumerate(bounds):
            if training and i1 - i0 < 2:
                raise ValueError("Expected more than 1 value per channel when training")
            L.check(lib.ecgmm_bn_small_fwd(ptr(x[i0:i1]), ptr(gamma), ptr(beta), ptr(rm), ptr(rv), ptr(nbt), ptr(y[i0:i1]),
                                           ptr(save[k]), i1 - i0, Cn, int(training), momentum, eps, stream()), "bn_small_fwd")
        ctx.params, ctx.bounds, ctx.training = (gamma, beta), bounds, training
        ctx.save_for_backward(x, save)
        return y

    @staticmethod
    def backward(ctx, dy):
        if not ctx.training:
            raise RuntimeError("BatchNorm backward needs the forward to have run in training mode")
        x, save = ctx.saved_tensors
        gamma, beta = ctx.params
        dy = f32c(dy)
        lib = L.lib()
        Cn = x.shape[1]
        dx = torch.empty_like(x)
        dg = grad_sink(gamma) if ctx.needs_input_grad[1] else None
        db = grad_sink(beta) if ctx.needs_input_grad[2] else None
        for k, (i0, i1) in enumerate(ctx.bounds):      # parameter gradients add up over the virtual batches
            L.check(lib.ecgmm_bn_small_bwd(ptr(x[i0:i1]), ptr(dy[i0:i1]), ptr(gamma), ptr(save[k]), ptr(dx[i0:i1]), ptr(dg),
                                           ptr(db), i1 - i0, Cn, int(k > 0), stream()), "bn_small_bwd")
        return dx, None, None, None, None, None, None, None, None, None


class SmallBatchNorm1d(hnn.BatchNorm1d):
    """BatchNorm1d over [B, C] for any C (the library's 16-byte-chunk kernels need C % 4 == 0; TabNet's input has 2)"""

    def forward(self, x):
        return _GhostBN.apply(x, self.weight, self.bias, self.running_mean, self.running_var, self.num_batches_tracked,
                              self.training, self.momentum, self.eps, None)


class _AccLinearFn(torch.autograd.Function):
    """y = x W^T for a weight that is used SEVERAL times per forward (TabNet's shared GLU layers).  The library's
    gradient sinks are overwritten by each backward, so here dW goes to a temporary and is added into the sink
    (the first backward of a step overwrites, the later ones accumulate)."""

    @staticmethod
    def forward(ctx, x, module):
        _require_cuda(x, "linear")
        x = f32c(x)
        w = module.weight
        y = torch.empty(x.shape[0], w.shape[0], device=x.device, dtype=torch.float32)
        L.check(L.lib().ecgmm_linear_fwd(ptr(x), ptr(w), None, ptr(y), x.shape[0], x.shape[1], w.shape[0], L.ACT_NONE,
                                         stream()), "linear_fwd")
        ctx.module = module
        module._pending += 1
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, = ctx.saved_tensors
        mod = ctx.module
        w = mod.weight
        B, In = x.shape
        Out = w.shape[0]
        lib = L.lib()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        sink = grad_sink(w, accumulate=mod._seen > 0)   # first use of a step overwrites (and is guarded), later uses add
        tmp = torch.empty_like(w) if sink is not None else None
        scratch = _Scratch.get("linear", lib.ecgmm_linear_bwd_scratch(B, In, Out), x.device)
        L.check(lib.ecgmm_linear_bwd(ptr(f32c(dy)), ptr(x), ptr(w), ptr(dx), ptr(tmp), None, B, In, Out, ptr(scratch),
                                     scratch.numel(), stream()), "linear_bwd")
        if sink is not None:
            L.check(lib.ecgmm_axpby(1.0, ptr(tmp), 0.0 if mod._seen == 0 else 1.0, ptr(sink), w.numel(), stream()), "axpby")
        mod._seen += 1
        if mod._seen >= mod._pending:
            mod._seen = mod._pending = 0
        return dx, None


class SharedLinear(hnn.Linear):
    """bias-free Linear whose weight gradient accumulates over its uses within one backward"""

    def __init__(self, in_features, out_features):
        super().__init__(in_features, out_features, bias=False)
        self._pending = self._seen = 0

    def forward(self, x):
        if not torch.is_grad_enabled() or not self.weight.requires_grad:
            return HF.linear(x, self.weight, None)
        return _AccLinearFn.apply(x, self)


def initialize_non_glu(module, input_dim, output_dim):
    nn.init.xavier_normal_(module.weight, gain=np.sqrt((input_dim + output_dim) / np.sqrt(4 * input_dim)))


def initialize_glu(module, input_dim, output_dim):
    nn.init.xavier_normal_(module.weight, gain=np.sqrt((input_dim + output_dim) / np.sqrt(input_dim)))


class GBN(nn.Module):
    def __init__(self, input_dim, virtual_batch_size=128, momentum=0.01):
        super().__init__()
        self.input_dim, self.virtual_batch_size = input_dim, virtual_batch_size
        self.bn = hnn.BatchNorm1d(input_dim, momentum=momentum)

    def forward(self, x):
        bn = self.bn
        return _GhostBN.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                              bn.training, bn.momentum, bn.eps, self.virtual_batch_size)


class GLU_Layer(nn.Module):
    def __init__(self, input_dim, output_dim, fc=None, virtual_batch_size=128, momentum=0.02):
        super().__init__()
        self.output_dim = output_dim
        self.fc = fc if fc is not None else hnn.Linear(input_dim, 2 * output_dim, bias=False)
        initialize_glu(self.fc, input_dim, 2 * output_dim)
        self.bn = GBN(2 * output_dim, virtual_batch_size, momentum)

    def forward(self, x):
        return _GLU.apply(self.bn(self.fc(x)))


class GLU_Block(nn.Module):
    def __init__(self, input_dim, output_dim, n_glu=2, first=False, shared_layers=None, virtual_batch_size=128,
                 momentum=0.02):
        super().__init__()
        self.first, self.n_glu = first, n_glu
        self.glu_layers = nn.ModuleList()
        for i in range(n_glu):
            fc = shared_layers[i] if shared_layers else None
            self.glu_layers.append(GLU_Layer(input_dim if i == 0 else output_dim, output_dim, fc, virtual_batch_size,
                                             momentum))

    def forward(self, x):
        scale, start = math.sqrt(0.5), 0
        if self.first:   # the first layer of the block has no residual
            x, start = self.glu_layers[0](x), 1
        for i in range(start, self.n_glu):
            x = _AddScale.apply(x, self.glu_layers[i](x), scale)
        return x


class FeatTransformer(nn.Module):
    def __init__(self, input_dim, output_dim, shared_layers, n_glu_independent, virtual_batch_size=128, momentum=0.02):
        super().__init__()
        if shared_layers is None:
            self.shared, is_first = nn.Identity(), True
        else:
            self.shared = GLU_Block(input_dim, output_dim, len(shared_layers), True, shared_layers, virtual_batch_size,
                                    momentum)
            is_first = False
        if n_glu_independent == 0:
            self.specifics = nn.Identity()
        else:
            self.specifics = GLU_Block(input_dim if is_first else output_dim, output_dim, n_glu_independent, is_first,
                                       None, virtual_batch_size, momentum)

    def forward(self, x):
        return self.specifics(self.shared(x))


class AttentiveTransformer(nn.Module):
    def __init__(self, input_dim, group_dim, virtual_batch_size=128, momentum=0.02):
        super().__init__()
        self.fc = hnn.Linear(input_dim, group_dim, bias=False)
        initialize_non_glu(self.fc, input_dim, group_dim)
        self.bn = GBN(group_dim, virtual_batch_size, momentum)

    def forward(self, priors, processed_feat):
        x = self.bn(self.fc(processed_feat))
        if priors is not None:     # None = the all-ones prior of the first step
            x = _Mul.apply(x, priors)
        return _Sparsemax.apply(x)


class TabNetEncoder(nn.Module):
    def __init__(self, input_dim, output_dim, n_d=8, n_a=8, n_steps=3, gamma=1.3, n_independent=2, n_shared=2,
                 epsilon=1e-15, virtual_batch_size=128, momentum=0.02):
        super().__init__()
        if input_dim > 64:
            raise ValueError("TabNetEncoder: the sparsemax kernel handles up to 64 features")
        self.input_dim, self.n_d, self.n_a, self.n_steps, self.gamma, self.epsilon = input_dim, n_d, n_a, n_steps, gamma, epsilon
        self.initial_bn = SmallBatchNorm1d(input_dim, momentum=0.01)
        self.group_attention_matrix = torch.eye(input_dim)   # plain attribute in the library (identity grouping only here)
        shared = None
        if n_shared > 0:
            shared = nn.ModuleList([SharedLinear(input_dim if i == 0 else n_d + n_a, 2 * (n_d + n_a))
                                    for i in range(n_shared)])
        self.initial_splitter = FeatTransformer(input_dim, n_d + n_a, shared, n_independent, virtual_batch_size, momentum)
        self.feat_transformers = nn.ModuleList()
        self.att_transformers = nn.ModuleList()
        for _ in range(n_steps):
            self.feat_transformers.append(FeatTransformer(input_dim, n_d + n_a, shared, n_independent,
                                                          virtual_batch_size, momentum))
            self.att_transformers.append(AttentiveTransformer(n_a, input_dim, virtual_batch_size, momentum))

    def forward(self, x):
        for m in self.modules():     # a forward that was never back-propagated must not leave use counts behind
            if isinstance(m, SharedLinear):
                m._pending = m._seen = 0
        x = self.initial_bn(x)
        prior, m_loss = None, None
        _, att = _Split.apply(self.initial_splitter(x), self.n_d, False)
        steps_output = []
        for step in range(self.n_steps):
            M = self.att_transformers[step](prior, att)
            ent = _Entropy.apply(M, self.epsilon)
            m_loss = ent if m_loss is None else _AddScale.apply(m_loss, ent, 1.0)
            prior = _PriorUpdate.apply(M, prior, self.gamma)
            out = self.feat_transformers[step](_Mul.apply(M, x))
            d, att = _Split.apply(out, self.n_d, True)
            steps_output.append(d)
        return steps_output, _Scale.apply(m_loss, 1.0 / self.n_steps)


class TabNetNoEmbeddings(nn.Module):
    def __init__(self, input_dim, output_dim, n_d=8, n_a=8, n_steps=3, gamma=1.3, n_independent=2, n_shared=2,
                 epsilon=1e-15, virtual_batch_size=128, momentum=0.02):
        super().__init__()
        self.encoder = TabNetEncoder(input_dim, output_dim, n_d, n_a, n_steps, gamma, n_independent, n_shared, epsilon,
                                     virtual_batch_size, momentum)
        self.final_mapping = hnn.Linear(n_d, output_dim, bias=False)
        initialize_non_glu(self.final_mapping, n_d, output_dim)

    def forward(self, x):
        steps_output, m_loss = self.encoder(x)
        res = steps_output[0]
        for d in steps_output[1:]:
            res = _AddScale.apply(res, d, 1.0)
        return self.final_mapping(res), m_loss.reshape(())


class ClinicalTabNetEncoder(nn.Module):
    """multimodal.py:109-148: forward(x) -> (latent [B, latent_dim], M_loss)"""

    def __init__(self, input_dim, latent_dim=32, device=None):
        super().__init__()
        self.device, self.latent_dim = device, latent_dim
        self.tabnet = TabNetNoEmbeddings(input_dim=input_dim, output_dim=latent_dim, n_d=latent_dim, n_a=latent_dim,
                                         n_steps=3, gamma=1.5, n_independent=2, n_shared=2)

    def forward(self, x):
        return self.tabnet(x)

    def load_pretrained_partial(self, weight_path):
        """everything but the output layer (multimodal.py:150-167)"""
        saved = torch.load(weight_path, map_location="cpu")
        filtered = {k: v for k, v in saved.items() if "final_mapping" not in k}
        return self.tabnet.load_state_dict(filtered, strict=False)
