"""Parameter-holding nn.Modules whose forward runs HIP kernels.  They keep torch's parameter names,
shapes, default initialisation and state_dict format, so reference checkpoints load unchanged."""
import math

import torch
import torch.nn as nn

from . import functional as HF
from . import lib as L


class Linear(nn.Module):
    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            bound = 1 / math.sqrt(self.in_features)
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x, act=L.ACT_NONE):
        return HF.linear(x, self.weight, self.bias, act)

    def extra_repr(self):
        return f"in_features={self.in_features}, out_features={self.out_features}, bias={self.bias is not None}"


class LayerNorm(nn.Module):
    def __init__(self, normalized_shape, eps=1e-5):
        super().__init__()
        self.normalized_shape = (int(normalized_shape),)
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(normalized_shape))
        self.bias = nn.Parameter(torch.zeros(normalized_shape))

    def forward(self, x):
        return HF.layer_norm(x, self.weight, self.bias, self.eps)


class _ConvNd(nn.Module):
    """Weights of a convolution that only ever runs inside an encoder launch plan."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, bias, nd):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, *([kernel_size] * nd)))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            bound = 1 / math.sqrt(in_channels * kernel_size ** nd)
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x):
        raise RuntimeError(f"{type(self).__name__} is executed by its parent encoder's fused launch plan; "
                           "call the encoder (e.g. model.image_encoder(x)), not the layer")

    def extra_repr(self):
        return (f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, stride={self.stride}, "
                f"padding={self.padding}, bias={self.bias is not None}")


class Conv2d(_ConvNd):
    def __init__(self, cin, cout, k, stride=1, padding=0, bias=True):
        super().__init__(cin, cout, k, stride, padding, bias, 2)


class Conv1d(_ConvNd):
    def __init__(self, cin, cout, kernel_size, stride=1, padding=0, bias=True):
        super().__init__(cin, cout, kernel_size, stride, padding, bias, 1)


class _BatchNorm(nn.Module):
    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def extra_repr(self):
        return f"{self.num_features}, eps={self.eps}, momentum={self.momentum}"


class BatchNorm2d(_BatchNorm):
    def forward(self, x):
        raise RuntimeError("BatchNorm2d is executed by its parent encoder's fused launch plan")


class BatchNorm1d(_BatchNorm):
    """Stand-alone use is the [B, C] case of the clinical MLP (PMB:258); inside ResNet1D_SE it is
    executed by the encoder plan."""

    def forward(self, x, relu=False):
        if x.dim() != 2:
            raise RuntimeError("BatchNorm1d over [B, C, L] is executed by its parent encoder's fused launch plan")
        return HF.batch_norm1d(x, self.weight, self.bias, self.running_mean, self.running_var,
                               self.num_batches_tracked, self.training, self.momentum, self.eps, relu)


class Sequential(nn.Sequential):
    """nn.Sequential whose forward fuses (Linear, ReLU|Sigmoid) and (BatchNorm1d, ReLU) pairs into one
    kernel epilogue and routes Dropout / Flatten to the HIP ops."""

    def forward(self, x):
        mods = list(self)
        i = 0
        while i < len(mods):
            m = mods[i]
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            if isinstance(m, (Linear, nn.Linear)):
                act = L.ACT_NONE
                if isinstance(nxt, nn.ReLU):
                    act, i = L.ACT_RELU, i + 1
                elif isinstance(nxt, nn.Sigmoid):
                    act, i = L.ACT_SIGMOID, i + 1
                x = HF.linear(x, m.weight, m.bias, act)
            elif isinstance(m, BatchNorm1d):
                relu = isinstance(nxt, nn.ReLU)
                x = m(x, relu=relu)
                i += int(relu)
            elif isinstance(m, nn.Dropout):
                x = HF.dropout(x, m.p, self.training and m.training)
            elif isinstance(m, nn.Flatten):
                x = x.reshape(x.shape[0], -1)
            elif isinstance(m, nn.Identity):
                pass
            else:
                x = m(x)
            i += 1
        return x
