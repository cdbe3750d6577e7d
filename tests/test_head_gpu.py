"""Fusion-head ops through the autograd layer (hip/functional.py) vs the torch-CPU ops the reference
calls: Linear(+act), LayerNorm, AttentionFusion, var_loss, CrossEntropy, FocalLoss (golden g4),
BatchNorm1d+ReLU, Dropout statistics."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from ecgmm.hip import functional as HF
from ecgmm.hip import lib as L
from ecgmm.hip import nn as hnn
from oracle import fill, ref_models as O

from .util import DEV, dev, rel_err

pytestmark = pytest.mark.gpu


def _leaf(t):
    return t.clone().requires_grad_(True)


def _gpu_param(t):
    return torch.nn.Parameter(dev(t.clone()))


@pytest.mark.parametrize("shape", [(8, 512, 256), (256, 768, 128), (8, 128, 2), (5, 16, 64), (6, 64, 4), (6, 4, 64),
                                   (256, 256, 64)])
@pytest.mark.parametrize("act", [L.ACT_NONE, L.ACT_RELU, L.ACT_SIGMOID])
def test_linear(shape, act):
    B, In, Out = shape
    x, w, b = fill.hash_tensor((B, In), 1), fill.hash_tensor((Out, In), 2, In ** -0.5), fill.hash_tensor((Out,), 3, 0.1)
    xr, wr, br = _leaf(x), _leaf(w), _leaf(b)
    z = F.linear(xr, wr, br)
    y_ref = {L.ACT_NONE: z, L.ACT_RELU: F.relu(z), L.ACT_SIGMOID: torch.sigmoid(z)}[act]
    dy = fill.hash_tensor((B, Out), 4)
    y_ref.backward(dy)
    xg = dev(x).requires_grad_(True)
    wg, bg = _gpu_param(w), _gpu_param(b)
    y = HF.linear(xg, wg, bg, act)
    y.backward(dev(dy))
    torch.cuda.synchronize()
    assert rel_err(y.detach().cpu(), y_ref.detach()) < 1e-5
    assert rel_err(xg.grad.cpu(), xr.grad) < 1e-5
    assert rel_err(wg.grad.cpu(), wr.grad) < 1e-5
    assert rel_err(bg.grad.cpu(), br.grad) < 1e-5


@pytest.mark.parametrize("B,D", [(8, 256), (33, 768), (3, 100)])
def test_layernorm(B, D):
    x = fill.hash_tensor((B, D), 5, 2.0) + 0.3
    g, b = 1 + 0.2 * fill.hash_tensor((D,), 6), 0.1 * fill.hash_tensor((D,), 7)
    xr, gr, br = _leaf(x), _leaf(g), _leaf(b)
    y_ref = F.layer_norm(xr, (D,), gr, br, 1e-5)
    dy = fill.hash_tensor((B, D), 8)
    y_ref.backward(dy)
    xg = dev(x).requires_grad_(True)
    gg, bg = _gpu_param(g), _gpu_param(b)
    y = HF.layer_norm(xg, gg, bg)
    y.backward(dev(dy))
    torch.cuda.synchronize()
    assert (y.detach().cpu() - y_ref.detach()).abs().max() < 2e-5
    assert rel_err(xg.grad.cpu(), xr.grad) < 2e-5
    assert rel_err(gg.grad.cpu(), gr.grad) < 2e-5 and rel_err(bg.grad.cpu(), br.grad) < 2e-5


def test_attention_fusion_and_var_loss():
    """AttentionFusion (PMB:31-46) + var_loss (PMB:349-352) against the oracle's module."""
    B, dims = 8, [256, 256, 256]
    feats = [fill.hash_tensor((B, d), 10 + i, 1.0 + 0.5 * i) for i, d in enumerate(dims)]
    ref = O.AttentionFusion(dims)
    with torch.no_grad():
        ref.weights.copy_(torch.tensor([1.3, 0.7, 1.0]))
        ref.norm.weight.copy_(1 + 0.1 * fill.hash_tensor((768,), 20))
        ref.norm.bias.copy_(0.1 * fill.hash_tensor((768,), 21))
    fr = [_leaf(f) for f in feats]
    fused_ref, w_ref = ref(*fr)
    vl_ref = (torch.var(fr[0], dim=1).mean() - torch.var(fr[1], dim=1).mean()).abs() + \
             (torch.var(fr[0], dim=1).mean() - torch.var(fr[2], dim=1).mean()).abs() + \
             (torch.var(fr[1], dim=1).mean() - torch.var(fr[2], dim=1).mean()).abs()
    dy = fill.hash_tensor((B, 768), 22)
    ((fused_ref * dy).sum() + 0.1 * vl_ref).backward()

    fg = [dev(f).requires_grad_(True) for f in feats]
    wg = _gpu_param(ref.weights.detach())
    gg, bg = _gpu_param(ref.norm.weight.detach()), _gpu_param(ref.norm.bias.detach())
    fused, soft = HF.attention_fusion(fg[0], fg[1], fg[2], wg, gg, bg)
    vl = HF.var_loss(*fg)
    ((fused * dev(dy)).sum() + 0.1 * vl).backward()
    torch.cuda.synchronize()
    assert (fused.detach().cpu() - fused_ref.detach()).abs().max() < 2e-5
    assert torch.allclose(soft.cpu(), w_ref.detach(), atol=1e-6)
    assert abs(vl.item() - vl_ref.item()) < 1e-5
    for a, b in zip(fg, fr):
        assert rel_err(a.grad.cpu(), b.grad) < 2e-5
    assert rel_err(wg.grad.cpu(), ref.weights.grad) < 1e-4
    assert rel_err(gg.grad.cpu(), ref.norm.weight.grad) < 2e-5
    assert rel_err(bg.grad.cpu(), ref.norm.bias.grad) < 2e-5


def test_cross_entropy_and_focal(golden_dir):
    g4 = np.load(f"{golden_dir}/g4_focal.npz")
    # the reference's own known answer: FocalLoss()([[2,-1],[.3,.1]], [0,1]) = 0.12070029
    lg = dev(torch.tensor([[2.0, -1.0], [0.3, 0.1]]))
    tg = dev(torch.tensor([0, 1]))
    assert abs(HF.focal_loss(lg, tg).item() - float(g4["kat"])) < 1e-6
    logits, targets = torch.from_numpy(g4["logits"]), torch.from_numpy(g4["targets"])
    v = HF.focal_loss(dev(logits), dev(targets), alpha=0.25, gamma=2.0)
    assert abs(v.item() - float(g4["loss_a025"])) < 1e-6
    for focal in (False, True):
        lr = _leaf(logits)
        ref = O.FocalLoss()(lr, targets) if focal else F.cross_entropy(lr, targets)
        (ref * 1.7).backward()
        lgp = dev(logits).requires_grad_(True)
        out = HF.focal_loss(lgp, dev(targets)) if focal else HF.cross_entropy(lgp, dev(targets))
        (out * 1.7).backward()
        torch.cuda.synchronize()
        assert abs(out.item() - ref.item()) < 1e-6
        assert rel_err(lgp.grad.cpu(), lr.grad) < 1e-5


def test_batchnorm1d_relu_module():
    B, Cn = 16, 64
    x = fill.hash_tensor((B, Cn), 30, 2.0)
    ref = torch.nn.BatchNorm1d(Cn)
    mine = hnn.BatchNorm1d(Cn).to(DEV)
    with torch.no_grad():
        ref.weight.copy_(1 + 0.2 * fill.hash_tensor((Cn,), 31)); ref.bias.copy_(0.1 * fill.hash_tensor((Cn,), 32))
        mine.weight.copy_(ref.weight); mine.bias.copy_(ref.bias)
    xr = _leaf(x)
    y_ref = F.relu(ref(xr))
    dy = fill.hash_tensor((B, Cn), 33)
    y_ref.backward(dy)
    xg = dev(x).requires_grad_(True)
    y = mine(xg, relu=True)
    y.backward(dev(dy))
    torch.cuda.synchronize()
    assert (y.detach().cpu() - y_ref.detach()).abs().max() < 1e-5
    assert rel_err(xg.grad.cpu(), xr.grad) < 1e-4
    assert rel_err(mine.weight.grad.cpu(), ref.weight.grad) < 1e-4
    assert torch.allclose(mine.running_var.cpu(), ref.running_var, rtol=1e-5)
    assert int(mine.num_batches_tracked.item()) == 1
    mine.eval(); ref.eval()
    with torch.no_grad():
        assert (mine(dev(x)).cpu() - ref(x)).abs().max() < 1e-5
    mine.train()
    with pytest.raises(ValueError):
        mine(dev(x[:1]))


def test_dropout_statistics_and_backward():
    HF.manual_seed(123)
    x = torch.ones(1 << 16, device=DEV, requires_grad=True)
    y = HF.dropout(x, 0.3, True)
    y.sum().backward()
    torch.cuda.synchronize()
    keep = (y.detach() != 0).float().mean().item()
    assert abs(keep - 0.7) < 0.01
    assert torch.allclose(y.detach()[y.detach() != 0], torch.tensor(1 / 0.7, device=DEV))
    assert torch.equal(x.grad != 0, y.detach() != 0)
    y2 = HF.dropout(x, 0.3, True)          # the Philox offset advanced: a different mask
    assert not torch.equal(y2.detach() != 0, y.detach() != 0)
    assert HF.dropout(x, 0.3, False) is x  # eval: identity
