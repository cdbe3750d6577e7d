"""Fusion-head ops through the autograd layer (hip/functional.py) vs the torch-CPU ops the reference
calls: Linear(+act), LayerNorm, AttentionFusion, var_loss, CrossEntropy, FocalLoss (golden g4),
BatchNorm1d+ReLU, Dropout statistics."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from ecgmm.hip import encoders as E
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


@pytest.mark.parametrize("B", [24, 32, 256])   # 24: per-op launch plan; multiples of 16: fused row kernels + dense16 (head_fused.hip)
@pytest.mark.parametrize("loss_kind", ["train_py", "all_heads", "branch_only"])
def test_fused_head_plan_matches_modules_and_oracle(loss_kind, B):
    """csrc/plan_head.hip (one native call per direction) == the module-by-module head == the oracle's head
    (PMB:326-354): all six outputs, every parameter gradient, the gradients handed to the encoders; and heads that are not
    in the loss (train.py:78 uses fusion_logits + var_loss only) keep grad None."""
    from ecgmm.config import Config
    from ecgmm.multimodal_paper_modal_balance import ECGMultimodalModel
    from oracle import fill, ref_models as O
    cfg = type("C", (Config,), {"clinical_input_dim": 16, "compute_dtype": "fp32"})
    ref = O.disable_dropout(fill.hash_fill_module(O.ECGMultimodalModel(2, 16), "mm.")).train()
    nets = []
    for fused in (True, False):
        c = type("Cf", (cfg,), {"fused_head": fused})
        n = ECGMultimodalModel(c)
        n.load_state_dict(ref.state_dict())
        for m in n.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        nets.append(n.to(DEV).train())
    raws = [fill.hash_tensor((B, 256), 90 + i, 1.5) for i in range(3)]
    lab = torch.arange(B) % 2

    def loss_of(out, y, CE):
        if loss_kind == "train_py":
            return CE(out[3], y) + 0.1 * out[4]
        if loss_kind == "all_heads":
            return CE(out[0], y) + CE(out[1], y) + CE(out[2], y) + CE(out[3], y) + 0.1 * out[4]
        return CE(out[1], y)

    def ref_head(r):
        f = [ref.image_norm(r[0]), ref.signal_norm(r[1]), ref.clinical_norm(r[2])]
        fused, w = ref.attention_fusion(*f)
        v = [torch.var(t, dim=1).mean() for t in f]
        var = (v[0] - v[1]).abs() + (v[0] - v[2]).abs() + (v[1] - v[2]).abs()
        return (ref.image_classifier(f[0]), ref.signal_classifier(f[1]), ref.clinical_classifier(f[2]),
                ref.fusion_classifier(fused), var, w)

    rr = [t.clone().requires_grad_(True) for t in raws]
    out_ref = ref_head(rr)
    loss_of(out_ref, lab, torch.nn.functional.cross_entropy).backward()
    results = []
    for n in nets:
        rg = [dev(t).requires_grad_(True) for t in raws]
        spec = n._head_spec()
        out = E.run_head(*rg, spec, n._head_params()) if spec is not None else n._head_by_modules(*rg)
        loss_of(out, dev(lab), HF.cross_entropy).backward()
        torch.cuda.synchronize()
        results.append((out, rg, n))
    assert nets[0]._head_spec() is not None and nets[1]._head_spec() is None
    for out, rg, n in results:
        for a, b in zip(out, out_ref):
            assert (a.detach().cpu() - b.detach()).abs().max() < 2e-5
        for a, b in zip(rg, rr):
            if b.grad is None:
                assert a.grad is None or a.grad.abs().max() == 0
            else:
                assert rel_err(a.grad.cpu(), b.grad) < 1e-4
        pr = dict(ref.named_parameters())
        for k, p in n.named_parameters():
            if "encoder" in k:
                continue
            if pr[k].grad is None:
                assert p.grad is None, k
            else:
                assert p.grad is not None and rel_err(p.grad.cpu(), pr[k].grad) < 2e-4, k
    # fused == unfused to fp32 rounding
    for a, b in zip(results[0][0], results[1][0]):
        assert (a - b).abs().max() < 1e-6


@pytest.mark.parametrize("variant", ["tabnet_512_128_32", "generic_320x3_3classes"])
def test_fused_head_row_kernels_at_other_widths(variant):
    """multimodal.py's head (widths 512 / 128 / 32, fused 672 -> 128 -> 2) and a width / class count outside both
    reference models: fused row kernels + dense16 kernels (csrc/head_fused.hip, its <8,2,1,2> and generic
    instantiations) == the module-by-module head, outputs and every gradient"""
    from ecgmm.config import Config
    from oracle import fill
    if variant.startswith("tabnet"):
        from ecgmm.multimodal import ECGMultimodalModel
        dims, extra, nc = (512, 128, 32), {}, 2
    else:
        from ecgmm.multimodal_paper_modal_balance import ECGMultimodalModel
        dims, extra, nc = (320, 320, 320), {"modal_dim": 320, "num_classes": 3, "clinical_input_dim": 16}, 3
    B = 48
    raws = [fill.hash_tensor((B, d), 70 + i, 1.5) for i, d in enumerate(dims)]
    lab = dev(torch.arange(B) % nc)
    res = []
    sd = None
    for fused in (True, False):
        cfg = type("Cf", (Config,), dict({"device": DEV, "compute_dtype": "fp32", "fused_head": fused}, **extra))
        n = ECGMultimodalModel(cfg)
        if sd is None:
            sd = fill.hash_fill_module(n, "hw.").state_dict()
        n.load_state_dict(sd)
        for m in n.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        n = n.to(DEV).train()
        rg = [dev(t).requires_grad_(True) for t in raws]
        spec = n._head_spec()
        assert (spec is not None) == fused
        out = E.run_head(*rg, spec, n._head_params()) if fused else n._head_by_modules(*rg)
        (HF.cross_entropy(out[3], lab) + HF.cross_entropy(out[0], lab) + 0.1 * out[4]).backward()
        torch.cuda.synchronize()
        res.append((out, rg, {k: p.grad for k, p in n.named_parameters() if "encoder" not in k}))
    for a, b in zip(res[0][0], res[1][0]):
        assert (a - b).abs().max() < 2e-5
    for a, b in zip(res[0][1], res[1][1]):
        assert rel_err(a.grad, b.grad) < 1e-4
    for k, g in res[1][2].items():
        if g is None:
            assert res[0][2][k] is None, k
        else:
            assert res[0][2][k] is not None and rel_err(res[0][2][k], g) < 2e-4, k


def test_cross_entropy_plus_is_ce_plus_weighted_extra():
    logits = dev(fill.hash_tensor((12, 2), 61, 2.0)).requires_grad_(True)
    extra = dev(torch.tensor(0.37)).requires_grad_(True)
    lab = dev(torch.arange(12) % 2)
    loss = HF.cross_entropy_plus(logits, lab, extra, 0.1)
    loss.backward()
    lr = logits.detach().cpu().clone().requires_grad_(True)
    er = torch.tensor(0.37, requires_grad=True)
    ref = torch.nn.functional.cross_entropy(lr, lab.cpu()) + 0.1 * er
    ref.backward()
    assert abs(loss.item() - ref.item()) < 1e-6 and rel_err(logits.grad.cpu(), lr.grad) < 1e-6
    assert abs(extra.grad.item() - 0.1) < 1e-7
