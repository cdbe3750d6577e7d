"""the TabNet clinical encoder (multimodal.py:109-148) on HIP kernels vs the torch-CPU restatement of
pytorch_tabnet's published algorithm (oracle/tabnet_ref.py -- PARITY UNPINNED: the library is neither vendored in the
reference nor installed; see that file's header)."""
import numpy as np
import pytest
import torch

from ecgmm import tabnet as G
from ecgmm.config import Config
from oracle import fill, ref_models as O, tabnet_ref as T

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.mark.parametrize("shape", [(7, 2), (130, 5), (33, 64)])
def test_sparsemax_glu_entropy_fwd_bwd(shape):
    x = fill.hash_tensor(shape, 5, 2.0).requires_grad_(True)
    g = fill.hash_tensor(shape, 6, 1.0)
    p_ref = T.sparsemax(x)
    p_ref.backward(g)
    xg = x.detach().to(DEV).requires_grad_(True)
    p = G._Sparsemax.apply(xg)
    p.backward(g.to(DEV))
    assert torch.allclose(p.cpu(), p_ref.detach(), atol=1e-6) and torch.allclose(p.sum(1).cpu(), torch.ones(shape[0]), atol=1e-5)
    assert torch.allclose(xg.grad.cpu(), x.grad, atol=1e-6)
    if shape[1] % 2 == 0:
        z = fill.hash_tensor(shape, 7, 2.0).requires_grad_(True)
        D = shape[1] // 2
        o_ref = z[:, :D] * torch.sigmoid(z[:, D:])
        go = fill.hash_tensor((shape[0], D), 8)
        o_ref.backward(go)
        zg = z.detach().to(DEV).requires_grad_(True)
        o = G._GLU.apply(zg)
        o.backward(go.to(DEV))
        assert torch.allclose(o.cpu(), o_ref.detach(), atol=1e-6) and torch.allclose(zg.grad.cpu(), z.grad, atol=1e-6)
    m = p_ref.detach().clone().requires_grad_(True)
    e_ref = torch.mean(torch.sum(m * torch.log(m + 1e-15), dim=1))
    e_ref.backward()
    mg = m.detach().to(DEV).requires_grad_(True)
    e = G._Entropy.apply(mg, 1e-15)
    e.sum().backward()
    assert abs(e.item() - e_ref.item()) < 1e-5 and torch.allclose(mg.grad.cpu(), m.grad, atol=1e-5)


@pytest.mark.parametrize("B", [256, 300, 130, 40])
def test_clinical_tabnet_encoder_train_step_vs_restatement(B):
    torch.manual_seed(1)
    ref = T.ClinicalTabNetEncoder(2).train()
    for n_, b in ref.named_buffers():            # non-trivial running statistics
        if n_.endswith("running_mean"):
            b.copy_(0.1 * fill.hash_tensor(tuple(b.shape), 3))
    model = G.ClinicalTabNetEncoder(2)
    assert list(model.state_dict()) == list(ref.state_dict())           # the library's key names
    model.load_state_dict(ref.state_dict())
    model = model.to(DEV).train()
    x = fill.hash_tensor((B, 2), 17, 1.5)
    w = fill.hash_tensor((B, 32), 18)
    out_r, ml_r = ref(x)
    ((out_r * w).sum() + 0.3 * ml_r).backward()
    out, ml = model(x.to(DEV))
    ((G._Mul.apply(out, w.to(DEV))).sum() + 0.3 * ml).backward()
    assert out.shape == (B, 32) and ml.dim() == 0
    assert rel(out.cpu(), out_r.detach()) < 1e-4 and abs(ml.item() - ml_r.item()) < 1e-5
    ref_p = dict(ref.named_parameters())
    for name, p in model.named_parameters():
        assert p.grad is not None, name
        assert rel(p.grad.cpu(), ref_p[name].grad) < 3e-3, name
    ref_b = dict(ref.named_buffers())
    for name, b in model.named_buffers():
        assert torch.allclose(b.cpu().float(), ref_b[name].float(), atol=1e-5), name
    # eval mode: running statistics
    model.eval(); ref.eval()
    with torch.no_grad():
        oe, me = model(x.to(DEV))
        oer, mer = ref(x)
    # (running statistics have barely moved after one step at momentum 0.02, so eval activations are O(10) and fp32
    # rounding through the chain scales with that: the bar is relative to the largest activation)
    assert rel(oe.cpu(), oer) < 1e-4 and abs(me.item() - mer.item()) < 1e-5


def test_multimodal_py_variant_matches_restatement_and_trains():
    """multimodal.py's ECGMultimodalModel: widths 512 / 128 / 32, TabNet clinical branch"""
    from ecgmm.multimodal import ECGMultimodalModel
    from ecgmm.hip import functional as HF
    from ecgmm.optim import FusedAdam
    cfg = type("C", (Config,), {"device": DEV, "compute_dtype": "fp32"})
    ref = O.disable_dropout(fill.hash_fill_module(T.multimodal_tabnet_model(2), "mmt."))
    model = ECGMultimodalModel(cfg)
    assert model.fusion_classifier[0].weight.shape == (128, 672) and model.get_clinical_feature_dim() == 2
    model.load_state_dict(ref.state_dict())
    model = O.disable_dropout(model).to(DEV)
    img, sig, _clin16, lab = fill.synthetic_batch(8, salt=4)
    clin = fill.hash_tensor((8, 2), 44, 1.0)
    ref.eval(); model.eval()
    with torch.no_grad():
        r = ref(img, sig, clin)
        o = model(img.to(DEV), sig.to(DEV), clin.to(DEV))
    assert torch.allclose(o[3].cpu(), r[3], atol=1e-3) and abs(o[4].item() - r[4].item()) < 1e-4   # logits bar 1e-3
    model.train()
    opt = FusedAdam(model.parameters(), lr=1e-3)
    losses = []
    for _ in range(4):
        opt.zero_grad()
        out = model(img.to(DEV), sig.to(DEV), clin.to(DEV))
        loss = HF.cross_entropy(out[3], lab.to(DEV)) + 0.1 * out[4]
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert np.isfinite(losses).all() and losses[-1] < losses[0]
