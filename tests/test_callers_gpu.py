"""the callers after the hot path -- the XAI scripts' way of reaching into the model (sub-module calls,
fusion_classifier wrapper with autograd on its input) and the nested k-fold harness."""
import numpy as np
import pytest
import torch

from ecgmm import shap_fusion_modal_balance as X
from ecgmm.config import Config
from ecgmm.multimodal_paper_modal_balance import ECGMultimodalModel
from oracle import fill, ref_models as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cfg(**kw):
    base = {"device": DEV, "compute_dtype": "fp32", "clinical_input_dim": 16, "batch_size": 8,
            "synthetic_train_size": 32, "synthetic_val_size": 8, "synthetic_test_size": 8}
    base.update(kw)
    return type("C", (Config,), base)


def test_submodule_call_sequence_matches_the_oracle():
    """shap_fusion_modal_balance.py:65-76 calls image_encoder / image_norm / ... / attention_fusion one by one"""
    cfg = _cfg()
    ref = O.disable_dropout(fill.hash_fill_module(O.ECGMultimodalModel(2, 16), "mm.")).eval()
    model = ECGMultimodalModel(cfg)
    model.load_state_dict(ref.state_dict())
    model = model.to(DEV).eval()
    img, sig, clin, _ = fill.synthetic_batch(4, salt=9)
    with torch.no_grad():
        feats = X.modal_features(model, img.to(DEV), sig.to(DEV), clin.to(DEV))
        fused, sw = model.attention_fusion(*feats)
        logits = X.FusionClassifierWrapper(model.fusion_classifier)(fused)
        r_img = ref.image_norm(ref.image_encoder(img))
        r_sig = ref.signal_norm(ref.signal_encoder(sig.unsqueeze(1)))
        r_clin = ref.clinical_norm(ref.clinical_encoder(clin))
        r_fused, r_sw = ref.attention_fusion(r_img, r_sig, r_clin)
        r_logits = ref.fusion_classifier(r_fused)
    assert torch.allclose(feats[0].cpu(), r_img, atol=2e-4) and torch.allclose(feats[1].cpu(), r_sig, atol=2e-4)
    assert torch.allclose(fused.cpu(), r_fused, atol=2e-4) and torch.allclose(sw.cpu(), r_sw, atol=1e-6)
    assert torch.allclose(logits.cpu(), r_logits, atol=1e-3)          # north-star bar on the logits
    assert len(X.fusion_fc_chunk_norms(model)) == 3


def test_expected_gradients_are_complete_and_match_cpu_autograd():
    """attributions through the HIP ops' autograd == the same estimator on the torch oracle; and they sum to
    f(x) - f(baseline) when the background is a single point (completeness of the path integral, many samples)"""
    cfg = _cfg()
    ref = O.disable_dropout(fill.hash_fill_module(O.ECGMultimodalModel(2, 16), "mm.")).eval()
    model = ECGMultimodalModel(cfg)
    model.load_state_dict(ref.state_dict())
    model = model.to(DEV).eval()
    x = fill.hash_tensor((3, 768), 71, 1.0)
    bg = fill.hash_tensor((5, 768), 72, 1.0)
    got = X.expected_gradients(X.FusionClassifierWrapper(model.fusion_classifier), bg.to(DEV), x.to(DEV), nsamples=16, seed=3)
    want = X.expected_gradients(ref.fusion_classifier, bg, x, nsamples=16, seed=3)
    assert torch.allclose(got.cpu(), want, atol=2e-5)
    one = bg[:1]
    sv = X.expected_gradients(X.FusionClassifierWrapper(model.fusion_classifier), one.to(DEV), x.to(DEV), nsamples=400, seed=1)
    with torch.no_grad():
        delta = ref.fusion_classifier(x) - ref.fusion_classifier(one)
    assert torch.allclose(sv.sum(1).cpu(), delta, atol=0.05 * delta.abs().max().item() + 1e-3)
    rows = X.modality_contributions(sv, [256, 256, 256], [0, 1, 1])
    assert len(rows) == 6 and abs(rows[0]["Image_%"] + rows[0]["Signal_%"] + rows[0]["Clinical_%"] - 100) < 1e-6


def test_attribution_entry_point_and_kfold_harness(tmp_path):
    from ecgmm import train_kfold
    cfg = _cfg(compute_dtype="bf16", checkpoint_dir=str(tmp_path / "ck"), k_outer=2, k_inner=2, num_epochs=1,
               synthetic_train_size=24)
    df = X.main(cfg, max_samples_per_class=4, nsamples=4, out_csv=str(tmp_path / "shap" / "out.csv"), quiet=True)
    assert set(df.columns) >= {"Sample_ID", "Image_%", "Signal_%", "Clinical_%", "Label", "Class"} and len(df) == 16
    aucs = train_kfold.main(cfg, num_epochs=1, quiet=True)
    assert len(aucs) == 2 and all(np.isnan(a) or 0.0 <= a <= 1.0 for a in aucs)


@pytest.mark.parametrize("training", [False, True])
@pytest.mark.parametrize("cin,cout,stride,Ln", [(64, 64, 1, 96), (64, 128, 2, 101), (128, 256, 2, 64)])
def test_block_level_modules_run_standalone(cin, cout, stride, Ln, training):
    """``SEBlock.forward`` / ``BasicBlock1D.forward`` are ordinary callables in the reference
    (multimodal_paper_modal_balance.py:58-62, 86-93); here they run on the per-op kernels (ecgmm/hip/blocks.py) and give
    the oracle block's values (fp32 path, train mode = batch statistics + running-statistics update, eval mode = running
    statistics)."""
    from ecgmm.multimodal_paper_modal_balance import BasicBlock1D, SEBlock
    ref = fill.hash_fill_module(O.BasicBlock1D(cin, cout, stride=stride), "blk.")
    ref.bn1.running_var.mul_(1.7); ref.bn2.running_mean.add_(0.1)
    blk = BasicBlock1D(cin, cout, stride=stride)
    blk.load_state_dict(ref.state_dict())
    blk = blk.to(DEV)
    ref.train(training); blk.train(training)
    x = fill.hash_tensor((5, cin, Ln), 77, 1.2)
    with torch.no_grad():
        want = ref(x)
    got = blk(x.to(DEV))
    assert got.shape == want.shape and not got.requires_grad
    assert (got.cpu() - want).abs().max() < 2e-4, float((got.cpu() - want).abs().max())
    if training:
        for a, b in ((blk.bn1, ref.bn1), (blk.bn2, ref.bn2)):
            assert torch.allclose(a.running_mean.cpu(), b.running_mean, atol=1e-5) and int(a.num_batches_tracked) == 1
            assert torch.allclose(a.running_var.cpu(), b.running_var, rtol=1e-4, atol=1e-6)
    se_ref = fill.hash_fill_module(O.SEBlock(cout), "se.")
    se = SEBlock(cout); se.load_state_dict(se_ref.state_dict()); se = se.to(DEV)
    with torch.no_grad():
        w2 = se_ref(want)
    assert (se(got).cpu() - w2).abs().max() < 2e-4
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        blk.cpu()(x)
