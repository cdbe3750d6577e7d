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
