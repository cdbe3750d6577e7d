"""Modality attribution of the fusion head (reference: shap_fusion_modal_balance.py:14-215).

Same flow and names: ``FusionClassifierWrapper`` (:14-20), ``get_embedding_batches`` (class-balanced background of
fused embeddings through ``model.image_encoder / image_norm / signal_encoder / ... / attention_fusion``, :50-92),
then per test sample the attribution of the 768 fusion inputs aggregated to Image / Signal / Clinical percentages
(:158-190).  The reference asks ``shap.GradientExplainer`` for the attributions; shap is not a dependency here, so
``expected_gradients`` implements the estimator GradientExplainer computes (expected gradients: for baselines x'
from the background and alpha ~ U(0,1), E[(x - x') * df/dx(x' + alpha (x - x'))]) on the HIP ops' own autograd.
"""
import numpy as np
import torch

from .config import Config
from .dataset import get_dataloaders
from .multimodal_paper_modal_balance import ECGMultimodalModel


class FusionClassifierWrapper(torch.nn.Module):
    def __init__(self, fusion_classifier):
        super().__init__()
        self.fusion_classifier = fusion_classifier

    def forward(self, fusion_embedding):
        return self.fusion_classifier(fusion_embedding)


def modal_features(model, images, ecg_signals, clinical):
    """the three normalised modality features, exactly as the reference scripts reach into the model (:65-75)"""
    img_feat = model.image_norm(model.image_encoder(images))
    signal_feat = model.signal_norm(model.signal_encoder(ecg_signals.unsqueeze(1)))
    clinical_feat = model.clinical_norm(model.clinical_encoder(clinical))
    return img_feat, signal_feat, clinical_feat


def get_embedding_batches(model, loader, device, max_samples_per_class=50):
    model.eval()
    embeddings, normal_samples, abnormal_samples = [], 0, 0
    with torch.no_grad():
        for *batch, _index in loader:
            images, ecg_signals, clinical, labels = (t.to(device) for t in batch)
            fused, _ = model.attention_fusion(*modal_features(model, images, ecg_signals, clinical))
            embeddings.append(fused.cpu())
            for label in labels.tolist():
                if label == 0 and normal_samples < max_samples_per_class:
                    normal_samples += 1
                elif label == 1 and abnormal_samples < max_samples_per_class:
                    abnormal_samples += 1
            if normal_samples >= max_samples_per_class and abnormal_samples >= max_samples_per_class:
                break
    return torch.cat(embeddings, dim=0)


def expected_gradients(fusion_model, background, x, nsamples=64, seed=0):
    """-> float32 [B, D, num_classes]; sums over D to f(x) - E_bg[f] in expectation (completeness)."""
    g = torch.Generator().manual_seed(seed)
    B, D = x.shape
    bi = torch.randint(0, background.shape[0], (nsamples, B), generator=g)
    alpha = torch.rand(nsamples, B, 1, generator=g).to(x.device)
    out = None
    # attributions need d logits / d input only: freeze the wrapped model for the duration, so that the HIP ops
    # neither compute weight gradients nor touch the parameters' .grad sinks (hip/functional.py grad_sink)
    frozen = [p for p in fusion_model.parameters() if p.requires_grad]
    for p in frozen:
        p.requires_grad_(False)
    try:
        for k in range(nsamples):
            base = background[bi[k].to(background.device)].to(x.device)
            point = (base + alpha[k] * (x - base)).detach().requires_grad_(True)
            logits = fusion_model(point)
            if out is None:
                out = torch.zeros(B, D, logits.shape[1], device=x.device)
            for c in range(logits.shape[1]):
                grad, = torch.autograd.grad(logits[:, c].sum(), point, retain_graph=c + 1 < logits.shape[1])
                out[:, :, c] += grad * (x - base)
    finally:
        for p in frozen:
            p.requires_grad_(True)
    return (out / nsamples).float()


def modality_contributions(shap_values, dims, labels):
    """rows of the reference's CSV (:166-190): mean |attribution| per modality block, as percentages"""
    n_img, n_signal, _ = dims
    vals = np.abs(shap_values.detach().cpu().numpy())
    rows = []
    for b in range(vals.shape[0]):
        for class_idx in range(vals.shape[2]):
            v = vals[b, :, class_idx]
            img, sig, clin = v[:n_img].mean(), v[n_img:n_img + n_signal].mean(), v[n_img + n_signal:].mean()
            total = img + sig + clin
            rows.append({"Image_%": img / total * 100, "Signal_%": sig / total * 100, "Clinical_%": clin / total * 100,
                         "Label": int(labels[b]), "Class": class_idx})
    return rows


def fusion_fc_chunk_norms(model):
    """norms of the image / signal / clinical column blocks of fusion_classifier[0].weight (:104-120)"""
    w = model.fusion_classifier[0].weight.detach().float().cpu().numpy()
    d = model.modal_dim
    return tuple(float(np.linalg.norm(w[:, i * d:(i + 1) * d])) for i in range(3))


def main(config=Config, model_path=None, max_samples_per_class=50, nsamples=64, out_csv=None, quiet=False):
    import pandas as pd
    device = torch.device(config.device)
    train_loader, _val_loader, test_loader = get_dataloaders(config)
    model = ECGMultimodalModel(config).to(device)
    if model_path:
        model.load_state_dict(torch.load(model_path, map_location=device))
    model.eval()
    fusion_model = FusionClassifierWrapper(model.fusion_classifier).to(device)
    bg = get_embedding_batches(model, train_loader, device, max_samples_per_class).to(device)
    results, soft_weights = [], None
    for *batch, _index in test_loader:
        images, ecg_signals, clinical, labels = (t.to(device) for t in batch)
        with torch.no_grad():
            feats = modal_features(model, images, ecg_signals, clinical)
            _fused, soft_weights = model.attention_fusion(*feats)
        x = torch.cat(feats, dim=1)     # the reference explains the un-weighted concatenation (:150)
        sv = expected_gradients(fusion_model, bg, x, nsamples=nsamples)
        for row in modality_contributions(sv, [f.shape[1] for f in feats], labels.tolist()):
            row["Sample_ID"] = len(results) // sv.shape[2] + 1
            results.append(row)
    df = pd.DataFrame(results)
    if out_csv:
        os_dir = __import__("os").path.dirname(out_csv)
        if os_dir:
            __import__("os").makedirs(os_dir, exist_ok=True)
        df.to_csv(out_csv, index=False)
    if not quiet:
        print(df.head())
        print("Fusion FC weight chunk norms (image, signal, clinical):", fusion_fc_chunk_norms(model))
        if soft_weights is not None:
            print("Attention weights (softmax):", [round(float(v), 4) for v in soft_weights.detach().cpu()])
    return df


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser(description="fusion-head modality attribution")
    ap.add_argument("--model_path", type=str, default=None)
    main(model_path=ap.parse_args().model_path)
