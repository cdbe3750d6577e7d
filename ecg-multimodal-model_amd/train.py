"""Multimodal training entry point (reference: train.py:22-336; ``train_paper_modal_balance`` wraps it for the
paper_modal_balance model).  Like the reference's train.py:14 it builds ``multimodal.ECGMultimodalModel`` (TabNet
clinical branch, widths 512 / 128 / 32) unless another ``model_cls`` is passed.

Reproduces the reference loop: seed 42; encoders frozen by default as train.py:35-40 does (pass
``freeze_encoders=False`` for the end-to-end step of train_kfold.py:42); Adam(lr=Config.lr) on the
trainable parameters; per step zero_grad -> forward -> CE(fusion_logits) + 0.1 * var_loss ->
backward -> step; accuracy from fusion_logits.argmax; validation under no_grad; last/best/epochN.pth
state_dict checkpoints under checkpoints/<MMDD_HHMMSS>/; LR / 10 after 2 non-improving epochs; early
stop after ``Config.patience``; final test with softmax[:, 1], accuracy / F1 / AUC.
The reference's 4-vs-5 tuple unpacking defect (train.py:60 vs dataset.py:74) is not replicated:
loops unpack ``*batch, index``.  TensorBoard is optional (not installed in this image).
"""
import os
import time

import numpy as np
import torch

from .config import Config
from .dataset import get_dataloaders
from .hip import functional as HF
from .multimodal import ECGMultimodalModel      # train.py:14
from .optim import FusedAdam


def _writer(logdir):
    try:
        from torch.utils.tensorboard import SummaryWriter
        return SummaryWriter(logdir)
    except Exception:
        return None


def step_loss(outputs, labels):
    """train.py:69-78: total = CE(fusion_logits) + 0.1 * var_loss (branch CE terms are computed by the
    reference but not part of total_loss)."""
    return HF.cross_entropy_plus(outputs[3], labels, outputs[4], 0.1)


def run_epoch(model, loader, device, optimizer=None):
    train = optimizer is not None
    model.train(train)
    tot, tot_var, correct, n = 0.0, 0.0, 0, 0
    ctx = torch.enable_grad() if train else torch.no_grad()
    with ctx:
        for *batch, _index in loader:
            images, ecg, clinical, labels = (t.to(device) for t in batch)
            if train:
                optimizer.zero_grad()
            outputs = model(images, ecg, clinical)
            loss = step_loss(outputs, labels)
            if train:
                loss.backward()
                optimizer.step()
            tot += loss.item()
            tot_var += outputs[4].item()
            correct += outputs[3].argmax(1).eq(labels).sum().item()
            n += labels.size(0)
    k = max(len(loader), 1)
    return tot / k, tot_var / k, correct / max(n, 1)


def evaluate(model, loader, device):
    """train.py:173-260: softmax[:,1] probabilities, accuracy / F1 / AUC on the test split."""
    from sklearn.metrics import f1_score, roc_auc_score
    model.eval()
    y_true, y_prob, y_pred = [], [], []
    with torch.no_grad():
        for *batch, _index in loader:
            images, ecg, clinical, labels = (t.to(device) for t in batch)
            logits = model(images, ecg, clinical)[3]
            prob = torch.softmax(logits.float().cpu(), dim=1)   # host-side metrics, as in the reference
            y_true += labels.cpu().tolist()
            y_prob += prob[:, 1].tolist()
            y_pred += prob.argmax(1).tolist()
    acc = float(np.mean(np.array(y_true) == np.array(y_pred)))
    f1 = float(f1_score(y_true, y_pred, zero_division=0))
    try:
        auc = float(roc_auc_score(y_true, y_prob))
    except ValueError:
        auc = float("nan")
    return {"accuracy": acc, "f1": f1, "auc": auc}


def main(config=Config, freeze_encoders=True, num_epochs=None, quiet=False, model_cls=None):
    torch.manual_seed(config.seed)
    HF.manual_seed(config.seed)
    device = torch.device(config.device)
    if not quiet:
        print(f"Using device: {device}")
    model = (model_cls or ECGMultimodalModel)(config).to(device)
    if getattr(config, "synthetic", True):   # synthetic batches carry the clinical width the model expects
        config = type(config.__name__, (config,), {"clinical_input_dim": model.get_clinical_feature_dim()})
    train_loader, val_loader, test_loader = get_dataloaders(config)

    if freeze_encoders:  # train.py:35-40
        for enc in (model.image_encoder, model.signal_encoder, model.clinical_encoder):
            for p in enc.parameters():
                p.requires_grad = False
    optimizer = FusedAdam(filter(lambda p: p.requires_grad, model.parameters()), lr=config.lr)

    modeltime = time.strftime("%m%d_%H%M%S", time.localtime())
    writer = _writer(f"runs/{modeltime}")
    ckpt_dir = os.path.join(config.checkpoint_dir, modeltime)
    os.makedirs(ckpt_dir, exist_ok=True)

    min_val, early, lr_ctr, history = float("inf"), 0, 0, []
    for epoch in range(num_epochs or config.num_epochs):
        tr_loss, _, tr_acc = run_epoch(model, train_loader, device, optimizer)
        va_loss, va_var, va_acc = run_epoch(model, val_loader, device)
        history.append(dict(epoch=epoch + 1, train_loss=tr_loss, train_acc=tr_acc, val_loss=va_loss, val_acc=va_acc))
        if not quiet:
            print(f"[{epoch + 1}] train {tr_loss:.4f}/{tr_acc:.4f}  val {va_loss:.4f}/{va_acc:.4f}")
        if writer is not None:
            writer.add_scalar("Loss/Train", tr_loss, epoch)
            writer.add_scalar("Loss/Val", va_loss, epoch)
            writer.add_scalar("VarLoss/Val", va_var, epoch)
            writer.add_scalar("Accuracy/Train", tr_acc, epoch)
            writer.add_scalar("Accuracy/Val", va_acc, epoch)
            w = torch.softmax(model.attention_fusion.weights.detach().float().cpu(), 0)
            for name, v in zip(("Image", "Signal", "Clinical"), w.tolist()):
                writer.add_scalar(f"AttentionWeights/{name}", v, epoch)
        torch.save(model.state_dict(), os.path.join(ckpt_dir, "last.pth"))
        if va_loss < min_val:   # train.py:145-151: epochN.pth and best.pth only on improvement
            min_val, early, lr_ctr = va_loss, 0, 0
            torch.save(model.state_dict(), os.path.join(ckpt_dir, f"epoch{epoch + 1}.pth"))
            torch.save(model.state_dict(), os.path.join(ckpt_dir, "best.pth"))
        else:
            early += 1
            lr_ctr += 1
            if lr_ctr >= 2:  # manual LR / 10 (train.py:157-163)
                for g in optimizer.param_groups:
                    g["lr"] /= 10
                lr_ctr = 0
            if early >= config.patience:
                break
    results = {}
    for tag in ("best", "last"):
        model.load_state_dict(torch.load(os.path.join(ckpt_dir, f"{tag}.pth"), map_location=device))
        results[tag] = evaluate(model, test_loader, device)
        if not quiet:
            print(f"test[{tag}]: {results[tag]}")
    if writer is not None:
        writer.close()
    return history, results, ckpt_dir


if __name__ == "__main__":
    main()
