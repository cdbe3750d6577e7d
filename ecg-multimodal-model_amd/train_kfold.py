"""Nested stratified cross-validation harness (reference: train_kfold.py:20-181, dataset_kfold.py).

Outer ``Config.k_outer`` folds for testing, inner ``Config.k_inner`` folds for model selection; every inner run is the
reference's loop (Adam(lr), CE on the fusion logits, best_inner.pth on the lowest validation loss, LR / 10 after two
non-improving epochs, early stop after ``Config.patience``); the outer fold's test AUC is taken with best_inner.pth,
as the reference does (:163-171, i.e. the checkpoint of the LAST inner fold that improved).  Scalers are fitted on the
training part of each split (:25-29, :156-160).  Works on the synthetic generator (default) and on the reference's
file layout (``Config.synthetic = False``: raw items, device-side transforms through DeviceLoader).
The reference unpacks the model output as one tensor (:58-59) although the model returns a tuple; here the fusion
logits (output [3]) are what the criterion and the AUC see.
"""
import os
import time

import numpy as np
import torch
from torch.utils.data import DataLoader, Subset

from . import dataset as D
from .config import Config
from .hip import functional as HF
from .multimodal_paper_modal_balance import ECGMultimodalModel
from .optim import FusedAdam


def get_all_data(config):
    """-> (context, labels[np.int64], positions) ; positions index the label array (dataset_kfold.get_all_data)"""
    if getattr(config, "synthetic", True):
        n = sum(getattr(config, k, d) for k, d in (("synthetic_train_size", 256), ("synthetic_val_size", 32),
                                                   ("synthetic_test_size", 32)))
        ds = D.SyntheticECGMultimodalDataset(n, config, seed=config.seed)
        labels = np.array([int(ds[i][3]) for i in range(n)], dtype=np.int64)
        return {"dataset": ds}, labels, np.arange(n)
    labels_df, ecg_signals, clinical_df = D.load_tables(config)
    return ({"labels_df": labels_df, "ecg_signals": ecg_signals, "clinical_df": clinical_df},
            labels_df["label"].values.astype(np.int64), np.arange(len(labels_df)))


def make_loader(config, ctx, positions, fit_positions, shuffle):
    """loader over ``positions``; scalers fitted on ``fit_positions`` (file mode)"""
    bs, nw = config.batch_size, getattr(config, "num_workers", 0)
    if "dataset" in ctx:
        return DataLoader(Subset(ctx["dataset"], [int(i) for i in positions]), batch_size=bs, shuffle=shuffle,
                          num_workers=nw, drop_last=shuffle and len(positions) % bs == 1)
    from sklearn.preprocessing import StandardScaler
    ldf, ecg, clin = ctx["labels_df"], ctx["ecg_signals"], ctx["clinical_df"]
    ids = ldf.iloc[positions]["index"].tolist()
    fit_ids = ldf.iloc[fit_positions]["index"].tolist()
    ecg_scaler = StandardScaler().fit(ecg.loc[ecg.index.isin(fit_ids)])
    clinical_scaler = StandardScaler().fit(clin[clin["index"].isin(fit_ids)][D.CLINICAL_NUMERIC_COLS])
    ds = D.ECGMultimodalDataset(ids, ldf, ecg, clin, ecg_scaler, clinical_scaler, D._Transform(config))
    raw = DataLoader(ds, batch_size=bs, shuffle=shuffle, num_workers=nw, pin_memory=True)
    return D.DeviceLoader(raw, D.DeviceInputPipeline(config, ecg_scaler))


def _epoch(model, loader, device, optimizer=None):
    train = optimizer is not None
    model.train(train)
    tot, correct, n = 0.0, 0, 0
    with (torch.enable_grad() if train else torch.no_grad()):
        for *batch, _index in loader:
            images, ecg, clinical, labels = (t.to(device) for t in batch)
            if train:
                optimizer.zero_grad()
            logits = model(images, ecg, clinical)[3]
            loss = HF.cross_entropy(logits, labels)
            if train:
                loss.backward()
                optimizer.step()
            tot += loss.item()
            correct += logits.argmax(1).eq(labels).sum().item()
            n += labels.size(0)
    return tot / max(len(loader), 1), correct / max(n, 1)


def train_inner(config, ctx, train_pos, val_pos, fold_dir, num_epochs=None, quiet=True):
    device = torch.device(config.device)
    train_loader = make_loader(config, ctx, train_pos, train_pos, True)
    val_loader = make_loader(config, ctx, val_pos, train_pos, False)
    model = ECGMultimodalModel(config).to(device)
    optimizer = FusedAdam(model.parameters(), lr=config.lr)
    min_val, early, lr_ctr, best = float("inf"), 0, 0, None
    for epoch in range(num_epochs or config.num_epochs):
        tr_loss, tr_acc = _epoch(model, train_loader, device, optimizer)
        va_loss, va_acc = _epoch(model, val_loader, device)
        if not quiet:
            print(f"  [{epoch + 1}] train {tr_loss:.4f}/{tr_acc:.3f}  val {va_loss:.4f}/{va_acc:.3f}")
        if va_loss < min_val:
            best = os.path.join(fold_dir, "best_inner.pth")
            torch.save(model.state_dict(), best)
            min_val, early, lr_ctr = va_loss, 0, 0
        else:
            early += 1
            lr_ctr += 1
            if lr_ctr >= 2:
                for g in optimizer.param_groups:
                    g["lr"] /= 10
                lr_ctr = 0
            if early >= config.patience:
                break
    return min_val, best


def test_outer(model, test_loader, device):
    from sklearn.metrics import roc_auc_score
    model.eval()
    labels_all, probs_all = [], []
    with torch.no_grad():
        for *batch, _index in test_loader:
            images, ecg, clinical, labels = (t.to(device) for t in batch)
            probs = torch.softmax(model(images, ecg, clinical)[3].float().cpu(), dim=1)[:, 1]
            probs_all += probs.tolist()
            labels_all += labels.cpu().tolist()
    try:
        return float(roc_auc_score(labels_all, probs_all))
    except ValueError:   # a fold with one class only
        return float("nan")


def main(config=Config, num_epochs=None, quiet=False):
    from sklearn.model_selection import StratifiedKFold
    torch.manual_seed(config.seed)
    HF.manual_seed(config.seed)
    ctx, labels, positions = get_all_data(config)
    device = torch.device(config.device)
    ckpt = os.path.join(config.checkpoint_dir, time.strftime("%m%d_%H%M%S", time.localtime()))
    outer = StratifiedKFold(n_splits=config.k_outer, shuffle=True, random_state=config.seed)
    outer_aucs = []
    for of, (train_val_pos, test_pos) in enumerate(outer.split(positions, labels)):
        fold_dir = os.path.join(ckpt, f"outer_fold_{of + 1}")
        os.makedirs(fold_dir, exist_ok=True)
        inner = StratifiedKFold(n_splits=config.k_inner, shuffle=True, random_state=config.seed)
        for inf, (itr, iva) in enumerate(inner.split(train_val_pos, labels[train_val_pos])):
            if not quiet:
                print(f"outer {of + 1}/{config.k_outer}  inner {inf + 1}/{config.k_inner}")
            train_inner(config, ctx, train_val_pos[itr], train_val_pos[iva], fold_dir, num_epochs, quiet)
        model = ECGMultimodalModel(config).to(device)
        model.load_state_dict(torch.load(os.path.join(fold_dir, "best_inner.pth"), map_location=device))
        auc = test_outer(model, make_loader(config, ctx, test_pos, train_val_pos, False), device)
        outer_aucs.append(auc)
        if not quiet:
            print(f"outer fold {of + 1} AUC: {auc:.4f}")
    if not quiet:
        print(f"Mean AUC: {np.nanmean(outer_aucs):.4f}")
    return outer_aucs


if __name__ == "__main__":
    main()
