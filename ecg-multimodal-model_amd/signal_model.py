"""Signal-only model pieces with the reference's names (signal_model.py:12-184; identical class
copies live in train_signal_12_af.py:136-228 and 11 other reference scripts): ``SEBlock``,
``BasicBlock1D``, ``ResNet1D_SE``, ``FocalLoss``, ``ECGDataset``, ``find_best_threshold``,
``train_model``.  All arithmetic runs in the HIP library."""
import numpy as np
import torch
import torch.nn as nn
from torch.utils.data import Dataset

from .hip import functional as HF
from .multimodal_paper_modal_balance import BasicBlock1D, ResNet1D_SE, SEBlock  # noqa: F401
from .optim import FusedAdam


class FocalLoss(nn.Module):
    """alpha * (1 - exp(-CE))^gamma * CE (signal_model.py:91-106)."""

    def __init__(self, alpha=1.0, gamma=2.0, logits=True, reduce=True):
        super().__init__()
        self.alpha, self.gamma, self.logits, self.reduce = alpha, gamma, logits, reduce

    def forward(self, inputs, targets):
        if not self.logits:
            raise NotImplementedError("FocalLoss(logits=False): the reference never uses the nll_loss form")
        if not self.reduce:
            raise NotImplementedError("FocalLoss(reduce=False): only the mean-reduced loss runs on the HIP path")
        return HF.focal_loss(inputs, targets, self.alpha, self.gamma)


class CrossEntropyLoss(nn.Module):
    """nn.CrossEntropyLoss() as used at train.py:31 (mean reduction, no weights / smoothing)."""

    def forward(self, logits, targets):
        return HF.cross_entropy(logits, targets)


class ECGDataset(Dataset):
    """signal_model.py:109-116 verbatim semantics: arrays in, (X[i], y[i]) tensors out."""

    def __init__(self, X, y):
        self.X = torch.as_tensor(np.asarray(X), dtype=torch.float32)
        self.y = torch.as_tensor(np.asarray(y), dtype=torch.long)

    def __len__(self):
        return len(self.X)

    def __getitem__(self, idx):
        return self.X[idx], self.y[idx]


def find_best_threshold(y_true, y_prob):
    """F1-optimal threshold from arange(0.1, 0.9, 0.05) (signal_model.py:119-123)."""
    from sklearn.metrics import f1_score
    thresholds = np.arange(0.1, 0.9, 0.05)
    scores = [f1_score(y_true, (np.asarray(y_prob) >= t).astype(int)) for t in thresholds]
    return thresholds[int(np.argmax(scores))]


def train_model(model, train_loader, val_loader, epochs=30, lr=0.001, device=None):
    """FocalLoss + Adam + OneCycleLR stepped per batch (signal_model.py:155-184)."""
    device = device or next(model.parameters()).device
    criterion = FocalLoss(alpha=1, gamma=2)
    optimizer = FusedAdam(model.parameters(), lr=lr)
    scheduler = torch.optim.lr_scheduler.OneCycleLR(optimizer, max_lr=lr, steps_per_epoch=len(train_loader),
                                                    epochs=epochs)
    min_val_loss = np.inf
    history = []
    for epoch in range(epochs):
        model.train()
        for X_batch, y_batch in train_loader:
            optimizer.zero_grad()
            loss = criterion(model(X_batch.to(device)), y_batch.to(device))
            loss.backward()
            optimizer.step()
            scheduler.step()
        model.eval()
        val_loss = 0.0
        with torch.no_grad():
            for X_batch, y_batch in val_loader:
                val_loss += criterion(model(X_batch.to(device)), y_batch.to(device)).item()
        avg = val_loss / max(len(val_loader), 1)
        history.append(avg)
        if avg < min_val_loss:
            min_val_loss = avg
    return history
