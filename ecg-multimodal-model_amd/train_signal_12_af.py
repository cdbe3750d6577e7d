"""12-lead signal-only trainer (reference: train_signal_12_af.py:238-444; BASELINE config 5):
ResNet1D_SE(input_channels=12) + FocalLoss(1, 2) + Adam(1e-3) + OneCycleLR(max_lr=1e-3) stepped per
batch.  Data: synthetic [12, L] series (the reference's per-patient xlsx files are private)."""
import os
import time

import torch
from torch.utils.data import DataLoader, Dataset

from .config import Config
from .hip import functional as HF
from .optim import FusedAdam
from .signal_model import FocalLoss, ResNet1D_SE


class SignalOnlyDataset(Dataset):
    def __init__(self, size, leads=12, length=5000, seed=0):
        self.size, self.leads, self.length, self.seed = size, leads, length, seed

    def __len__(self):
        return self.size

    def __getitem__(self, idx):
        g = torch.Generator().manual_seed(self.seed * 1000003 + idx)
        label = torch.randint(0, 2, (), generator=g)
        x = torch.randn(self.leads, self.length, generator=g)
        t = torch.arange(self.length, dtype=torch.float32)
        x = x + 0.5 * float(label) * torch.sin(t * (2 * 3.14159265 / 200.0))
        return x, label.to(torch.long)


def get_signalonly_dataloaders(batch_size=8, length=5000):
    mk = lambda n, s, sh: DataLoader(SignalOnlyDataset(n, 12, length, s), batch_size=batch_size, shuffle=sh,
                                     drop_last=sh)
    return mk(128, 42, True), mk(32, 43, False), mk(32, 44, False)


def main(epochs=30, batch_size=8, length=5000, quiet=False):
    torch.manual_seed(42)
    HF.manual_seed(42)
    device = torch.device(Config.device)
    train_loader, val_loader, test_loader = get_signalonly_dataloaders(batch_size, length)
    model = ResNet1D_SE(input_channels=12, compute_dtype=getattr(Config, "compute_dtype", "bf16")).to(device)
    criterion = FocalLoss(alpha=1.0, gamma=2.0)
    optimizer = FusedAdam(model.parameters(), lr=0.001)
    scheduler = torch.optim.lr_scheduler.OneCycleLR(optimizer, max_lr=0.001, steps_per_epoch=len(train_loader),
                                                    epochs=epochs)
    ckpt_dir = os.path.join("./checkpoints", time.strftime("%m%d_%H%M%S"))
    os.makedirs(ckpt_dir, exist_ok=True)
    min_val, early, hist = float("inf"), 0, []
    for epoch in range(epochs):
        model.train()
        tl, correct, total = 0.0, 0, 0
        for signals, labels in train_loader:
            signals, labels = signals.to(device), labels.to(device)
            optimizer.zero_grad()
            out = model(signals)
            loss = criterion(out, labels)
            loss.backward()
            optimizer.step()
            scheduler.step()
            tl += loss.item()
            correct += out.argmax(1).eq(labels).sum().item()
            total += labels.size(0)
        model.eval()
        vl, vc, vt = 0.0, 0, 0
        with torch.no_grad():
            for signals, labels in val_loader:
                signals, labels = signals.to(device), labels.to(device)
                out = model(signals)
                vl += criterion(out, labels).item()
                vc += out.argmax(1).eq(labels).sum().item()   # accumulated (the reference's `=` at :295 is a bug)
                vt += labels.size(0)
        avg = vl / max(len(val_loader), 1)
        hist.append((tl / max(len(train_loader), 1), correct / max(total, 1), avg, vc / max(vt, 1)))
        if not quiet:
            print(f"epoch {epoch + 1}: train {hist[-1][0]:.4f}/{hist[-1][1]:.3f} val {avg:.4f}/{hist[-1][3]:.3f}")
        torch.save(model.state_dict(), os.path.join(ckpt_dir, "last.pth"))
        if avg < min_val:
            min_val, early = avg, 0
            torch.save(model.state_dict(), os.path.join(ckpt_dir, "best.pth"))
        else:
            early += 1
            if early >= 5:
                break
    return hist, ckpt_dir


if __name__ == "__main__":
    main()
