"""Image-only ResNet18 classifier and its training loop (reference: train_image_only.py:92-310;
BASELINE config 2).  ``ImageOnlyClassifier`` keeps the ``image_encoder.*`` state_dict prefix."""
import os
import time

import torch
import torch.nn as nn
from torch.utils.data import DataLoader, Dataset

from .config import Config
from .hip import functional as HF
from .hip import nn as hnn
from .image_encoder import resnet18
from .optim import FusedAdam


class ImageOnlyClassifier(nn.Module):
    def __init__(self, compute_dtype=None, pretrained_state_dict=None):
        super().__init__()
        cd = compute_dtype or getattr(Config, "compute_dtype", "bf16")
        # the reference fetches IMAGENET1K_V1 weights here (:95) -- offline: random init, or a local state_dict
        self.image_encoder = resnet18(compute_dtype=cd)
        if pretrained_state_dict is not None:
            self.image_encoder.load_state_dict(pretrained_state_dict, strict=False)
        self.image_encoder.fc = hnn.Linear(self.image_encoder.fc.in_features, Config.num_classes)

    def forward(self, x):
        return self.image_encoder(x)


class ImageOnlyDataset(Dataset):
    """Synthetic stand-in for train_image_only.py:20-41 (image, label)."""

    def __init__(self, size, seed=0):
        self.size, self.seed = size, seed

    def __len__(self):
        return self.size

    def __getitem__(self, idx):
        g = torch.Generator().manual_seed(self.seed * 1000003 + idx)
        label = torch.randint(0, Config.num_classes, (), generator=g)
        img = torch.randn(3, Config.img_height, Config.img_width, generator=g).clamp_(-1, 1)
        img[0, : Config.img_height // 4] += 0.3 * float(label)
        return img.clamp_(-1, 1), label.to(torch.long)


def get_imageonly_dataloaders():
    mk = lambda n, s, sh: DataLoader(ImageOnlyDataset(n, s), batch_size=Config.batch_size, shuffle=sh, drop_last=sh)
    return (mk(getattr(Config, "synthetic_train_size", 256), Config.seed, True),
            mk(getattr(Config, "synthetic_val_size", 32), Config.seed + 1, False),
            mk(getattr(Config, "synthetic_test_size", 32), Config.seed + 2, False))


def main(num_epochs=None, quiet=False):
    torch.manual_seed(Config.seed)
    HF.manual_seed(Config.seed)
    device = torch.device(Config.device)
    train_loader, val_loader, test_loader = get_imageonly_dataloaders()
    model = ImageOnlyClassifier().to(device)
    optimizer = FusedAdam(model.parameters(), lr=Config.lr)          # :111
    modeltime = time.strftime("%m%d_%H%M%S", time.localtime())
    ckpt_dir = os.path.join(Config.checkpoint_dir, modeltime)
    os.makedirs(ckpt_dir, exist_ok=True)
    min_val, stop_ctr, hist = float("inf"), 0, []
    for epoch in range(num_epochs or Config.num_epochs):
        model.train()
        tl, correct, total = 0.0, 0, 0
        for images, labels in train_loader:
            images, labels = images.to(device), labels.to(device)
            optimizer.zero_grad()
            out = model(images)
            loss = HF.cross_entropy(out, labels)                      # nn.CrossEntropyLoss(), :110
            loss.backward()
            optimizer.step()
            tl += loss.item()
            correct += out.argmax(1).eq(labels).sum().item()
            total += labels.size(0)
        model.eval()
        vl, vc, vt = 0.0, 0, 0
        with torch.no_grad():
            for images, labels in val_loader:
                images, labels = images.to(device), labels.to(device)
                out = model(images)
                vl += HF.cross_entropy(out, labels).item()
                vc += out.argmax(1).eq(labels).sum().item()
                vt += labels.size(0)
        avg_val = vl / max(len(val_loader), 1)
        hist.append((tl / max(len(train_loader), 1), correct / max(total, 1), avg_val, vc / max(vt, 1)))
        if not quiet:
            print(f"epoch {epoch + 1}: train {hist[-1][0]:.4f}/{hist[-1][1]:.3f} val {avg_val:.4f}/{hist[-1][3]:.3f}")
        torch.save(model.state_dict(), os.path.join(ckpt_dir, "last.pth"))
        if avg_val < min_val:
            min_val, stop_ctr = avg_val, 0
            torch.save(model.state_dict(), os.path.join(ckpt_dir, "best.pth"))
        else:
            stop_ctr += 1
            if stop_ctr >= Config.patience:
                break
    return hist, ckpt_dir


if __name__ == "__main__":
    main()
