"""Data surface of the reference's dataset.py (ECGMultimodalDataset :15-116, get_dataloaders :118-213).

The reference reads a private hospital dataset (./data, git-ignored) that ships with neither repo, so
the default here is a synthetic generator with the same tuple arity, dtypes, shapes and value ranges:
  image    float32 [3, H, W] in [-1, 1]   (Resize -> ToTensor -> Normalize(0.5, 0.5), :119-123)
  signal   float32 [L]  ~ zero-mean, unit-variance, low-passed (StandardScaler + filtfilt, :66-71)
  clinical float32 [D]  ~ standardised                          (:72)
  label    int64 scalar, index (int)  -> a 5-tuple like :74
Real-file loading (xlsx/csv/JPEG + scipy filtering) is SURVEY 8(f1) "next".
"""
import torch
from torch.utils.data import DataLoader, Dataset


class ECGMultimodalDataset(Dataset):
    def __init__(self, size, config, seed=0, return_index=True):
        self.size, self.config, self.seed, self.return_index = int(size), config, int(seed), return_index
        self.hw = (config.img_height, config.img_width)
        self.sig_len = getattr(config, "signal_length", 5000)
        self.clin_dim = getattr(config, "clinical_input_dim", 24)
        self.num_classes = config.num_classes

    def __len__(self):
        return self.size

    def __getitem__(self, idx):
        g = torch.Generator().manual_seed(self.seed * 1000003 + int(idx))
        label = torch.randint(0, self.num_classes, (), generator=g)
        image = torch.randn(3, *self.hw, generator=g).clamp_(-1, 1)
        sig = torch.randn(self.sig_len, generator=g)
        # a weak class-dependent component so that training has something to fit
        t = torch.arange(self.sig_len, dtype=torch.float32)
        sig = sig + 0.5 * float(label) * torch.sin(t * (2 * 3.14159265 / 250.0))
        clinical = torch.randn(self.clin_dim, generator=g) + 0.3 * float(label)
        image[0, : self.hw[0] // 4] += 0.2 * float(label)
        item = (image.clamp_(-1, 1), sig, clinical, label.to(torch.long))
        return item + (int(idx),) if self.return_index else item


def get_dataloaders(config):
    """-> (train_loader, val_loader, test_loader); the loops unpack ``*batch, index``."""
    bs, nw = config.batch_size, getattr(config, "num_workers", 0)
    sizes = (getattr(config, "synthetic_train_size", 256), getattr(config, "synthetic_val_size", 32),
             getattr(config, "synthetic_test_size", 32))
    if not getattr(config, "synthetic", True):
        raise NotImplementedError("real-file loading (labels.xlsx / clinical.csv / ecg_signals.csv / JPEGs) is not "
                                  "built yet (SURVEY 8f1); set Config.synthetic = True")
    ds = [ECGMultimodalDataset(n, config, seed=config.seed + i) for i, n in enumerate(sizes)]
    # drop_last on train: BatchNorm needs > 1 sample per batch (PMB:258)
    return (DataLoader(ds[0], batch_size=bs, shuffle=True, num_workers=nw, drop_last=True),
            DataLoader(ds[1], batch_size=bs, shuffle=False, num_workers=nw),
            DataLoader(ds[2], batch_size=bs, shuffle=False, num_workers=nw))


def get_testloader(config):
    return get_dataloaders(config)[2]
