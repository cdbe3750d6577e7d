"""Data surface of the reference's dataset.py (ECGMultimodalDataset :15-116, get_dataloaders :118-213).

Two sources behind the same ``get_dataloaders(config)`` -> (train, val, test) loaders of
``(image, signal, clinical, label, index)`` batches:

* ``Config.synthetic = True`` (default): the reference reads a private hospital dataset (./data, git-ignored)
  that ships with neither repo, so this generator yields the same tuple arity, dtypes, shapes and ranges.
* ``Config.synthetic = False``: the reference's files (labels.xlsx|csv, clinical.csv, ecg_signals.csv,
  images/<idx>/<idx:03d>ECG_lead2.jpg) with the reference's filtering, index intersection, stratified 80/10/10
  split and train-split StandardScalers (:126-200).  The per-sample CPU work of the reference's workers --
  Resize/ToTensor/Normalize of the picture (:119-123) and StandardScaler + baseline removal + Butterworth
  filtfilt of the signal (:66-71, :81-95) -- is moved onto the GPU: workers only decode the JPEG
  and look rows up; ``DeviceLoader`` uploads the raw batch on a side HIP stream and runs the two HIP kernels
  (``image_transform``, ``preprocess_signal``) one batch ahead of the training step.
"""
import os

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

KNOWN_MISSING = {17, 23, 36, 43, 51, 62, 115, 158}   # dataset.py:145
CLINICAL_NUMERIC_COLS = ["AGE", "Wt"]                # dataset.py:27,196


class SyntheticECGMultimodalDataset(Dataset):
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


class ECGMultimodalDataset(Dataset):
    """Reference constructor (dataset.py:16-17).  Items are RAW: ``(uint8 [H, W, 3] picture, float32 [L] unscaled
    signal, float32 clinical, int64 label, index)``; the transform and the signal filtering run on the device in
    ``DeviceLoader`` (``transform`` only carries ``.config`` for the image directory, as in the reference :58-60).
    ``ECGMultimodalDataset.synthetic(size, config)`` builds the file-free stand-in."""

    def __init__(self, indices, labels_df, ecg_signals, clinical_df, ecg_scaler=None, clinical_scaler=None,
                 transform=None):
        self.transform = transform
        self.labels_df = labels_df[labels_df["index"].isin(indices)].reset_index(drop=True)
        self.ecg_signals = ecg_signals.loc[ecg_signals.index.isin(indices)]
        self.clinical_df = clinical_df[clinical_df["index"].isin(indices)].reset_index(drop=True)
        self.ecg_scaler, self.clinical_scaler = ecg_scaler, clinical_scaler
        self.clinical_numeric_scaler_cols = CLINICAL_NUMERIC_COLS
        if clinical_scaler is not None:   # :38-45: only the scaled numeric columns survive
            vals = clinical_scaler.transform(self.clinical_df[CLINICAL_NUMERIC_COLS])
            import pandas as pd
            self.clinical_scaled = pd.DataFrame(vals, index=self.clinical_df["index"], columns=CLINICAL_NUMERIC_COLS)
        else:
            self.clinical_scaled = self.clinical_df.drop(columns=["index"])

    @staticmethod
    def synthetic(size, config, seed=0, return_index=True):
        return SyntheticECGMultimodalDataset(size, config, seed, return_index)

    def __len__(self):
        return len(self.labels_df)

    def image_path(self, index):
        return os.path.join(self.transform.config.image_dir, str(index), f"{str(index).zfill(3)}ECG_lead2.jpg")

    def __getitem__(self, idx):
        from PIL import Image
        row = self.labels_df.iloc[idx]
        index, label = int(row["index"]), int(row["label"])
        picture = torch.from_numpy(np.asarray(Image.open(self.image_path(index)).convert("RGB")).copy())
        signal = torch.from_numpy(self.ecg_signals.loc[index].values.astype(np.float32))
        clinical = torch.tensor(np.asarray(self.clinical_scaled.loc[index].values, dtype=np.float64), dtype=torch.float)
        return picture, signal, clinical, torch.tensor(label, dtype=torch.long), index


class DeviceInputPipeline:
    """raw host batch -> model-ready device batch: H2D copies + image_transform + preprocess_signal."""

    def __init__(self, config, ecg_scaler=None, device=None):
        self.device = torch.device(device or config.device)
        self.size = (config.img_height, config.img_width)
        self.resize = getattr(config, "resize_images", True)   # dataset_image.py:67-70 trains at full resolution
        self.mean = None if ecg_scaler is None else torch.as_tensor(ecg_scaler.mean_, dtype=torch.float32, device=self.device)
        self.scale = None if ecg_scaler is None else torch.as_tensor(ecg_scaler.scale_, dtype=torch.float32, device=self.device)

    def __call__(self, batch):
        from .image_transform import image_transform
        from .preprocess import preprocess_signal
        picture, signal, clinical, label, index = batch
        signal = signal.to(self.device, non_blocking=True)
        if isinstance(picture, (list, tuple)):
            # pictures of several sizes in one batch (collate_raw): the reference resizes per sample before its
            # collate (dataset.py:119-123), here every size bucket is one kernel launch into the common output
            if not self.resize:
                raise ValueError("pictures of different sizes cannot be batched without Resize (dataset_image.py:67-70)")
            image = torch.empty(len(picture), 3, *self.size, dtype=torch.float32, device=self.device)
            buckets = {}
            for i, pic in enumerate(picture):
                buckets.setdefault(tuple(pic.shape), []).append(i)
            for idxs in buckets.values():
                grp = torch.stack([picture[i] for i in idxs]).to(self.device, non_blocking=True)
                image[torch.as_tensor(idxs, device=self.device)] = image_transform(grp, self.size)
        else:
            picture = picture.to(self.device, non_blocking=True)
            image = image_transform(picture, self.size if self.resize else None)
        signal = preprocess_signal(signal, scaler_mean=self.mean, scaler_scale=self.scale)
        return (image, signal, clinical.to(self.device, non_blocking=True), label.to(self.device, non_blocking=True),
                index)


class DeviceLoader:
    """Iterates a raw DataLoader and yields device-ready batches; batch i+1 is uploaded and pre-processed on a side
    HIP stream while the caller trains on batch i."""

    def __init__(self, loader, pipeline):
        self.loader, self.pipeline = loader, pipeline
        self.dataset, self.batch_size = loader.dataset, loader.batch_size
        self._stream = None

    def __len__(self):
        return len(self.loader)

    def _stage(self, raw):
        if self._stream is None:
            self._stream = torch.cuda.Stream(self.pipeline.device)
        with torch.cuda.stream(self._stream):
            out = self.pipeline(raw)
        return out, self._stream.record_event()

    def __iter__(self):
        pending = None
        for raw in self.loader:
            nxt = self._stage(raw)
            if pending is not None:
                yield self._release(pending)
            pending = nxt
        if pending is not None:
            yield self._release(pending)

    @staticmethod
    def _release(staged):
        out, ev = staged
        cur = torch.cuda.current_stream(out[0].device)
        cur.wait_event(ev)
        for t in out[:4]:
            t.record_stream(cur)
        return out


def collate_raw(items):
    """default collate, except that decoded pictures of different sizes stay a list (DeviceInputPipeline resizes
    them per size bucket; the reference's per-sample Resize accepts such data, dataset.py:119-123)"""
    from torch.utils.data import default_collate
    pics = [it[0] for it in items]
    rest = default_collate([tuple(it[1:]) for it in items])
    same = all(p.shape == pics[0].shape for p in pics)
    return (torch.stack(pics) if same else pics, *rest)


def _read_table(path):
    import pandas as pd
    if path.endswith((".xlsx", ".xls")):
        alt = os.path.splitext(path)[0] + ".csv"
        if not os.path.exists(path) and os.path.exists(alt):
            return pd.read_csv(alt)
        try:
            return pd.read_excel(path)
        except ImportError as e:   # openpyxl is not part of this image
            if os.path.exists(alt):
                return pd.read_csv(alt)
            raise RuntimeError(f"{path}: reading .xlsx needs openpyxl ({e}); export it to {alt}") from e
    return pd.read_csv(path)


def load_tables(config, label_map=None):
    """dataset.py:126-162: read, filter 'Borderline', map labels, align the four index sets."""
    import pandas as pd
    labels_df = _read_table(config.label_file)
    clinical_df = _read_table(config.clinical_file)
    if "ECG" in clinical_df.columns:
        clinical_df = clinical_df.drop("ECG", axis=1)
    ecg_signals = pd.read_csv(config.ecg_csv, index_col=0)
    labels_df = labels_df[labels_df["label"] != "Borderline"].copy()
    labels_df["label"] = labels_df["label"].map(label_map or {"Normal": 0, "Abnormal": 1})
    if "IDX" in clinical_df.columns:
        clinical_df = clinical_df.rename(columns={"IDX": "index"})
    labels_df["index"] = labels_df["index"].astype(int)
    clinical_df["index"] = clinical_df["index"].astype(int)
    ecg_signals.index = ecg_signals.index.astype(int)
    image_indices = {int(f) for f in os.listdir(config.image_dir) if f.isdigit()} - KNOWN_MISSING
    common = set(labels_df["index"]) & set(ecg_signals.index) & set(clinical_df["index"]) & image_indices
    labels_df = labels_df[labels_df["index"].isin(common)].reset_index(drop=True)
    ecg_signals = ecg_signals.loc[ecg_signals.index.isin(common)]
    clinical_df = clinical_df[clinical_df["index"].isin(common)].reset_index(drop=True)
    return labels_df, ecg_signals, clinical_df


def split_indices(labels_df, seed):
    """dataset.py:164-187: stratified 80 / 10 / 10 by label, random_state = Config.seed"""
    from sklearn.model_selection import train_test_split
    labels = labels_df["label"].values
    idx = np.arange(len(labels))
    train_idx, temp_idx, _, temp_y = train_test_split(idx, labels, test_size=0.2, stratify=labels, random_state=seed)
    val_idx, test_idx = train_test_split(temp_idx, test_size=0.5, stratify=temp_y, random_state=seed)
    return tuple(labels_df.iloc[i]["index"].tolist() for i in (train_idx, val_idx, test_idx))


class _Transform:
    """stands where the reference's transforms.Compose does: carries ``.config`` (dataset.py:124)"""

    def __init__(self, config):
        self.config = config


def build_file_datasets(config):
    from sklearn.preprocessing import StandardScaler
    labels_df, ecg_signals, clinical_df = load_tables(config)
    splits = split_indices(labels_df, config.seed)
    train_ecg = ecg_signals.loc[ecg_signals.index.isin(splits[0])]
    ecg_scaler = StandardScaler().fit(train_ecg)                                                   # :195
    clinical_scaler = StandardScaler().fit(clinical_df[clinical_df["index"].isin(splits[0])][CLINICAL_NUMERIC_COLS])
    tf = _Transform(config)
    sets = [ECGMultimodalDataset(s, labels_df, ecg_signals, clinical_df, ecg_scaler, clinical_scaler, tf) for s in splits]
    return sets, ecg_scaler, clinical_scaler


def get_dataloaders(config):
    """-> (train_loader, val_loader, test_loader); the loops unpack ``*batch, index``."""
    bs, nw = config.batch_size, getattr(config, "num_workers", 0)
    if getattr(config, "synthetic", True):
        sizes = (getattr(config, "synthetic_train_size", 256), getattr(config, "synthetic_val_size", 32),
                 getattr(config, "synthetic_test_size", 32))
        ds = [SyntheticECGMultimodalDataset(n, config, seed=config.seed + i) for i, n in enumerate(sizes)]
        # drop_last on train: BatchNorm needs > 1 sample per batch (PMB:258)
        return (DataLoader(ds[0], batch_size=bs, shuffle=True, num_workers=nw, drop_last=True),
                DataLoader(ds[1], batch_size=bs, shuffle=False, num_workers=nw),
                DataLoader(ds[2], batch_size=bs, shuffle=False, num_workers=nw))
    sets, ecg_scaler, _ = build_file_datasets(config)
    pipe = DeviceInputPipeline(config, ecg_scaler)
    raw = (DataLoader(sets[0], batch_size=bs, shuffle=True, num_workers=nw, pin_memory=True, collate_fn=collate_raw),
           DataLoader(sets[1], batch_size=bs, shuffle=False, num_workers=nw, pin_memory=True, collate_fn=collate_raw),
           DataLoader(sets[2], batch_size=bs, shuffle=False, num_workers=nw, pin_memory=True, collate_fn=collate_raw))
    return tuple(DeviceLoader(r, pipe) for r in raw)


def get_testloader(config, test_indices=None):
    """dataset.py:215-260: a loader over caller-chosen indices (scalers re-fitted on those rows, as the reference
    does); without indices, the test split of get_dataloaders."""
    if test_indices is None or getattr(config, "synthetic", True):
        return get_dataloaders(config)[2]
    from sklearn.preprocessing import StandardScaler
    labels_df, ecg_signals, clinical_df = load_tables(config)
    sub = ecg_signals.loc[ecg_signals.index.isin(test_indices)]
    ecg_scaler = StandardScaler().fit(sub)
    clinical_scaler = StandardScaler().fit(clinical_df[clinical_df["index"].isin(test_indices)][CLINICAL_NUMERIC_COLS])
    ds = ECGMultimodalDataset(test_indices, labels_df, ecg_signals, clinical_df, ecg_scaler, clinical_scaler,
                              _Transform(config))
    loader = DataLoader(ds, batch_size=config.batch_size, shuffle=False, num_workers=getattr(config, "num_workers", 0),
                        collate_fn=collate_raw)
    return DeviceLoader(loader, DeviceInputPipeline(config, ecg_scaler))
