"""ORACLE (test infrastructure only): the reference's per-sample CPU input pipeline, restated --
ECGMultimodalDataset.__getitem__ (dataset.py:52-74) over get_dataloaders' tables/scalers (:118-200):
  picture: Image.open(...).convert('RGB') -> Resize((H, W)) -> ToTensor -> Normalize(.5, .5)       (:61-64, :119-123)
  signal : ecg_scaler.transform(all rows) in float64 -> preprocess_signal -> float32               (:30-35, :66-68)
  clinical: clinical_scaler.transform(AGE, Wt) -> float32                                          (:38-45, :72)
The reference module itself cannot be imported here (it imports torchvision, which this image lacks), so the
pieces are pinned separately: Pillow's resize by golden g8, preprocess_signal by golden g7 (the reference's own
function), the tables/splits by calling the same pandas / scikit-learn routines.
Also: ``write_tiny_dataset`` lays a small synthetic ./data tree out on disk in the reference's file layout.
"""
import os

import numpy as np

from . import image_ref as IR
from . import preprocess_ref as PR


def write_tiny_dataset(root, n=40, sig_len=600, hw=(250, 2500), seed=0):
    """labels.csv (xlsx needs openpyxl, absent here), clinical.csv, ecg_signals.csv, images/<i>/<i:03d>ECG_lead2.jpg"""
    import pandas as pd
    from PIL import Image
    rng = np.random.RandomState(seed)
    ids = [i for i in range(1, n + 8) if i not in (17, 23, 36)][:n]
    labels = ["Normal" if rng.rand() < 0.5 else "Abnormal" for _ in ids]
    labels[3] = "Borderline"                                     # filtered out (:131)
    os.makedirs(os.path.join(root, "images"), exist_ok=True)
    pd.DataFrame({"index": ids, "label": labels}).to_csv(os.path.join(root, "labels.csv"), index=False)
    pd.DataFrame({"IDX": ids, "AGE": rng.randint(20, 90, len(ids)), "Wt": np.round(rng.normal(65, 12, len(ids)), 1),
                  "SEX": rng.randint(0, 2, len(ids)), "ECG": ["x"] * len(ids)}).to_csv(os.path.join(root, "clinical.csv"), index=False)
    t = np.arange(sig_len)
    sig = np.stack([np.round(200 * np.sin(t / (9.0 + i % 5)) + 40 * rng.randn(sig_len) + 0.3 * t, 3) for i in ids])
    pd.DataFrame(sig, index=ids).to_csv(os.path.join(root, "ecg_signals.csv"))
    for i in ids[:-1]:                                           # the last subject has no picture folder -> dropped
        d = os.path.join(root, "images", str(i))
        os.makedirs(d, exist_ok=True)
        Image.fromarray(IR.synthetic_ecg_picture(hw[0], hw[1], i), "RGB").save(
            os.path.join(d, f"{str(i).zfill(3)}ECG_lead2.jpg"), quality=90)
    return ids


def reference_item(ds, idx, out_hw):
    """what the reference's ECGMultimodalDataset.__getitem__ returns for item ``idx`` of dataset ``ds`` (our raw
    dataset object is used only as the holder of the tables / scalers / paths)"""
    from PIL import Image
    row = ds.labels_df.iloc[idx]
    index = int(row["index"])
    pic = Image.open(ds.image_path(index)).convert("RGB")
    if out_hw is not None:
        pic = pic.resize((out_hw[1], out_hw[0]), Image.BILINEAR)
    image = IR.to_tensor_normalize(np.asarray(pic))
    scaled = ds.ecg_scaler.transform(ds.ecg_signals)             # float64, all rows, as :30-35
    row_pos = list(ds.ecg_signals.index).index(index)
    signal = PR.preprocess_signal(scaled[row_pos]).astype(np.float32)
    clinical = np.asarray(ds.clinical_scaled.loc[index].values, dtype=np.float64).astype(np.float32)
    return image, signal, clinical, int(row["label"]), index
