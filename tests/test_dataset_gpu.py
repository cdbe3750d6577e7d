"""the real-file input path end to end -- files on disk in the reference's layout -> raw DataLoader ->
DeviceLoader (H2D + image_transform + preprocess_signal on a side stream) == the reference's per-sample CPU pipeline
(oracle/dataset_ref.py: Pillow resize + float32 ToTensor/Normalize, sklearn scaler + scipy filtering)."""
import numpy as np
import pytest
import torch

from ecgmm import dataset as D
from ecgmm.config import Config
from oracle import dataset_ref as DR

pytestmark = pytest.mark.gpu


def _cfg(tmp_path, **kw):
    base = {"synthetic": False, "data_dir": str(tmp_path), "image_dir": str(tmp_path / "images"),
            "ecg_csv": str(tmp_path / "ecg_signals.csv"), "label_file": str(tmp_path / "labels.xlsx"),
            "clinical_file": str(tmp_path / "clinical.csv"), "batch_size": 4, "device": "cuda:0",
            "clinical_input_dim": 2, "checkpoint_dir": str(tmp_path / "ckpt")}
    base.update(kw)
    return type("Files", (Config,), base)


@pytest.mark.parametrize("resize", [True, False])
def test_device_loader_matches_reference_pipeline(tmp_path, resize):
    hw = (250, 2500) if resize else (64, 320)
    DR.write_tiny_dataset(str(tmp_path), n=28, sig_len=1000, hw=hw)
    cfg = _cfg(tmp_path, resize_images=resize)
    train, val, test = D.get_dataloaders(cfg)
    assert isinstance(test, D.DeviceLoader) and len(train.dataset) + len(val.dataset) + len(test.dataset) == 26
    out_hw = (cfg.img_height, cfg.img_width) if resize else None
    seen = 0
    for loader in (val, test):
        for bi, (image, signal, clinical, label, index) in enumerate(loader):
            assert image.is_cuda and signal.is_cuda and clinical.is_cuda and label.is_cuda
            torch.cuda.synchronize()
            for j in range(image.shape[0]):
                ri, rs, rc, rl, rx = DR.reference_item(loader.dataset, bi * cfg.batch_size + j, out_hw)
                assert int(index[j]) == rx and int(label[j]) == rl
                assert np.array_equal(image[j].cpu().numpy(), ri)                      # integer resample: exact
                assert np.abs(signal[j].cpu().numpy() - rs).max() < 2e-5 * max(1.0, np.abs(rs).max())
                assert np.allclose(clinical[j].cpu().numpy(), rc, atol=1e-6)
                seen += 1
    assert seen == len(val.dataset) + len(test.dataset)


def test_device_loader_buckets_pictures_of_different_sizes(tmp_path):
    """two JPEG sizes inside one batch: the reference's per-sample Resize accepts that (dataset.py:119-123)"""
    import os
    from PIL import Image
    from oracle import image_ref as IR
    ids = DR.write_tiny_dataset(str(tmp_path), n=28, sig_len=600, hw=(120, 900))
    for i in ids[:-1:2]:                        # every other subject gets a picture of another size
        path = os.path.join(str(tmp_path), "images", str(i), f"{str(i).zfill(3)}ECG_lead2.jpg")
        Image.fromarray(IR.synthetic_ecg_picture(90, 1100, i + 100), "RGB").save(path, quality=90)
    cfg = _cfg(tmp_path, resize_images=True)
    _, val, test = D.get_dataloaders(cfg)
    out_hw, seen, sizes = (cfg.img_height, cfg.img_width), 0, set()
    for loader in (val, test):
        for bi, (image, signal, clinical, label, index) in enumerate(loader):
            torch.cuda.synchronize()
            assert image.shape[1:] == (3, *out_hw)
            for j in range(image.shape[0]):
                ri, _, _, rl, rx = DR.reference_item(loader.dataset, bi * cfg.batch_size + j, out_hw)
                sizes.add(Image.open(loader.dataset.image_path(rx)).size)
                assert int(index[j]) == rx and int(label[j]) == rl and np.array_equal(image[j].cpu().numpy(), ri)
                seen += 1
    assert seen == len(val.dataset) + len(test.dataset) and len(sizes) == 2


def test_training_entry_point_runs_on_files(tmp_path):
    from ecgmm import train_paper_modal_balance as train
    DR.write_tiny_dataset(str(tmp_path), n=28, sig_len=1000, hw=(100, 1000))
    cfg = _cfg(tmp_path, num_epochs=2)
    history, results, _ = train.main(cfg, num_epochs=2, quiet=True)
    assert set(results["last"]) == {"accuracy", "f1", "auc"} and len(history) == 2
    assert np.isfinite(history[-1]["train_loss"])
