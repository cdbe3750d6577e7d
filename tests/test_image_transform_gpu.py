"""SURVEY 8(f1), image half: the device transform is bit-identical to Pillow's BILINEAR resize (golden g8 = Pillow's
own outputs) followed by float32 ToTensor / Normalize -- integer work, so the bar is exact equality."""
import numpy as np
import pytest
import torch

from ecgmm import image_transform as IT
from oracle import image_ref as IR

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_matches_pillow_golden_bit_for_bit(golden_dir):
    g8 = np.load(f"{golden_dir}/g8_image.npz")
    for i, (h, w, oh, ow) in enumerate(g8["cases"]):
        img = IR.synthetic_ecg_picture(int(h), int(w), 31 + i)
        out = IT.image_transform(torch.from_numpy(img).to(DEV), (int(oh), int(ow)))
        want = IR.to_tensor_normalize(g8[f"resized_{i}"])
        assert out.shape == (3, oh, ow) and out.dtype == torch.float32
        assert np.array_equal(out.cpu().numpy(), want), f"case {i}"


@pytest.mark.parametrize("case", [(5, 250, 2500, 224, 224), (3, 224, 224, 224, 224), (2, 250, 2500, 250, 2500),
                                  (4, 250, 2500, 125, 1250), (3, 64, 48, 224, 224), (2, 7, 9, 5, 4),
                                  (1, 1000, 31, 10, 30), (2, 300, 5000, 224, 224), (2, 40, 6000, 20, 100),
                                  (2, 100, 5461, 50, 224), (3, 30, 40, 30, 64)])
def test_batches_and_other_shapes_vs_oracle(case):
    B, h, w, oh, ow = case
    imgs = np.stack([IR.synthetic_ecg_picture(h, w, 70 + b) for b in range(B)])
    tf = IT.Compose([IT.Resize((oh, ow)), IT.ToTensor(), IT.Normalize([0.5] * 3, [0.5] * 3)])
    out = tf(torch.from_numpy(imgs).to(DEV)).cpu().numpy()
    for b in range(B):
        assert np.array_equal(out[b], IR.image_transform(imgs[b], oh, ow)), f"picture {b}"


def test_unaligned_views_and_other_statistics():
    """a picture batch that does not start on a 16-byte boundary, and per-channel mean/std"""
    raw = np.concatenate([np.zeros(5, np.uint8), IR.synthetic_ecg_picture(60, 333, 3).reshape(-1)])
    dev = torch.from_numpy(raw).to(DEV)
    view = dev[5:].view(1, 60, 333, 3)
    assert view.data_ptr() % 16 != 0
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    out = IT.image_transform(view, (32, 100), mean, std).cpu().numpy()
    want = IR.to_tensor_normalize(IR.resize_bilinear_u8(raw[5:].reshape(60, 333, 3), 32, 100), mean, std)
    assert np.array_equal(out[0], want)


def test_feeds_the_model(golden_dir):
    """transform output is what ECGMultimodalModel.forward takes: [B, 3, 224, 224] float32 in [-1, 1]"""
    imgs = torch.from_numpy(np.stack([IR.synthetic_ecg_picture(250, 2500, b) for b in range(2)])).to(DEV)
    x = IT.image_transform(imgs, (224, 224))
    assert x.shape == (2, 3, 224, 224) and float(x.min()) >= -1.0 and float(x.max()) <= 1.0


def test_rejects_bad_arguments():
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        IT.image_transform(torch.zeros(1, 8, 8, 3, dtype=torch.uint8), (4, 4))
    with pytest.raises(ValueError):
        IT.image_transform(torch.zeros(1, 8, 8, 3, device=DEV), (4, 4))
    with pytest.raises(ValueError):
        IT.Compose([IT.Normalize([0.5] * 3, [0.5] * 3)])
