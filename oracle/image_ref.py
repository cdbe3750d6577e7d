"""ORACLE (test infrastructure only): numpy restatement of the reference's image transform
    transforms.Resize((H, W)) -> transforms.ToTensor() -> transforms.Normalize([0.5]*3, [0.5]*3)
(dataset.py:119-123, dataset_image.py:67-70 (no Resize), train_image_only.py:58-62) applied to the
``Image.open(...).convert('RGB')`` picture (dataset.py:61).

Third-party arithmetic: torchvision (absent from /root/reference and from this image; unpinned in the
reference's README) hands a PIL image to ``PIL.Image.resize(size[::-1], BILINEAR)``.  Pillow's published
algorithm (src/libImaging/Resample.c, 8 bits per channel): two passes, horizontal then vertical, each a
convolution with a triangle filter whose support is stretched by the down-scaling factor (antialiasing),
coefficients normalised in double, quantised to 22 fractional bits, integer accumulation started at
2^21, arithmetic shift right by 22, saturation to 0..255 -- the intermediate image is uint8.
Pinned: oracle/make_golden.py checks this restatement bit-for-bit against the installed Pillow (12.2.0)
and stores Pillow's own outputs in tests/golden/g8_image.npz.
"""
import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _coeffs(in_size, out_size):
    """Pillow precompute_coeffs + normalize_coeffs_8bpc for the bilinear filter over the box [0, in_size]."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = np.array([max(0.0, 1.0 - abs((x + xmin - center + 0.5) * ss)) for x in range(xmax)], np.float64)
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = w / ww
        q = np.where(w < 0, -0.5 + w * (1 << PRECISION_BITS), 0.5 + w * (1 << PRECISION_BITS))
        kk[xx, :xmax] = q.astype(np.int64)  # C (int) cast truncates toward zero; all terms here are >= 0
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img, bounds, kk, axis):
    """one resampling pass of a uint8 [H, W, C] picture along ``axis`` (1 = horizontal, 0 = vertical)"""
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((bounds.shape[0],) + src.shape[1:], np.uint8)
    for i, (lo, n) in enumerate(bounds):
        acc = np.tensordot(kk[i, :n].astype(np.int64), src[lo:lo + n], axes=(0, 0)) + (1 << (PRECISION_BITS - 1))
        out[i] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bilinear_u8(img, out_h, out_w):
    """img uint8 [H, W, 3] -> uint8 [out_h, out_w, 3], == PIL.Image.resize((out_w, out_h), BILINEAR)"""
    H, W = img.shape[:2]
    if (H, W) == (out_h, out_w):
        return img.copy()
    hb, hk = _coeffs(W, out_w)
    vb, vk = _coeffs(H, out_h)
    if W != out_w:
        first, last = vb[0, 0], vb[-1, 0] + vb[-1, 1]
        tmp = _pass(img[first:last], hb, hk, 1)
        vb = vb.copy()
        vb[:, 0] -= first
        img = tmp
    if H != out_h:
        img = _pass(img, vb, vk, 0)
    return img


def to_tensor_normalize(img_u8, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)):
    """ToTensor (uint8 HWC -> float32 CHW / 255) then Normalize, in float32 as torchvision does"""
    t = np.moveaxis(img_u8, -1, 0).astype(np.float32) / np.float32(255)
    m = np.asarray(mean, np.float32).reshape(3, 1, 1)
    s = np.asarray(std, np.float32).reshape(3, 1, 1)
    return ((t - m) / s).astype(np.float32)


def image_transform(img_u8, out_h=224, out_w=224):
    return to_tensor_normalize(resize_bilinear_u8(img_u8, out_h, out_w))


def synthetic_ecg_picture(h, w, salt):
    """formula-generated RGB test picture: paper-like background + grid + a dark trace + hash noise"""
    from . import fill
    yy, xx = np.mgrid[0:h, 0:w]
    base = 235 + 12 * np.sin(xx / 53.0 + salt) * np.cos(yy / 17.0)
    grid = ((xx % 25 == 0) | (yy % 25 == 0)) * -60.0
    trace_y = h / 2 + h / 3 * np.sin(xx / 40.0 + salt) * np.exp(-((xx % 300) - 150.0) ** 2 / 2000.0)
    trace = (np.abs(yy - trace_y) < 1.5) * -200.0
    noise = fill.hash_uniform(h * w * 3, 4000 + salt).reshape(h, w, 3) * 25.0
    img = (base + grid + trace)[..., None] * np.array([1.0, 0.93, 0.9]) + noise
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)
