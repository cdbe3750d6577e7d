"""RNG-independent parameter / input fills shared by the oracle-side tests (TEST INFRASTRUCTURE ONLY).

``torch.manual_seed`` streams are not guaranteed identical across torch builds / machines, so
anything compared bit-tightly between this container (where the goldens are generated) and the
GPU box is filled from an integer hash evaluated in exact uint64 arithmetic (no libm involved).
"""
import math
import zlib

import numpy as np
import torch


def hash_uniform(n, salt):
    """n values in [-1, 1), a pure function of (index, salt); exact on every machine."""
    i = np.arange(n, dtype=np.uint64)
    off = (int(salt) * 0xBF58476D1CE4E5B9 + 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    x = i * np.uint64(0x9E3779B97F4A7C15) + np.uint64(off)
    x ^= x >> np.uint64(30)
    x *= np.uint64(0xBF58476D1CE4E5B9)
    x ^= x >> np.uint64(27)
    x *= np.uint64(0x94D049BB133111EB)
    x ^= x >> np.uint64(31)
    m = (x >> np.uint64(40)).astype(np.float64)          # 24 random bits
    return (m / float(1 << 23) - 1.0).astype(np.float32)  # exact in fp32


def _salt(name):
    return zlib.crc32(name.encode()) & 0x7FFFFFFF


def hash_tensor(shape, salt, scale=1.0):
    n = int(np.prod(shape)) if len(shape) else 1
    return torch.from_numpy(hash_uniform(n, salt) * np.float32(scale)).reshape(shape).clone()


@torch.no_grad()
def hash_fill_module(model, tag=""):
    """Fill every parameter and BN buffer of ``model`` from the hash (keyed by its state-dict name)."""
    for name, t in model.state_dict().items():
        s = _salt(tag + name)
        if name.endswith("num_batches_tracked"):
            t.zero_()
        elif name.endswith("running_mean"):
            t.copy_(hash_tensor(t.shape, s, 0.1))
        elif name.endswith("running_var"):
            t.copy_(1.0 + 0.2 * hash_tensor(t.shape, s).abs())
        elif t.dim() >= 3:                                   # conv weight: He-like scale
            fan_in = t[0].numel()
            t.copy_(hash_tensor(t.shape, s, math.sqrt(6.0 / fan_in)))
        elif t.dim() == 2:                                   # linear weight
            t.copy_(hash_tensor(t.shape, s, math.sqrt(3.0 / t.shape[1])))
        elif name.endswith("weight") and t.dim() == 1 and "attention_fusion.weights" not in name:
            t.copy_(1.0 + 0.1 * hash_tensor(t.shape, s))     # BN / LN gamma
        elif name.endswith("attention_fusion.weights"):
            t.copy_(1.0 + 0.5 * hash_tensor(t.shape, s))
        else:                                                # biases
            t.copy_(hash_tensor(t.shape, s, 0.1))
    return model


def synthetic_batch(batch, img_hw=(224, 224), sig_len=5000, clin_dim=16, num_classes=2, salt=1,
                    leads=None):
    """Deterministic stand-in for the reference's (image, signal, clinical, label) batch
    (dataset.py:53-74): image in [-1,1] like Normalize(0.5,0.5), unit-scale signal/clinical."""
    h, w = img_hw
    image = hash_tensor((batch, 3, h, w), 1000 + salt)
    if leads is None:
        signal = hash_tensor((batch, sig_len), 2000 + salt, 1.7)
    else:
        signal = hash_tensor((batch, leads, sig_len), 2000 + salt, 1.7)
    clinical = hash_tensor((batch, clin_dim), 3000 + salt, 1.7)
    lab = (hash_uniform(batch, 4000 + salt) > 0).astype(np.int64) % num_classes
    return image, signal, clinical, torch.from_numpy(lab)
