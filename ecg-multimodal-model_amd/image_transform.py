"""The reference's image transform on the GPU:

    transforms.Compose([transforms.Resize((H, W)), transforms.ToTensor(), transforms.Normalize([0.5]*3, [0.5]*3)])

(dataset.py:119-123, train_image_only.py:58-62; dataset_image.py:67-70 omits the Resize).  The reference
applies it per sample on CPU workers to a ``PIL.Image``; here the decoded pictures of a whole batch
(uint8 [B, H, W, 3], e.g. ``torch.from_numpy(np.asarray(img))`` stacked) are transformed by one HIP
kernel, bit-identically to Pillow's antialiased bilinear resample + float32 ToTensor/Normalize.  JPEG
decoding stays on the host.  Same class names as the torchvision pieces the reference composes, so
``get_dataloaders``-style code reads unchanged; they operate on batched uint8 device tensors.
"""
import ctypes as C

import torch

from .hip import lib as L
from .hip.functional import _require_cuda, ptr, stream

_TABLES = {}


def _tables(H, W, OH, OW, device):
    key = (H, W, OH, OW, str(device))
    t = _TABLES.get(key)
    if t is None:
        lib = L.lib()
        nb = lib.ecgmm_image_resize_tables_bytes(H, W, OH, OW)
        host = torch.empty(nb // 4, dtype=torch.int32)
        L.check(lib.ecgmm_image_resize_tables(H, W, OH, OW, host.data_ptr(), nb), "image_resize_tables")
        t = _TABLES[key] = host.to(device)
    return t


def image_transform(images, size=None, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)):
    """images: CUDA uint8 [B, H, W, 3] (or [H, W, 3]); size=(out_h, out_w) or None for no resize.
    Returns float32 [B, 3, out_h, out_w] (or [3, out_h, out_w])."""
    _require_cuda(images, "image_transform")
    if images.dtype != torch.uint8 or images.shape[-1] != 3 or images.dim() not in (3, 4):
        raise ValueError("image_transform expects uint8 [B, H, W, 3] RGB pictures")
    single = images.dim() == 3
    x = (images.unsqueeze(0) if single else images).contiguous()
    B, H, W, _ = x.shape
    OH, OW = (H, W) if size is None else (int(size[0]), int(size[1]))
    out = torch.empty(B, 3, OH, OW, dtype=torch.float32, device=x.device)
    lib = L.lib()
    tab = None if (OH, OW) == (H, W) else _tables(H, W, OH, OW, x.device)
    m3, s3 = (C.c_float * 3)(*[float(v) for v in mean]), (C.c_float * 3)(*[float(v) for v in std])
    L.check(lib.ecgmm_image_transform(ptr(x), ptr(out), B, H, W, OH, OW, ptr(tab), 0 if tab is None else tab.numel() * 4,
                                      m3, s3, stream()), "image_transform")
    return out[0] if single else out


class Resize:
    def __init__(self, size):
        self.size = (size, size) if isinstance(size, int) else tuple(size)


class ToTensor:
    pass


class Normalize:
    def __init__(self, mean, std):
        self.mean, self.std = tuple(mean), tuple(std)


class Compose:
    """Compose([Resize((h, w)), ToTensor(), Normalize(mean, std)]) -> one fused device kernel."""

    def __init__(self, transforms):
        self.transforms = list(transforms)
        self.size, self.mean, self.std, seen_tensor = None, (0.0, 0.0, 0.0), (1.0, 1.0, 1.0), False
        for t in self.transforms:
            if isinstance(t, Resize) and not seen_tensor:
                self.size = t.size
            elif isinstance(t, ToTensor):
                seen_tensor = True
            elif isinstance(t, Normalize) and seen_tensor:
                self.mean, self.std = t.mean, t.std
            else:
                raise ValueError("Compose supports [Resize], ToTensor, [Normalize] in the reference's order")
        if not seen_tensor:
            raise ValueError("Compose needs a ToTensor stage")

    def __call__(self, images):
        return image_transform(images, self.size, self.mean, self.std)
