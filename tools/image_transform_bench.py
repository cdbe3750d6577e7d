"""f1 (image half) measurement: pictures/s of the device transform (2500x250 RGB -> 3x224x224 fp32) vs Pillow +
numpy ToTensor/Normalize on one host core (what a DataLoader worker does per sample in the reference)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
from ecgmm import image_transform as IT
from oracle import image_ref as IR
B, H, W, OH, OW = 256, 250, 2500, 224, 224
one = IR.synthetic_ecg_picture(H, W, 1)
imgs = torch.from_numpy(one).to("cuda:0").unsqueeze(0).repeat(B, 1, 1, 1).contiguous()
imgs += torch.randint(0, 3, imgs.shape, device="cuda:0", dtype=torch.uint8)
for _ in range(3): IT.image_transform(imgs, (OH, OW))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): y = IT.image_transform(imgs, (OH, OW))
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
e0.record()
for _ in range(20): y2 = IT.image_transform(imgs, None)
e1.record(); torch.cuda.synchronize()
ms_full = e0.elapsed_time(e1) / 20
pil = Image.fromarray(one, "RGB")
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 5.0:
    IR.to_tensor_normalize(np.asarray(pil.resize((OW, OH), Image.BILINEAR))); n += 1
cpu_s = (time.perf_counter() - t0) / n
alg = B * (H * W * 3 + OH * OW * 3 * 4)
alg_full = B * (H * W * 3 * 5)
print(json.dumps({"op": "Resize(224,224)+ToTensor+Normalize of 2500x250 RGB", "pictures": B,
                  "gpu_ms_per_batch": round(ms, 4), "gpu_pictures_per_s": round(B / ms * 1e3, 1),
                  "algorithmic_bytes": alg, "achieved_GBps": round(alg / ms / 1e6, 1), "hbm_peak_GBps": 8000,
                  "frac_of_hbm_peak": round(alg / ms / 1e6 / 8000, 4),
                  "no_resize_ms_per_batch": round(ms_full, 4), "no_resize_GBps": round(alg_full / ms_full / 1e6, 1),
                  "cpu_pictures_per_s_one_core": round(1 / cpu_s, 1), "cpu_ms_per_picture": round(cpu_s * 1e3, 3),
                  "bound": "hbm"}))
