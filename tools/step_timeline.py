"""Main-stream timeline of the multimodal training step WITHOUT a profiler: timestamped events at the ResNet18 plan's phase
boundaries (ecgmm_tl_*) + host marks, averaged over steps, overlapped schedule vs everything on one stream.
Shows where the compute stream's time goes and how much each phase is stretched by the other streams' kernels."""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ecgmm.hip import lib as L
from ecgmm.hip.functional import stream
from ecgmm.optim import FusedAdam
from ecgmm.parallel import flatten, reduction_order
ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=20); ap.add_argument("--batch", type=int, default=256)
a = ap.parse_args()
args = argparse.Namespace(dtype="bf16", batch=a.batch, image_hw="224x224", workload="multimodal", freeze_encoders=False)
dev = torch.device("cuda:0")
model, batch, labels, loss_fn = bench.build(args, dev)
flatten(model, order=reduction_order(model))
opt = FusedAdam(model.parameters(), lr=1e-4)
lib = L.lib()
NAMES = {1: "step start", 2: "forward enqueued->done (head incl.)", 3: "loss", 4: "backward done", 5: "adam done",
         100: "img fwd start", 101: "stem+pool", 110: "avgpool+fc", 200: "img bwd start", 201: "fc+avgpool bwd", 210: "stem bwd", 211: "side joined"}
for i in range(8):
    NAMES[102 + i] = f"fwd block {i}"
    NAMES[202 + i] = f"bwd block {7 - i}"
def run(serial):
    lib.ecgmm_side_wgrad(0 if serial else 1)
    model.config.overlap_encoders = not serial
    def step(mark):
        m = (lambda i: lib.ecgmm_tl_mark(i, stream())) if mark else (lambda i: None)
        m(1); opt.zero_grad(); out = model(*batch); m(2); loss = loss_fn(out, labels); m(3); loss.backward(); m(4); opt.step(); m(5)
    for _ in range(5): step(False)
    torch.cuda.synchronize()
    acc, n = {}, 0
    for _ in range(a.steps):
        lib.ecgmm_tl_enable(1)
        step(True)
        torch.cuda.synchronize()
        ids, ms = (C.c_int * 256)(), (C.c_float * 256)()
        k = lib.ecgmm_tl_collect(256, ids, ms)
        lib.ecgmm_tl_enable(0)
        prev = 0.0
        for j in range(k):
            acc.setdefault((j, ids[j]), []).append(ms[j] - prev); prev = ms[j]
        n += 1
    return {key: sum(v) / len(v) for key, v in acc.items()}
ov, se = run(False), run(True)
print(f"{'mark':38s} {'overlapped ms':>14s} {'one stream ms':>14s}   (time since the previous mark on the compute stream)")
tot_o = tot_s = 0.0
for key in sorted(ov):
    o, s = ov[key], se.get(key, float('nan'))
    tot_o += o; tot_s += s
    print(f"{NAMES.get(key[1], str(key[1])):38s} {o:14.3f} {s:14.3f}")
print(f"{'total':38s} {tot_o:14.3f} {tot_s:14.3f}")
