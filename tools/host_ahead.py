"""Does the host run ahead of the GPU in steady state?  Host time per step for 24 back-to-back steps (no sync)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.argv = ["bench.py"]
import bench
args = bench.parse()
dev = torch.device("cuda:0")
model, batch, labels, loss_fn = bench.build(args, dev)
from ecgmm.optim import FusedAdam
from ecgmm.parallel import flatten
flatten(model)
opt = FusedAdam(model.parameters(), lr=1e-4)
marks = []
def step():
    t0 = time.perf_counter(); opt.zero_grad(); out = model(*batch); t1 = time.perf_counter()
    loss = loss_fn(out, labels); t2 = time.perf_counter(); loss.backward(); t3 = time.perf_counter(); opt.step(); t4 = time.perf_counter()
    marks.append((t0, t1, t2, t3, t4))
for _ in range(5): step()
torch.cuda.synchronize(); marks.clear()
T0 = time.perf_counter()
for _ in range(24): step()
T1 = time.perf_counter(); torch.cuda.synchronize(); T2 = time.perf_counter()
print(f"host loop {1e3*(T1-T0):.1f} ms, with drain {1e3*(T2-T0):.1f} ms for 24 steps")
for i, (t0, t1, t2, t3, t4) in enumerate(marks):
    print(f"step {i:2d} start {1e3*(t0-T0):7.2f}  fwd {1e3*(t1-t0):5.2f}  loss {1e3*(t2-t1):5.2f}  bwd {1e3*(t3-t2):5.2f}  opt {1e3*(t4-t3):5.2f}  total {1e3*(t4-t0):5.2f}")
