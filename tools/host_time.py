"""How long does the host take to ENQUEUE one step (no sync) vs the GPU to execute it?"""
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
def step():
    opt.zero_grad(); loss = loss_fn(model(*batch), labels); loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"enqueue {1e3*(t1-t0):.2f} ms, total {1e3*(t2-t0):.2f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
