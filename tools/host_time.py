"""Host-side enqueue time of one training step (no device sync inside): is the step host-bound, and how late does the
host start the second encoder's backward?  Usage: python tools/host_time.py"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-prof"]
args = bench.parse()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
from ecgmm.optim import FusedAdam
from ecgmm.parallel import flatten
from ecgmm.hip import encoders as E
model, batch, labels, loss_fn = bench.build(args, dev)
flatten(model)
opt = FusedAdam((p for p in model.parameters() if p.requires_grad), lr=1e-4)
marks = []
orig_bwd = E._PlanFn.backward
def timed_bwd(ctx, dfeat):
    t = time.perf_counter(); r = orig_bwd(ctx, dfeat); marks.append((ctx.spec.name + ".bwd", t, time.perf_counter())); return r
E._PlanFn.backward = staticmethod(timed_bwd)
def step():
    marks.clear()
    t0 = time.perf_counter()
    opt.zero_grad()
    out = model(*batch)
    t1 = time.perf_counter()
    loss = loss_fn(out, labels)
    t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter()
    opt.step()
    t4 = time.perf_counter()
    return t0, t1, t2, t3, t4
for _ in range(5): step()
torch.cuda.synchronize()
for _ in range(3):
    torch.cuda.synchronize()
    t0, t1, t2, t3, t4 = step()
    torch.cuda.synchronize()
    t5 = time.perf_counter()
    print(f"host: fwd {1e3*(t1-t0):.2f} loss {1e3*(t2-t1):.2f} bwd {1e3*(t3-t2):.2f} adam {1e3*(t4-t3):.2f} | enqueue total {1e3*(t4-t0):.2f} ms, step done at {1e3*(t5-t0):.2f} ms")
    print("   ", " ".join(f"{n} {1e3*(a-t2):.2f}->{1e3*(b-t2):.2f}" for n, a, b in marks), "(ms after backward() was called)")
