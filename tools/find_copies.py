"""Which host code paths issue device-to-device copies in one training step? (torch.profiler, aten::copy_ with stacks)"""
import os, sys
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
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    step()
torch.cuda.synchronize()
import collections
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::to", "aten::_to_copy"):
        st = [s for s in (e.stack or []) if "ecg-multimodal" in s or "bench.py" in s or "ecgmm" in s]
        cnt[(e.name, str(e.input_shapes)[:60], st[0][-70:] if st else "?")] += 1
for k, v in cnt.most_common(40):
    print(v, k)
