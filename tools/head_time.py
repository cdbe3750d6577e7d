"""GPU time of the fusion head alone (forward + CE/var loss + backward) at batch 256: the part of the step during which
nothing else can run.  python tools/head_time.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ecgmm.config import Config
from ecgmm.multimodal_paper_modal_balance import ECGMultimodalModel
from ecgmm.hip import functional as HF
from ecgmm.hip import encoders as E
dev = torch.device("cuda:0")
cfg = Config(); cfg.clinical_input_dim = 16; cfg.synthetic = True
m = ECGMultimodalModel(cfg).to(dev).train()
B = 256
f = [torch.randn(B, d, device=dev, requires_grad=True) for d in (m.image_dim, m.signal_dim, m.clinical_dim)]
y = torch.randint(0, 2, (B,), device=dev)
def it():
    out = E.run_head(f[0], f[1], f[2], m._head_spec(), m._head_params())
    loss = HF.cross_entropy_plus(out[3], y, out[4], 0.1)
    loss.backward()
    HF.release_grads(params)
params = list(m.parameters())
for _ in range(5): it()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
ev[0].record()
for i in range(40):
    it(); ev[i + 1].record()
torch.cuda.synchronize()
t = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(40))
import time
t0 = time.perf_counter()
for _ in range(40): it()
host = (time.perf_counter() - t0) / 40 * 1e3
torch.cuda.synchronize()
print(f"head fwd+loss+bwd: median {t[20]*1e3:.0f} us, min {t[0]*1e3:.0f} us per iteration (host enqueue {host*1e3:.0f} us)")
