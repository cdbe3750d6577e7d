"""Run-to-run determinism of the full multimodal step (batch 256, bf16) under the current environment switches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecgmm.config import Config
from ecgmm.hip import functional as HF
from ecgmm.multimodal_paper_modal_balance import ECGMultimodalModel
B = 256
cfg = type("FullCfg", (Config,), {}); cfg.compute_dtype, cfg.clinical_input_dim = "bf16", 16
g = torch.Generator().manual_seed(7)
batch = (torch.randn(B, 3, 224, 224, generator=g).clamp_(-1, 1).cuda(), torch.randn(B, 5000, generator=g).cuda(),
         torch.randn(B, 16, generator=g).cuda(), torch.randint(0, 2, (B,), generator=g).cuda())
def run():
    torch.manual_seed(42); HF.manual_seed(42)
    m = ECGMultimodalModel(cfg).cuda().train()
    HF.manual_seed(123)
    out = m(*batch[:3])
    loss = HF.cross_entropy(out[3], batch[3]) + 0.1 * out[4]
    loss.backward(); torch.cuda.synchronize()
    return out[3].detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
runs = [run() for _ in range(3)]
bad = [k for k in runs[0][1] if not all(torch.equal(runs[0][1][k], r[1][k]) for r in runs[1:])]
print("forward equal:", all(torch.equal(runs[0][0], r[0]) for r in runs[1:]), " differing gradients:", len(bad), bad[:3], bad[-3:])
