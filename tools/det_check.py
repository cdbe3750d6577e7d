"""Run-to-run determinism of the image encoder's step at batch 256 (bf16) under the current environment switches:
two forward+backward passes from identical state; prints which gradients differ.  tools/det_check.py [--batch 256]"""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecgmm.image_encoder import resnet18
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=256); a = ap.parse_args()
torch.manual_seed(3)
net = resnet18(num_classes=8, compute_dtype="bf16").to("cuda").train()
g = torch.Generator().manual_seed(11)
x = torch.randn(a.batch, 3, 224, 224, generator=g).clamp_(-1, 1).cuda()
r = torch.randn(a.batch, 8, generator=g).cuda()
def run():
    for p in net.parameters():
        p.grad = None
    y = net(x)
    (y * r).sum().backward()
    torch.cuda.synchronize()
    return y.detach().clone(), {k: p.grad.detach().clone() for k, p in net.named_parameters()}
y0, g0 = run(); y1, g1 = run(); y2, g2 = run()
bad = [k for k in g0 if not (torch.equal(g0[k], g1[k]) and torch.equal(g0[k], g2[k]))]
print("forward equal:", torch.equal(y0, y1) and torch.equal(y0, y2), " differing gradients:", len(bad), bad[-6:])
