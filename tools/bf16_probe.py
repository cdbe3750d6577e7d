"""Diagnostic: gradient agreement of the bf16 trunk vs the exact-fp32 trunk (both HIP) and vs the CPU oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecgmm.image_encoder import ResNet18
from oracle import fill, ref_models as O

def run(shape, with_cpu, loss='sq'):
    ref = fill.hash_fill_module(O.ResNet18(num_classes=256), "r18.").train()
    x = fill.hash_tensor(shape, 607)
    nets = {}
    for cd in ("fp32", "bf16"):
        n = ResNet18(num_classes=256, compute_dtype=cd); n.load_state_dict(ref.state_dict()); n = n.cuda().train()
        f = n(x.cuda()); (f.square().mean() if loss == 'sq' else (f * fill.hash_tensor(tuple(f.shape), 99).cuda()).sum()).backward(); torch.cuda.synchronize()
        nets[cd] = (n, f.detach().cpu())
    if with_cpu:
        fr = ref(x); (fr.square().mean() if loss == 'sq' else (fr * fill.hash_tensor(tuple(fr.shape), 99)).sum()).backward()
    print("shape", shape, "feat rel err bf16 vs fp32:", ((nets['bf16'][1]-nets['fp32'][1]).norm()/nets['fp32'][1].norm()).item())
    g32 = dict(nets['fp32'][0].named_parameters()); g16 = dict(nets['bf16'][0].named_parameters())
    for k in g32:
        if 'conv' in k or 'fc.weight' in k or 'downsample.0' in k:
            a, b = g32[k].grad.flatten().double().cpu(), g16[k].grad.flatten().double().cpu()
            cos = (a @ b / (a.norm() * b.norm())).item()
            rel = ((a - b).norm() / a.norm()).item()
            extra = ""
            if with_cpu:
                c = dict(ref.named_parameters())[k].grad.flatten().double()
                extra = f" | fp32-vs-cpu rel {((a - c).norm() / c.norm()).item():.2e}"
            print(f"  {k:34s} cos {cos:.4f} rel {rel:.3f}{extra}")

run((4, 3, 64, 64), True, 'rnd')
run((32, 3, 224, 224), False, 'rnd')
