"""A/B of the BatchNorm-backward reduction fused into the dgrad epilogue (conv_halo.hip, ConvEpi) against the separate
reduction pass, on the ResNet18 layer shapes (B = 256, bf16):  plain dgrad + full bn_bwd   vs   dgrad+reduction + tail."""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecgmm.hip import lib as L
from ecgmm.hip.functional import ptr, stream

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--reps", type=int, default=10)
a = ap.parse_args()
lib = L.lib()
dev = torch.device("cuda:0")
B, dt = a.batch, L.BF16
SH = [("l1", 56, 64), ("l2", 28, 128), ("l3", 14, 256), ("l4", 7, 512)]


def timeit(fn):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / a.reps


for name, HW, Cn in SH:
    M = B * HW * HW
    d = L.ConvDesc(B, HW, HW, Cn, Cn, 3, 3, 1, 1, 1)
    dy = torch.randn(M * Cn, device=dev).bfloat16()
    y = torch.randn(M * Cn, device=dev).bfloat16()
    out = torch.randn(M * Cn, device=dev).bfloat16()
    add = torch.randn(M * Cn, device=dev).bfloat16()
    w = (torch.randn(Cn * Cn * 9, device=dev) * 0.05).bfloat16()
    dx, dyo, dz = torch.empty_like(dy), torch.empty_like(dy), torch.empty_like(dy)
    coef = torch.randn(4, Cn, device=dev).abs() + 0.1
    gam = torch.ones(Cn, device=dev); dg = torch.empty(Cn, device=dev); db = torch.empty(Cn, device=dev)
    rows = torch.empty(256, 2, Cn, device=dev); n = C.c_int(0)
    scr = torch.empty(lib.ecgmm_bn_bwd_scratch(dt, M, Cn), device=dev, dtype=torch.uint8)
    t = {}
    t["dgrad"] = timeit(lambda: lib.ecgmm_conv_bwd_data(dt, C.byref(d), ptr(dy), ptr(w), None, ptr(dx), stream()))
    t["dgrad+add"] = timeit(lambda: lib.ecgmm_conv_bwd_data(dt, C.byref(d), ptr(dy), ptr(w), ptr(add), ptr(dx), stream()))
    t["dgrad+red1"] = timeit(lambda: lib.ecgmm_conv_bwd_data_bnred(dt, C.byref(d), ptr(dy), ptr(w), None, ptr(dx), ptr(y), ptr(y), ptr(coef), ptr(rows), C.byref(n), stream()))
    t["dgrad+add+red2"] = timeit(lambda: lib.ecgmm_conv_bwd_data_bnred(dt, C.byref(d), ptr(dy), ptr(w), ptr(add), ptr(dx), ptr(y), ptr(out), ptr(coef), ptr(rows), C.byref(n), stream()))
    t["bn1 full"] = timeit(lambda: lib.ecgmm_bn_bwd(dt, ptr(dx), ptr(y), None, None, 1, ptr(y), ptr(coef), ptr(gam), ptr(dg), ptr(db), ptr(dyo), None, None, M, Cn, ptr(scr), stream()))
    t["bn1 tail"] = timeit(lambda: lib.ecgmm_bn_bwd_from_rows(dt, ptr(dx), ptr(y), ptr(y), ptr(coef), ptr(gam), ptr(dg), ptr(db), ptr(dyo), ptr(rows), max(n.value, 1), M, Cn, ptr(scr), stream()))
    t["bn2 full"] = timeit(lambda: lib.ecgmm_bn_bwd(dt, ptr(dx), ptr(out), None, None, 1, ptr(y), ptr(coef), ptr(gam), ptr(dg), ptr(db), ptr(dyo), ptr(dz), None, M, Cn, ptr(scr), stream()))
    t["bn2 tail"] = timeit(lambda: lib.ecgmm_bn_bwd_from_rows(dt, ptr(dx), None, ptr(y), ptr(coef), ptr(gam), ptr(dg), ptr(db), ptr(dyo), ptr(rows), max(n.value, 1), M, Cn, ptr(scr), stream()))
    print(f"{name} rows {n.value:3d} | " + " | ".join(f"{k} {v:6.1f}" for k, v in t.items()), flush=True)
    print(f"     bn1: separate {t['dgrad'] + t['bn1 full']:6.1f} us, fused {t['dgrad+red1'] + t['bn1 tail']:6.1f} us;   "
          f"bn2: separate {t['dgrad+add'] + t['bn2 full']:6.1f} us, fused {t['dgrad+add+red2'] + t['bn2 tail']:6.1f} us", flush=True)
