"""Micro-benchmark of the conv kernels on the ResNet18 / ResNet1D_SE layer shapes (B=256 by default).
Usage: python tools/conv_bench.py [--batch 256] [--reps 10] [--only fwd,dgrad,wgrad] [--layers l1,l2,...]"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecgmm.hip import lib as L
from ecgmm.hip.functional import ptr, stream

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--only", default="fwd,dgrad,wgrad")
ap.add_argument("--layers", default="")
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--no-stats", action="store_true", help="forward without the fused BatchNorm partial sums")
ap.add_argument("--lib", default="", help="A/B: path of another build of libecgmm_hip.so")
ap.add_argument("--ring", type=int, default=-1, help="wgrad_ring_kernel: 0 off, 1 on (default)")
ap.add_argument("--halo", type=int, default=-1, help="conv_halo.hip: 0 never, 1 where faster (default), 2 wherever applicable")
ap.add_argument("--w4", type=int, default=-1, help="64->64 3x3 layers on 4-wave workgroups, two per CU: 0 off, 1 on (default)")
ap.add_argument("--wgrows", action="store_true", help="forward statistics as one row per workgroup (the plans' form)")
ap.add_argument("--pp", type=int, default=-1, help="halo kernel, 128-channel tiles: ping-pong K loop 0 off, 1 on (default)")
ap.add_argument("--wpp", type=int, default=-1, help="wgrad ring kernel: ping-pong between its two wave groups 0 off, 1 on (default)")
ap.add_argument("--stagger", type=int, default=-1, help="halo kernel: waves 4-7 one MFMA block late: 0 off, 1 on (default)")
ap.add_argument("--stream", type=int, default=-1, help="halo kernel, 64 -> 64 channel 3x3 tiles: stream form: 0 off, 1 on (default)")
a = ap.parse_args()
if a.lib:
    L.LIB_PATH = a.lib
B, dt = a.batch, (L.BF16 if a.dtype == "bf16" else L.F32)
tdt = torch.bfloat16 if dt == L.BF16 else torch.float32
SHAPES = [  # name, H, W, Cin, Cout, R, S, stride, ph, pw
    ("l1.3x3", 56, 56, 64, 64, 3, 3, 1, 1, 1), ("l2.3x3s2", 56, 56, 64, 128, 3, 3, 2, 1, 1),
    ("l2.3x3", 28, 28, 128, 128, 3, 3, 1, 1, 1), ("l2.ds", 56, 56, 64, 128, 1, 1, 2, 0, 0),
    ("l3.3x3s2", 28, 28, 128, 256, 3, 3, 2, 1, 1), ("l3.3x3", 14, 14, 256, 256, 3, 3, 1, 1, 1),
    ("l4.3x3s2", 14, 14, 256, 512, 3, 3, 2, 1, 1), ("l4.3x3", 7, 7, 512, 512, 3, 3, 1, 1, 1),
    ("s1.k3", 1, 1250, 64, 64, 1, 3, 1, 0, 1), ("s2.k3s2", 1, 1250, 64, 128, 1, 3, 2, 0, 1),
    ("s2.k3", 1, 625, 128, 128, 1, 3, 1, 0, 1), ("s3.k3", 1, 313, 256, 256, 1, 3, 1, 0, 1),
]
lib = L.lib()
if a.halo >= 0:
    lib.ecgmm_conv_halo_enable(a.halo)
if a.pp >= 0:
    lib.ecgmm_conv_halo_pingpong(a.pp)
if a.wpp >= 0:
    lib.ecgmm_conv_wgrad_pingpong(a.wpp)
if a.stream >= 0:
    lib.ecgmm_conv_halo_stream(a.stream)
if a.stagger >= 0:
    lib.ecgmm_conv_halo_stagger(a.stagger)
if a.w4 >= 0:
    lib.ecgmm_conv_halo_w4(a.w4)
if a.ring >= 0:
    lib.ecgmm_conv_wgrad_ring_enable(a.ring)
dev = torch.device("cuda:0")
sel = set(a.layers.split(",")) if a.layers else None
tot = {}
for name, H, W, Cin, Cout, R, S, st, ph, pw in SHAPES:
    if sel and name not in sel:
        continue
    d = L.ConvDesc(B, H, W, Cin, Cout, R, S, st, ph, pw)
    OH, OW = (H + 2 * ph - R) // st + 1, (W + 2 * pw - S) // st + 1
    x = torch.randn(B * H * W * Cin, device=dev).to(tdt)
    dy = torch.randn(B * OH * OW * Cout, device=dev).to(tdt)
    w = (torch.randn(Cout * Cin * R * S, device=dev) * 0.05).to(tdt)
    y = torch.empty_like(dy); dx = torch.empty_like(x)
    dw = torch.empty(Cout * Cin * R * S, device=dev)
    rows = lib.ecgmm_conv_stats_rows(B * OH * OW)
    stats = torch.empty((rows + 64) * 2 * Cout, device=dev)
    nb = lib.ecgmm_conv_bwd_weight_workspace(dt, C.byref(d))
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    flops = 2.0 * B * OH * OW * Cout * R * S * Cin
    nrows = C.c_int(0)
    ybn = torch.randn(B * H * W * Cin, device=dev).to(tdt)          # (fused BatchNorm-backward reduction: y of the consumer BN / a residual addend)
    coefbn = torch.rand(4 * Cin, device=dev) + 0.5
    rowsbn = torch.empty(512 * 2 * Cin, device=dev)
    runs = {
        "fwd": (lambda: lib.ecgmm_conv_fwd_wgrows(dt, C.byref(d), ptr(x), ptr(w), None, ptr(y), ptr(stats), C.byref(nrows), 0, stream())) if a.wgrows else lambda: lib.ecgmm_conv_fwd(dt, C.byref(d), ptr(x), ptr(w), None, ptr(y), None if a.no_stats else ptr(stats), 0, stream()),
        "dgrad": lambda: lib.ecgmm_conv_bwd_data(dt, C.byref(d), ptr(dy), ptr(w), None, ptr(dx), stream()),
        "dgradred": lambda: lib.ecgmm_conv_bwd_data_bnred(dt, C.byref(d), ptr(dy), ptr(w), None, ptr(dx), ptr(ybn), ptr(ybn), ptr(coefbn), ptr(rowsbn), C.byref(nrows), stream()),
        "dgradadd": lambda: lib.ecgmm_conv_bwd_data(dt, C.byref(d), ptr(dy), ptr(w), ptr(ybn), ptr(dx), stream()),
        "wgrad": lambda: lib.ecgmm_conv_bwd_weight(dt, C.byref(d), ptr(x), ptr(dy), ptr(dw), 0, ptr(ws), nb, stream()),
    }
    line = f"{name:10s}"
    for k in a.only.split(","):
        for _ in range(2):
            L.check(runs[k]())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            runs[k]()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.reps
        tot[k] = tot.get(k, 0) + us
        line += f"  {k} {us:7.1f} us {flops / us / 1e6:6.0f} TF"
    print(line, flush=True)
print("sum us:", {k: round(v, 1) for k, v in tot.items()})
