"""Where a tile of the halo conv kernel spends its cycles: diagnostic build (make -C ecg-multimodal-model_amd/csrc stamp ->
libecgmm_hip_stamp.so, s_memtime stamps at a tile's phase boundaries; wave 0 of every workgroup) on the four 3x3 layer shapes,
forward (one statistics row per workgroup, the plans' form) and input gradient.  Shares, not lengths."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecgmm.hip import lib as L
L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), sys.argv[1] if len(sys.argv) > 1 else "libecgmm_hip_stamp.so")
from ecgmm.hip.functional import ptr, stream
lib = L.lib()
raw = C.CDLL(L.LIB_PATH)
raw.ecgmm_hstamp_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
SH = {"l1.3x3": (56, 56, 64, 64), "l2.3x3": (28, 28, 128, 128), "l3.3x3": (14, 14, 256, 256), "l4.3x3": (7, 7, 512, 512)}
B = 256
for name, (H, W, Cin, Cout) in SH.items():
    d = L.ConvDesc(B, H, W, Cin, Cout, 3, 3, 1, 1, 1)
    x = torch.randn(B * H * W * Cin, device="cuda:0").to(torch.bfloat16)
    w = (torch.randn(Cout * Cin * 9, device="cuda:0") * 0.05).to(torch.bfloat16)
    y = torch.empty(B * H * W * Cout, device="cuda:0", dtype=torch.bfloat16)
    rows = lib.ecgmm_conv_stats_rows(B * H * W)
    st = torch.empty((rows + 64) * 2 * Cout, device="cuda:0")
    n = C.c_int(0)
    runs = {"fwd": lambda: lib.ecgmm_conv_fwd_wgrows(L.BF16, C.byref(d), ptr(x), ptr(w), None, ptr(y), ptr(st), C.byref(n), 0, stream()),
            "dgrad": lambda: lib.ecgmm_conv_bwd_data(L.BF16, C.byref(d), ptr(x), ptr(w), None, ptr(y), stream()),
            "dgrad+addend": lambda: lib.ecgmm_conv_bwd_data(L.BF16, C.byref(d), ptr(x), ptr(w), ptr(x), ptr(y), stream())}
    for kind, fn in runs.items():
        out = (C.c_ulonglong * 8)()
        for _ in range(2):
            L.check(fn())
        torch.cuda.synchronize(); raw.ecgmm_hstamp_read(out, 1)
        for _ in range(5):
            L.check(fn())
        torch.cuda.synchronize(); raw.ecgmm_hstamp_read(out, 1)
        a, b, c, e, tiles, wgc, wgs, bar = [float(v) for v in out[:8]]
        tot = a + b + c + e
        print(f"{name:8s} {kind:13s} per tile: address table {a / tot:5.1%}  wait first fills {b / tot:5.1%} (barrier part {bar / tot:5.1%})  K loop {c / tot:5.1%}  "
              f"epilogue {e / tot:5.1%} | ticks/tile {tot / tiles:8.0f} | tiles/workgroup {tiles / wgs:5.2f} | tile phases = {tot / wgc:5.1%} of the workgroup's life")
