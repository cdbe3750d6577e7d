"""Where an implicit-GEMM K-loop iteration spends its cycles: runs one conv shape on the diagnostic build
(make -C ecg-multimodal-model_amd/csrc stamp -> libecgmm_hip_stamp.so, s_memtime brackets) and prints the SHARES of
DMA issue / fragment reads + MFMA / barrier (+vmcnt drain), and K loop vs epilogue."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecgmm.hip import lib as L
L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), "libecgmm_hip_stamp.so")
from ecgmm.hip.functional import ptr, stream
lib = L.lib()
raw = C.CDLL(L.LIB_PATH)
raw.ecgmm_stamp_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
SH = {"l1.3x3": (56, 56, 64, 64), "l2.3x3": (28, 28, 128, 128), "l3.3x3": (14, 14, 256, 256), "l4.3x3": (7, 7, 512, 512)}
B = 256
for name, (H, W, Cin, Cout) in SH.items():
    d = L.ConvDesc(B, H, W, Cin, Cout, 3, 3, 1, 1, 1)
    x = torch.randn(B * H * W * Cin, device="cuda:0").to(torch.bfloat16)
    w = (torch.randn(Cout * Cin * 9, device="cuda:0") * 0.05).to(torch.bfloat16)
    y = torch.empty(B * H * W * Cout, device="cuda:0", dtype=torch.bfloat16)
    rows = lib.ecgmm_conv_stats_rows(B * H * W)
    st = torch.empty((rows + 64) * 2 * Cout, device="cuda:0")
    out = (C.c_ulonglong * 8)()
    for _ in range(2):
        L.check(lib.ecgmm_conv_fwd(L.BF16, C.byref(d), ptr(x), ptr(w), None, ptr(y), ptr(st), 0, stream()))
    torch.cuda.synchronize(); raw.ecgmm_stamp_read(out, 1)
    n = 5
    for _ in range(n):
        L.check(lib.ecgmm_conv_fwd(L.BF16, C.byref(d), ptr(x), ptr(w), None, ptr(y), ptr(st), 0, stream()))
    torch.cuda.synchronize(); raw.ecgmm_stamp_read(out, 1)
    a, b, c, e1, kl, ep, wg = [float(v) for v in out[:7]]
    tot = a + b + c
    print(f"{name}: per iteration: DMA issue {a / tot:5.1%}  frag reads + MFMA {b / tot:5.1%}  barrier (+vmcnt) {c / tot:5.1%}"
          f" | cycles/iteration {tot / wg / (9 * Cin // 64):7.0f} | K loop {kl / (kl + ep):5.1%} epilogue {ep / (kl + ep):5.1%}"
          f" (part 1 {e1 / (kl + ep):5.1%}; fast path's value/pack/swap/store loop {float(out[7]) / (kl + ep):5.1%}) | workgroups/launch {wg / n:.0f}")

if "--wgrad" in sys.argv:
    raw.ecgmm_wstamp_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    for name, (H, W, Cin, Cout) in SH.items():
        d = L.ConvDesc(B, H, W, Cin, Cout, 3, 3, 1, 1, 1)
        x = torch.randn(B * H * W * Cin, device="cuda:0").to(torch.bfloat16)
        dy = torch.randn(B * H * W * Cout, device="cuda:0").to(torch.bfloat16)
        dw = torch.empty(Cout * Cin * 9, device="cuda:0")
        nb = lib.ecgmm_conv_bwd_weight_workspace(L.BF16, C.byref(d))
        ws = torch.empty(nb, dtype=torch.uint8, device="cuda:0")
        out = (C.c_ulonglong * 8)()
        run = lambda: L.check(lib.ecgmm_conv_bwd_weight(L.BF16, C.byref(d), ptr(x), ptr(dy), ptr(dw), 0, ptr(ws), nb, stream()))
        run(); torch.cuda.synchronize(); raw.ecgmm_wstamp_read(out, 1)
        for _ in range(5): run()
        torch.cuda.synchronize(); raw.ecgmm_wstamp_read(out, 1)
        a, b, c, _, kl, _, wg, steps = [float(v) for v in out[:8]]
        tot = a + b + c
        print(f"wgrad {name}: per K step: DMA issue {a / tot:5.1%}  frag reads + MFMA {b / tot:5.1%}  barrier (+vmcnt) {c / tot:5.1%}"
              f" | cycles/step {tot / steps:7.0f} | loop share of kernel body {tot / kl:5.1%}")
