"""Stream form of the 64 -> 64 channel halo conv tiles against the tile-at-a-time form, same process: outputs and statistics
rows must be bit-identical (same products, same accumulation order, same row grouping).  Several geometries: a batch whose tile
count is not a multiple of the workgroup count, one tile per workgroup, a single workgroup with many tiles (CU cap), bias + ReLU."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecgmm.hip import lib as L
from ecgmm.hip.functional import ptr, stream
lib = L.lib()
lib.ecgmm_conv_halo_enable(2)
bad = 0
for (B, H, W, cap, act, use_bias) in [(256, 56, 56, 0, 0, 0), (16, 56, 56, 0, 0, 0), (20, 56, 56, 7, 0, 0), (4, 32, 32, 1, 0, 0), (12, 56, 56, 5, 1, 1), (256, 56, 56, 100, 0, 0), (8, 8, 60, 3, 0, 0)]:
    d = L.ConvDesc(B, H, W, 64, 64, 3, 3, 1, 1, 1)
    g = torch.Generator(device="cuda:0").manual_seed(B * 131 + H)
    x = torch.randn(B * H * W * 64, device="cuda:0", generator=g).to(torch.bfloat16)
    w = (torch.randn(64 * 64 * 9, device="cuda:0", generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(64, device="cuda:0", generator=g) if use_bias else None
    add = torch.randn(B * H * W * 64, device="cuda:0", generator=g).to(torch.bfloat16)
    lib.ecgmm_conv_halo_cus(cap)
    res = {}
    for on in (0, 1):
        lib.ecgmm_conv_halo_stream(on)
        y = torch.full((B * H * W * 64,), 7.0, device="cuda:0").to(torch.bfloat16)
        dx = torch.full((B * H * W * 64,), 7.0, device="cuda:0").to(torch.bfloat16)
        st = torch.zeros(600 * 2 * 64, device="cuda:0")
        n = C.c_int(0)
        L.check(lib.ecgmm_conv_fwd_wgrows(L.BF16, C.byref(d), ptr(x), ptr(w), ptr(bias) if use_bias else None, ptr(y), ptr(st), C.byref(n), act, stream()))
        L.check(lib.ecgmm_conv_bwd_data(L.BF16, C.byref(d), ptr(x), ptr(w), None, ptr(dx), stream()))
        dxa = torch.full((B * H * W * 64,), 7.0, device="cuda:0").to(torch.bfloat16)
        L.check(lib.ecgmm_conv_bwd_data(L.BF16, C.byref(d), ptr(x), ptr(w), ptr(add), ptr(dxa), stream()))
        y2 = torch.full((B * H * W * 64,), 7.0, device="cuda:0").to(torch.bfloat16)
        L.check(lib.ecgmm_conv_fwd(L.BF16, C.byref(d), ptr(x), ptr(w), None, ptr(y2), None, act, stream()))
        torch.cuda.synchronize()
        res[on] = (y.clone(), st[: n.value * 2 * 64].clone(), dx.clone(), y2.clone(), n.value, dxa.clone())
    ok = all(torch.equal(res[0][i].view(torch.int16) if res[0][i].dtype == torch.bfloat16 else res[0][i], res[1][i].view(torch.int16) if res[1][i].dtype == torch.bfloat16 else res[1][i]) for i in (0, 1, 2, 3, 5)) and res[0][4] == res[1][4]
    for i, nm in ((0, "y+rows"), (1, "rows"), (2, "dx"), (3, "y(no stats)"), (5, "dx+addend")):
        a_, b_ = res[0][i].float(), res[1][i].float()
        if a_.shape == b_.shape and not torch.equal(a_, b_):
            df = (a_ != b_).nonzero().flatten()
            per = 256 * 64 if nm != "rows" else 128
            print(f"   {nm}: {df.numel()} of {a_.numel()} differ, first at {df[0].item()} (tile/row {df[0].item() // per}, offset {df[0].item() % per}), last tile {df[-1].item() // per}, max |d| {(a_ - b_).abs().max().item():.3e}, tiles hit {torch.unique(df // per).numel()}")
    fin = all(torch.isfinite(res[1][i].float()).all().item() for i in (0, 1, 2, 3, 5))
    print(f"B={B} {H}x{W} cap={cap} act={act} bias={use_bias}: rows {res[1][4]}  identical={ok} finite={fin}  |y|={res[1][0].float().abs().mean().item():.4f}")
    bad += (not ok) or (not fin)
lib.ecgmm_conv_halo_cus(0)
print("STREAM CHECK", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
