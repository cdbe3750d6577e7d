"""Stand-alone rates of the BatchNorm element-wise passes at the ResNet18 layer shapes (batch 256, bf16): bn_act without / with
a residual, full BatchNorm backward (reduce + finalize + apply) and the apply-from-rows form.  Bytes = the tensors each pass must
move once; GB/s against the 6.29 TB/s a float4 copy reaches on this chip (MI355X_MICROARCH.md)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecgmm.hip import lib as L
from ecgmm.hip.functional import ptr, stream
lib = L.lib()
B = 256
def timeit(fn, n=30):
    for _ in range(5): L.check(fn())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): L.check(fn())
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda:0")
for name, HW, Cn in (("l1", 56 * 56, 64), ("l2", 28 * 28, 128), ("l3", 14 * 14, 256), ("l4", 7 * 7, 512)):
    M = B * HW
    n = M * Cn
    mk = lambda: torch.randn(n, device="cuda:0").to(torch.bfloat16)
    # rotate over several buffers so that the passes do not run out of the 256 MB Infinity Cache
    K = max(2, int(600e6 // (n * 2)))
    ys, rs, os_ = [mk() for _ in range(K)], [mk() for _ in range(K)], [torch.empty(n, device="cuda:0", dtype=torch.bfloat16) for _ in range(K)]
    coef = torch.rand(4 * Cn, device="cuda:0") + 0.5
    gamma = torch.rand(Cn, device="cuda:0") + 0.5
    dg, db = torch.empty(Cn, device="cuda:0"), torch.empty(Cn, device="cuda:0")
    scratch = torch.empty(lib.ecgmm_bn_bwd_scratch(L.BF16, M, Cn), dtype=torch.uint8, device="cuda:0")
    it = [0]
    def nxt():
        it[0] = (it[0] + 1) % K
        return it[0]
    def act(): i = nxt(); return lib.ecgmm_bn_act(L.BF16, ptr(ys[i]), ptr(coef), None, None, None, 1, 1, ptr(os_[i]), M, Cn, stream())
    def act_res(): i = nxt(); return lib.ecgmm_bn_act(L.BF16, ptr(ys[i]), ptr(coef), ptr(rs[i]), None, None, 1, 1, ptr(os_[i]), M, Cn, stream())
    def bwd(): i = nxt(); return lib.ecgmm_bn_bwd(L.BF16, ptr(rs[i]), ptr(ys[i]), None, None, 1, ptr(ys[i]), ptr(coef), ptr(gamma), ptr(dg), ptr(db), ptr(os_[i]), None, None, M, Cn, ptr(scratch), stream())
    rows = torch.rand(256 * 2 * Cn, device="cuda:0")
    def bwd_rows(): i = nxt(); return lib.ecgmm_bn_bwd_from_rows(L.BF16, ptr(rs[i]), ptr(ys[i]), ptr(ys[i]), ptr(coef), ptr(gamma), ptr(dg), ptr(db), ptr(os_[i]), ptr(rows), 256, M, Cn, ptr(scratch), stream())
    for label, fn, tensors in (("bn_act", act, 2), ("bn_act+residual", act_res, 3), ("bn_bwd reduce+apply", bwd, 5), ("bn_bwd apply from rows", bwd_rows, 3)):
        us = timeit(fn)
        print(f"{name} C={Cn:3d} M={M:7d} {label:24s} {us:7.1f} us  {tensors * n * 2 / us / 1e6:6.2f} TB/s ({tensors} x {n * 2 / 1e6:.0f} MB, {K} buffer sets in rotation)")
