"""Times the 2-D stem kernels alone (B x 3 x 224 x 224, bf16) with events on the launch stream.  --lib: another build."""
import argparse, os, sys
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--lib", default="")
a = ap.parse_args()
if a.lib:
    os.environ["ECGMM_LIB"] = a.lib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecgmm.hip import lib as L
from ecgmm.hip.functional import ptr, stream
lib = L.lib()
dev = "cuda"
N, Cin, H, W, R = a.batch, 3, 224, 224, 7
x = torch.randn(N, Cin, H, W, device=dev).clamp_(-1, 1)
w = torch.randn(64, Cin, R, 7, device=dev) * 0.05
pk = torch.empty(lib.ecgmm_stem_packed_elems(Cin, R), device=dev, dtype=torch.bfloat16)
L.check(lib.ecgmm_stem_pack(1, ptr(w), ptr(pk), Cin, R, stream()))
OH = OW = 112
y = torch.empty(N, OH, OW, 64, device=dev, dtype=torch.bfloat16)
rows = lib.ecgmm_stem_stats_rows(N, Cin, H, W, R)
stats = torch.empty(rows, 2, 64, device=dev)
dy = torch.randn(N, OH, OW, 64, device=dev).to(torch.bfloat16)
dw = torch.empty(64, Cin, R, 7, device=dev)
nb = lib.ecgmm_stem_bwd_weight_workspace(N, Cin, H, W, R)
ws = torch.empty(nb, device=dev, dtype=torch.uint8)

def fwd():
    L.check(lib.ecgmm_stem_fwd(1, ptr(x), ptr(pk), None, ptr(y), ptr(stats), N, Cin, H, W, R, stream()))
def wgrad():
    L.check(lib.ecgmm_stem_bwd_weight(1, ptr(x), ptr(dy), ptr(dw), 0, ptr(ws), nb, N, Cin, H, W, R, stream()))
# ---- the stem by recompute (conv_stem_fused.hip) and the passes of the two-pass route it replaces
PH = PW = 56
coef = torch.empty(4, 64, device=dev)
gam, bet = torch.ones(64, device=dev), torch.zeros(64, device=dev)
rm, rv, nbt = torch.zeros(64, device=dev), torch.ones(64, device=dev), torch.zeros((), dtype=torch.int64, device=dev)
fwd()
L.check(lib.ecgmm_bn_finalize(ptr(stats), rows, 64, float(N * OH * OW), ptr(gam), ptr(bet), ptr(rm), ptr(rv), ptr(nbt), 0.1, 1e-5, ptr(coef), stream()))
pooled = torch.empty(N, PH, PW, 64, device=dev, dtype=torch.bfloat16)
idx = torch.empty(N, PH, PW, 64, device=dev, dtype=torch.uint8)
dp = torch.randn(N, PH, PW, 64, device=dev).to(torch.bfloat16)
rows2 = lib.ecgmm_stem_stats_only_rows(N, Cin, H, W, R)
stats2 = torch.empty(rows2 + 64, 2, 64, device=dev)
nb2 = lib.ecgmm_stem_pool_bwd_workspace(N, Cin, H, W)
ws2 = torch.empty(nb2, device=dev, dtype=torch.uint8)
scratch = torch.empty(lib.ecgmm_bn_bwd_scratch(1, N * OH * OW, 64), device=dev, dtype=torch.uint8)
dgam, dbet = torch.empty(64, device=dev), torch.empty(64, device=dev)
dy2 = torch.empty_like(dy)
rows3 = lib.ecgmm_stem_wg_stats_rows(N, Cin, H, W, R)
stats3 = torch.empty(rows3 + 64, 2, 64, device=dev)
def fwd_wg():
    L.check(lib.ecgmm_stem_fwd_wgrows(1, ptr(x), ptr(pk), None, ptr(y), ptr(stats3), N, Cin, H, W, R, stream()))
def stats_only():
    L.check(lib.ecgmm_stem_stats_only(1, ptr(x), ptr(pk), None, ptr(stats2), N, Cin, H, W, R, stream()))
def pool_fwd():
    L.check(lib.ecgmm_stem_pool_fwd(ptr(x), ptr(pk), ptr(coef), ptr(pooled), ptr(idx), N, Cin, H, W, stream()))
def maxpool():
    L.check(lib.ecgmm_bnrelu_maxpool(1, ptr(y), ptr(coef), ptr(pooled), ptr(idx), N, OH, OW, 64, stream()))
def pool_bn_bwd():
    L.check(lib.ecgmm_pool_bn_bwd(1, ptr(dp), ptr(pooled), ptr(idx), ptr(y), ptr(coef), ptr(gam), ptr(dgam), ptr(dbet), ptr(dy2), None, N, OH, OW, 64, ptr(scratch), stream()))
def pool_bwd_fused():
    L.check(lib.ecgmm_stem_pool_bwd(ptr(x), ptr(pk), ptr(coef), ptr(gam), ptr(dp), ptr(pooled), ptr(idx), ptr(dgam), ptr(dbet), ptr(dw), ptr(ws2), nb2, N, Cin, H, W, stream()))
for name, fn in (("stem_fwd", fwd), ("stem_fwd (per-workgroup statistics rows)", fwd_wg), ("bnrelu_maxpool", maxpool), ("stem_stats_only", stats_only), ("stem_pool_fwd", pool_fwd),
                 ("pool_bn_bwd (reduce+finalize+apply)", pool_bn_bwd), ("stem_wgrad(+reduce)", wgrad),
                 ("stem_pool_bwd (reduce+finalize+fused wgrad+reduce)", pool_bwd_fused)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{os.path.basename(a.lib) or 'default'} {name}: {e0.elapsed_time(e1) / a.reps * 1e3:.1f} us")
