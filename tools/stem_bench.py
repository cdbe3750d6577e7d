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
for name, fn in (("stem_fwd", fwd), ("stem_wgrad(+reduce)", wgrad)):
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
