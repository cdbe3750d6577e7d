"""Fused Adam on the HIP kernel -- same constructor, param_groups and update rule as the
``torch.optim.Adam`` the reference uses (train.py:43,81; train_image_only.py:111,132;
train_signal_12_af.py:249-252 under OneCycleLR, which rewrites ``lr`` and ``betas`` every step).

Parameters/gradients that sit back to back in memory (after ``parallel.flatten``) are updated by ONE
kernel launch per contiguous run; otherwise one launch per tensor.
"""
import torch

from .hip import functional as HF
from .hip import lib as L


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.grad_scale = grad_scale      # e.g. 1/world_size after a summed all-reduce
        self._runs = None

    def zero_grad(self, set_to_none=False):
        """Backward OVERWRITES gradients (hip/functional.py), so nothing is cleared here: the buffers are kept
        (they stay views of the flat all-reduce buffer) and marked consumed, so the next backward may write
        them.  A parameter that the next backward does NOT write is zeroed by ``step()`` (torch's
        ``zero_grad(set_to_none=False)`` semantics), never updated from a stale gradient."""
        for g in self.param_groups:
            for p in g["params"]:
                p._ecg_dirty = False
                if set_to_none:
                    p.grad = None

    def _build_runs(self):
        runs = []
        for gi, group in enumerate(self.param_groups):
            cur = None
            # (address order, not registration order: the step is elementwise, and a flat buffer laid out in another
            # order -- parallel.reduction_order -- still becomes one launch)
            for p in sorted(group["params"], key=lambda t: t.data_ptr()):
                if not p.requires_grad or p.grad is None:
                    continue   # no gradient yet: skipped like torch.optim.Adam does (never created here)
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("FusedAdam: parameters must be contiguous fp32")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                g = p.grad
                if g.dtype != torch.float32 or not g.is_contiguous():
                    raise RuntimeError("FusedAdam: gradients must be contiguous fp32")
                n = p.numel()
                gap = p.data_ptr() - cur["p_end"] if cur is not None else -1
                # flatten() pads tensors to 16 B: a gap of <= 12 B inside the flat buffers is padding whose
                # gradient stays zero, so it can ride along in the same launch
                # (one step counter per run = per launch: a parameter whose own count differs -- it gained its gradient
                # later than its neighbours -- starts a new run, so its bias correction is torch's per-parameter one)
                if (cur is not None and 0 <= gap <= 12 and gap % 4 == 0 and g.data_ptr() - cur["g_end"] == gap
                        and st["step"] == cur["step"]):
                    cur["offs"].append(cur["n"] + gap // 4)
                    cur["n"] += gap // 4 + n
                    cur["p_end"] = p.data_ptr() + n * 4
                    cur["g_end"] = g.data_ptr() + n * 4
                    cur["params"].append(p)
                else:
                    cur = dict(group=gi, p=p.data_ptr(), g=g.data_ptr(), n=n, p_end=p.data_ptr() + n * 4,
                               g_end=g.data_ptr() + n * 4, params=[p], offs=[0], step=st["step"])
                    runs.append(cur)
        for r in runs:
            dev = r["params"][0].device
            r["m"] = torch.zeros(r["n"], device=dev, dtype=torch.float32)
            r["v"] = torch.zeros(r["n"], device=dev, dtype=torch.float32)
        self._runs = runs

    def _runs_valid(self):
        if self._runs is None:
            return False
        n_with_grad = sum(1 for g in self.param_groups for p in g["params"] if p.requires_grad and p.grad is not None)
        if n_with_grad != sum(len(r["params"]) for r in self._runs):
            return False   # a parameter gained (or lost) its gradient since the runs were planned
        for r in self._runs:
            for p, off in zip(r["params"], r["offs"]):
                if p.grad is None or p.data_ptr() != r["p"] + off * 4 or p.grad.data_ptr() != r["g"] + off * 4:
                    return False
        return True

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if not self._runs_valid():
            old = self._runs
            self._build_runs()
            if old is not None:  # keep the moments when the memory layout changed (e.g. flatten() after a step)
                prev = {}
                for r in old:
                    for p, off in zip(r["params"], r["offs"]):
                        n = p.numel()
                        prev[id(p)] = (r["m"][off:off + n], r["v"][off:off + n])
                for r in self._runs:
                    for p, off in zip(r["params"], r["offs"]):
                        n = p.numel()
                        if id(p) in prev:
                            r["m"][off:off + n].copy_(prev[id(p)][0])
                            r["v"][off:off + n].copy_(prev[id(p)][1])
        # a gradient buffer that some earlier backward wrote but the one since the last zero_grad()/step() did
        # not holds a stale value: zero it (what torch's zero_grad(set_to_none=False) would have left there)
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is not None and not getattr(p, "_ecg_dirty", True) and getattr(p, "_ecg_written", False):
                    p.grad.zero_()
                    p._ecg_written = False
        lib = L.lib()
        s = HF.stream()
        for r in self._runs:
            g = self.param_groups[r["group"]]
            r["step"] += 1
            b1, b2 = g["betas"]
            L.check(lib.ecgmm_adam(r["p"], r["g"], HF.ptr(r["m"]), HF.ptr(r["v"]), r["n"], float(g["lr"]), float(b1),
                                   float(b2), float(g["eps"]), float(g["weight_decay"]), r["step"],
                                   float(self.grad_scale), s), "adam")
            for p in r["params"]:
                self.state[p]["step"] = r["step"]
        for g in self.param_groups:
            for p in g["params"]:
                p._ecg_dirty = False    # consumed: the next backward may overwrite
        return loss


Adam = FusedAdam
