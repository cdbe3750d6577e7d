"""Data parallelism for the multimodal step: one process per GPU, per-patient batch sharded across
ranks, gradients summed by RCCL (``torch.distributed`` backend "nccl" on ROCm) over xGMI.

The reference has no distributed code at all; this is the one collective the MI355X build adds
(SURVEY 8e).  Design for xGMI (point-to-point links, per-link-bound rings): all gradients live in ONE
flat fp32 buffer (``flatten``), reduced in a few large buckets on a side stream while the remaining
backward stages still run (``DataParallel`` installs backward-stage hooks on the encoder plans), and
the 1/world scale is folded into the fused Adam (``grad_scale``) instead of a separate pass.
BatchNorm statistics stay per replica, as torch DDP does.
"""
import os
import torch
import torch.distributed as dist



def reduction_order(model):
    """Parameters in the order their gradients become final in the backward of the multimodal model: image encoder
    fc + layer4, then layer3, then everything else (layer2, layer1, stem, the other encoders, the head).  With the flat
    buffer laid out this way every stage group of DataParallel is ONE contiguous range, and what is left after the last
    overlapped group -- reduced after the backward, nothing left to hide it behind -- is one range too: one collective
    instead of two at the tail."""
    enc = getattr(model, "image_encoder", None)
    allp = list(model.parameters())
    if enc is None or not all(hasattr(enc, a) for a in ("fc", "layer4", "layer3")):
        return allp
    first = list(enc.fc.parameters()) + list(enc.layer4.parameters())
    second = list(enc.layer3.parameters())
    taken = {id(p) for p in first + second}
    return first + second + [p for p in allp if id(p) not in taken]


def flatten(model, only_trainable=True, order=None):
    """Re-point every (trainable) parameter and its gradient at slices of two flat fp32 buffers.
    Returns (flat_params, flat_grads).  Order = model.parameters() order, or `order` (a permutation of them)."""
    src = list(model.parameters()) if order is None else list(order)
    if order is not None and sorted(map(id, src)) != sorted(map(id, model.parameters())):
        raise ValueError("flatten: `order` must be a permutation of model.parameters()")
    params = [p for p in src if (p.requires_grad or not only_trainable)]
    if not params:
        return None, None
    dev = params[0].device
    total = sum(p.numel() for p in params)
    # pad each tensor to a multiple of 4 elements so every view stays 16-byte aligned
    offs, off = [], 0
    for p in params:
        offs.append(off)
        off += (p.numel() + 3) // 4 * 4
    flat_p = torch.zeros(off, device=dev, dtype=torch.float32)
    flat_g = torch.zeros(off, device=dev, dtype=torch.float32)
    with torch.no_grad():
        for p, o in zip(params, offs):
            n = p.numel()
            flat_p[o:o + n].copy_(p.data.reshape(-1))
            p.data = flat_p[o:o + n].view(p.shape)
            gv = flat_g[o:o + n].view(p.shape)
            p._ecg_grad_view = gv
            p.grad = gv
    model._ecg_flat = (flat_p, flat_g, params, offs)
    return flat_p, flat_g


class DataParallel(torch.nn.Module):
    """Minimal DDP: broadcast initial parameters from rank 0, then ``reduce_gradients()`` after (or,
    with ``overlap=True``, during) backward.  Gradients are SUMMED; pass ``grad_scale=1/world`` to
    FusedAdam (``self.grad_scale``)."""

    def __init__(self, module, process_group=None, bucket_mb=64, overlap=True, force=False):
        super().__init__()
        self.module = module
        self.pg = process_group
        self.world = dist.get_world_size(self.pg) if dist.is_initialized() else 1
        self.grad_scale = 1.0 / self.world
        if getattr(module, "_ecg_flat", None) is None:
            flatten(module, order=reduction_order(module))
        self.flat_p, self.flat_g, self._params, self._offs = module._ecg_flat
        self.bucket_elems = int(bucket_mb * 1024 * 1024 // 4)
        self._pending = []
        self._comm_stream = None
        import os
        self._tail_on_compute = os.environ.get("ECGMM_DDP_TAIL_ON_COMPUTE", "1") != "0"
        self._ar_from_side = os.environ.get("ECGMM_DDP_AR_FROM_SIDE", "1") != "0"
        self._works = []
        self.force = force and dist.is_initialized()
        self.overlap = overlap and (self.world > 1 or self.force) and self.flat_g.is_cuda
        if self.world > 1 or self.force:
            dist.broadcast(self.flat_p, src=0, group=self.pg)
            for b in module.buffers():
                dist.broadcast(b, src=0, group=self.pg)
        if self.overlap:
            self._comm_stream = torch.cuda.Stream()
            self._install_hooks()

    def forward(self, *a, **k):
        return self.module(*a, **k)

    # -- gradient ranges ------------------------------------------------------------------------
    def _range_of(self, params):
        idx = {id(p): i for i, p in enumerate(self._params)}
        ids = [idx[id(p)] for p in params if id(p) in idx]
        if not ids:
            return None
        lo, hi = min(ids), max(ids)
        return self._offs[lo], self._offs[hi] + (self._params[hi].numel() + 3) // 4 * 4

    def _install_hooks(self):
        """Split the image encoder's backward into stage groups; after each group its finished
        gradient range is all-reduced on the comm stream while earlier layers still compute."""
        enc = getattr(self.module, "image_encoder", None)
        if enc is None or not hasattr(enc, "_spec"):
            return
        import os
        if os.environ.get("ECGMM_DDP_GROUPS", "3") == "2":
            groups = [(0, 3), (3, 10)]               # fc+layer4 (70 % of the bytes) | everything else
            owners = [[enc.fc, enc.layer4], [enc.layer3, enc.layer2, enc.layer1, enc.conv1, enc.bn1]]
        else:
            groups = [(0, 3), (3, 5), (5, 10)]       # fc+layer4 | layer3 | layer2, layer1, stem
            owners = [[enc.fc, enc.layer4], [enc.layer3], [enc.layer2, enc.layer1, enc.conv1, enc.bn1]]
        ranges = []
        for mods in owners:
            ps = [p for m in mods for p in m.parameters() if p.requires_grad]
            ranges.append(self._range_of(ps))
        self._enc_ranges = ranges
        enc._spec.stage_groups = groups
        enc._spec.stage_hook = self._stage_hook

    def _stage_hook(self, spec, gi):
        r = self._enc_ranges[gi]
        if r is None:
            return
        if gi == len(self._enc_ranges) - 1 and self._tail_on_compute:
            # Nothing is left to overlap the LAST group's all-reduce with, and every stream hop costs ~80-100 us of exposed
            # latency on this platform (compute -> comm stream -> the process group's own stream -> comm -> compute = 4
            # hops, 0.32 ms in the one-rank rehearsal): the last group is reduced by reduce_gradients() on the compute
            # stream itself (2 hops), together with the parameters outside the image encoder.
            return
        self._launch(r[0], r[1], side=True)
        self._done_ranges.append(r)

    def _launch_from_side_stream(self, lo, hi):
        """Early (overlapped) all-reduce issued from the library's weight-gradient side stream: that stream is forked from
        the compute stream (so it also covers the gradients the compute stream wrote), the collective itself runs on the
        process group's own stream, and the handles are waited for in reduce_gradients().  No extra HIP stream."""
        from .hip import lib as L
        lib = L.lib()
        ptr = lib.ecgmm_side_stream()
        if not ptr:
            return False
        cur = torch.cuda.current_stream()
        L.check(lib.ecgmm_side_fork(cur.cuda_stream), "side_fork")
        ext = self._side_ext = getattr(self, "_side_ext", None) or torch.cuda.ExternalStream(ptr, device=cur.device)
        with torch.cuda.stream(ext):
            pos = lo
            while pos < hi:
                end = min(hi, pos + self.bucket_elems)
                self._works.append(dist.all_reduce(self.flat_g[pos:end], op=dist.ReduceOp.SUM, group=self.pg,
                                                   async_op=True))
                pos = end
        return True

    def _launch(self, lo, hi, side=False):
        if side and self._ar_from_side and self._launch_from_side_stream(lo, hi):
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self._comm_used = True
        with torch.cuda.stream(self._comm_stream):
            self._comm_stream.wait_event(ev)
            if side:
                # the stage's weight gradients were produced on the library's side stream, which that stage's
                # backward call did not join to the compute stream (hip/encoders.py): wait for it here instead
                from .hip import lib as L
                L.check(L.lib().ecgmm_side_wait(self._comm_stream.cuda_stream), "side_wait")
            pos = lo
            while pos < hi:
                end = min(hi, pos + self.bucket_elems)
                dist.all_reduce(self.flat_g[pos:end], op=dist.ReduceOp.SUM, group=self.pg)
                pos = end

    def _remaining(self, done):
        """Ranges of the flat buffer NOT covered by `done` (the ranges the stage hooks have shipped), in address order."""
        total = self.flat_g.numel()
        todo, pos = [], 0
        for lo, hi in sorted(done):
            if lo > pos:
                todo.append((pos, lo))
            pos = max(pos, hi)
        if pos < total:
            todo.append((pos, total))
        return todo

    def prepare_backward(self):
        self._done_ranges = []
        self._works = []

    def reduce_gradients(self):
        """All-reduce whatever the stage hooks have not already shipped, then make the compute
        stream wait for the comm stream."""
        if self.world == 1 and not self.force:
            return
        todo = self._remaining(getattr(self, "_done_ranges", []))
        if self.overlap:
            if self._tail_on_compute:
                # The backward has returned: autograd has joined the streams its nodes ran on, and the last stage group's
                # plan call joined the weight-gradient side stream.  The tail collective reads gradients that Functions
                # wrote straight into the flat buffer (they return None for parameters, so autograd's own leaf-stream
                # sync does not know about those writes): order it EXPLICITLY after the model's encoder side stream and
                # the library's weight-gradient stream -- two event waits that have normally fired already.
                cur = torch.cuda.current_stream()
                enc_side = getattr(self.module, "_side_stream", None)
                if enc_side is not None:
                    cur.wait_stream(enc_side)
                from .hip import lib as L
                L.check(L.lib().ecgmm_side_wait(cur.cuda_stream), "side_wait")
                for lo, hi in todo:
                    p = lo
                    while p < hi:
                        e = min(hi, p + self.bucket_elems)
                        dist.all_reduce(self.flat_g[p:e], op=dist.ReduceOp.SUM, group=self.pg)
                        p = e
            else:
                for lo, hi in todo:
                    self._launch(lo, hi)
            for w in getattr(self, "_works", []):
                w.wait()           # (orders the compute stream after the collective; no host block)
            self._works = []
            if getattr(self, "_comm_used", False):   # (an untouched stream stays off the hardware queues)
                torch.cuda.current_stream().wait_stream(self._comm_stream)
                self._comm_used = False
        else:
            for lo, hi in todo:
                p = lo
                while p < hi:
                    e = min(hi, p + self.bucket_elems)
                    dist.all_reduce(self.flat_g[p:e], op=dist.ReduceOp.SUM, group=self.pg)
                    p = e
        self._done_ranges = []
