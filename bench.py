#!/usr/bin/env python3
"""Headline benchmark: samples/s of the full multimodal training step (forward + backward + Adam,
+ RCCL gradient all-reduce when N > 1) on synthetic ECG batches, bf16 trunks, per-GPU batch 256.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU; weak scaling: 256 samples per GPU)

Prints ONE JSON line on rank 0 with the driver's fields plus
  "roofline":     the dominant kernel class (MFMA implicit-GEMM conv) timed with HIP events on its
                  launch stream inside the timed region, algorithmic FLOPs / time vs the dense bf16
                  MFMA peak (2.5 PFLOP/s, /opt/skills/guides/MI355X_MICROARCH.md);
  "cpu_baseline": the CPU oracle (oracle/ref_models.py, a torch-fp32 restatement of the reference
                  step) timed on this box's host cores at the reference's CPU-runnable config
                  (batch 8), rank 0 and N == 1 only.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

FLOP_PER_SAMPLE = {  # SURVEY 8d / BASELINE.md section 4 (algorithmic, fwd + bwd)
    "multimodal": 11.7585e9,
    "image_only": 10.6453e9,
    "signal12": 1.1607e9,
}
HBM_ACHIEVABLE_TBS = 6.29     # float4 copy on MI355X (MI355X_MICROARCH.md, HBM)
BYTES_PER_SAMPLE_BF16 = {     # SURVEY 8d minimum-traffic model, bf16 activations (fp32: x2): conv inputs/outputs once each
    "image224": 26.1e6, "signal1": 8.34e6, "signal12": 8.34e6 + 2 * 11 * 5000 * 4.0,
}
ADAM_BYTES_PER_PARAM = 28.0


def resnet18_conv_macs(H, W):
    """(forward MACs per sample of the 20 convs, MACs of the stem alone) for an HxW picture."""
    def o(n, k, s, p):
        return (n + 2 * p - k) // s + 1
    h, w = o(H, 7, 2, 3), o(W, 7, 2, 3)
    stem = h * w * 64 * 147
    macs = stem
    h, w = o(h, 3, 2, 1), o(w, 3, 2, 1)
    cin = 64
    for L in range(4):
        cout = 64 << L
        for b in range(2):
            st = 2 if (b == 0 and L > 0) else 1
            h2, w2 = o(h, 3, st, 1), o(w, 3, st, 1)
            macs += h2 * w2 * cout * 9 * cin + h2 * w2 * cout * 9 * cout
            if st != 1 or cin != cout:
                macs += h2 * w2 * cout * cin
            h, w, cin = h2, w2, cout
    return macs, stem


def flop_per_sample(workload, H, W, frozen):
    """algorithmic FLOPs of one sample's step: fwd + dgrad + wgrad (no dgrad for the first conv of an encoder);
    with the encoders frozen (train.py:35-40) only their forward runs."""
    if workload == "signal12":
        return FLOP_PER_SAMPLE["signal12"]
    macs, stem = resnet18_conv_macs(H, W)
    img = 2.0 * macs if frozen else 2.0 * (3 * macs - stem)
    if workload == "image_only":
        return img + 6 * 1024.0
    sig_fwd_macs = 185_603_840
    sig = 2.0 * sig_fwd_macs if frozen else 1.1114e9
    return img + sig + 0.0018e9


PEAK_BF16_MFMA_TFLOPS = 2500.0
PEAK_F32_MFMA_TFLOPS = 157.3
PROF_KINDS = ["conv_igemm_fwd", "conv_igemm_dgrad", "conv_wgrad", "stem_fwd", "stem_wgrad", "conv_igemm_f32_fwd",
              "conv_igemm_f32_dgrad"]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)   # SURVEY 8d timing protocol: >= 20 warm-up, >= 100 timed steps
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--workload", default="multimodal", choices=list(FLOP_PER_SAMPLE))
    ap.add_argument("--image-hw", default="224x224", help="image size HxW (250x2500 = the reference's full-resolution "
                    "lead image, dataset_image.py:67-70)")
    ap.add_argument("--freeze-encoders", action="store_true", help="train.py:35-40 faithful step: the three encoders "
                    "get no gradients (forward in train mode, backward through the heads only); not the headline number")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="skip the HIP-event kernel timing")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--serialize", action="store_true", help="profiling aid: every kernel on ONE stream (no encoder / "
                    "weight-gradient stream overlap), so rocprofv3 durations are per-kernel-alone; not the headline number")
    return ap.parse_args()


def build(args, device):
    from ecgmm.config import Config
    from ecgmm.hip import functional as HF
    cfg = type("BenchConfig", (Config,), {})
    cfg.compute_dtype, cfg.clinical_input_dim, cfg.num_classes = args.dtype, 16, 2
    torch.manual_seed(42)
    HF.manual_seed(42)
    B = args.batch
    IH, IW = (int(v) for v in args.image_hw.lower().split("x"))
    g = torch.Generator(device="cpu").manual_seed(42 + int(os.environ.get("RANK", 0)))
    if args.workload == "multimodal":
        from ecgmm.multimodal_paper_modal_balance import ECGMultimodalModel
        model = ECGMultimodalModel(cfg)
        batch = (torch.randn(B, 3, IH, IW, generator=g).clamp_(-1, 1), torch.randn(B, 5000, generator=g),
                 torch.randn(B, 16, generator=g))

        def loss_fn(out, y):   # train.py:69-78
            return HF.cross_entropy_plus(out[3], y, out[4], 0.1)
    elif args.workload == "image_only":
        from ecgmm.train_image_only import ImageOnlyClassifier
        model = ImageOnlyClassifier(compute_dtype=args.dtype)
        batch = (torch.randn(B, 3, IH, IW, generator=g).clamp_(-1, 1),)

        def loss_fn(out, y):
            return HF.cross_entropy(out, y)
    else:
        from ecgmm.signal_model import ResNet1D_SE
        model = ResNet1D_SE(input_channels=12, num_classes=2, compute_dtype=args.dtype)
        batch = (torch.randn(B, 12, 5000, generator=g),)

        def loss_fn(out, y):
            return HF.focal_loss(out, y, 1.0, 2.0)
    labels = torch.randint(0, 2, (B,), generator=g)
    if args.freeze_encoders:
        if args.workload != "multimodal":
            raise SystemExit("--freeze-encoders applies to the multimodal workload (train.py:35-40)")
        for enc in (model.image_encoder, model.signal_encoder, model.clinical_encoder):
            for p in enc.parameters():
                p.requires_grad = False
    model = model.to(device).train()
    batch = tuple(t.to(device) for t in batch)
    return model, batch, labels.to(device), loss_fn


def usable_cores():
    """Host cores this process may actually use: min(affinity mask, cgroup CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    env = os.environ.get("ECGMM_CPU_CORES")
    if env:
        n = int(env)
    return max(1, min(n, 64))


def cpu_model_name():
    """CPU model string of this box (the CPU baseline states core count, CPU model and torch version)."""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine() or "unknown"


def latest_traffic_file():
    """profiles/rNN_igemm_traffic.json of the highest round present (written by tools/final_profiles.sh), or None."""
    import glob
    import re as _re
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_igemm_traffic.json")):
        m = _re.search(r"r(\d+)_igemm_traffic\.json$", f)
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), f)
    return best[1] if best else None


def cpu_baseline(seconds):
    """Reference CPU path (BASELINE config 1): oracle train step, batch 8, fp32, all host cores."""
    from oracle import ref_models as O
    cores = usable_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(42)
    model = O.ECGMultimodalModel(2, 16).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    B = 8
    img, sig, clin = torch.randn(B, 3, 224, 224).clamp_(-1, 1), torch.randn(B, 5000), torch.randn(B, 16)
    lab = torch.randint(0, 2, (B,))

    def step():
        opt.zero_grad()
        O.multimodal_loss(model(img, sig, clin), lab).backward()
        opt.step()
    step()
    t0 = time.perf_counter()
    n = 0
    while True:
        step()
        n += 1
        el = time.perf_counter() - t0
        if el > seconds or n >= 2000:
            break
    return {"value": round(B * n / el, 2), "unit": "samples/s", "cores": cores, "kind": "port",
            "cpu_model": cpu_model_name(), "torch": torch.__version__,
            "sample": f"{n} train steps of oracle.ECGMultimodalModel (torch {torch.__version__} CPU fp32), batch 8, "
                      f"3x224x224 + 5000-pt + 16-dim, CE + 0.1 var_loss, Adam"}


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start N ranks (one per GPU) with
    torch.distributed.run as a CHILD process -- before this process has touched the GPU, and never by exec --
    relay rank 0's JSON line and return the child's exit code."""
    import socket
    import subprocess
    ndev = torch.cuda.device_count()      # does not initialise the GPU
    if ndev < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {ndev} GPU(s) visible on this node")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    lines = [ln for ln in proc.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
    if proc.returncode != 0 or not lines:
        sys.stderr.write(f"bench.py: {args.gpus}-rank launch failed (exit {proc.returncode}, {len(lines)} result lines)\n")
        return proc.returncode or 1
    print(lines[-1], flush=True)
    return 0


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if env_world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={env_world} ranks; "
                         "pass --gpus equal to --nproc-per-node")
    # stdout carries exactly ONE line (the JSON result): libraries that print banners to fd 1 (RCCL prints its version
    # block there at communicator creation) are sent to stderr for the rest of the run
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP library is the only compute path (no CPU fallback)")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("nccl", device_id=device)   # RCCL over xGMI

    from ecgmm.hip import lib as L
    from ecgmm.optim import FusedAdam
    from ecgmm.parallel import DataParallel, flatten

    model, batch, labels, loss_fn = build(args, device)
    from ecgmm.parallel import reduction_order
    flatten(model, order=reduction_order(model))   # (same layout on one rank and on eight)
    # ECGMM_FORCE_DDP=1 rehearses the multi-rank code path (hooks, comm stream, bucketed all-reduce) on one rank
    force_ddp = os.environ.get("ECGMM_FORCE_DDP") == "1"
    if force_ddp and world == 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    # ECGMM_DDP_OVERLAP=0 (A/B): one all-reduce pass after the backward instead of per-stage-group launches on a comm stream
    ddp = (DataParallel(model, force=force_ddp, overlap=os.environ.get("ECGMM_DDP_OVERLAP", "1") != "0")
           if (world > 1 or force_ddp) else None)
    opt = FusedAdam((p for p in model.parameters() if p.requires_grad), lr=1e-4, grad_scale=1.0 / world)

    def step():
        opt.zero_grad()
        if ddp is not None:
            ddp.prepare_backward()
        loss = loss_fn(model(*batch), labels)
        loss.backward()
        if ddp is not None:
            ddp.reduce_gradients()
        opt.step()
        return loss

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    lib = L.lib()
    if args.serialize:
        lib.ecgmm_side_wgrad(0)
        if getattr(model, "config", None) is not None:
            model.config.overlap_encoders = False
    for _ in range(args.warmup):
        step()
    prof = not args.no_prof
    fence()
    if prof:
        # in the timed region only the kernel class the roofline line is about is bracketed by events (an event
        # pair costs ~1 us of stream time); the untimed one-stream pass below times every conv kernel kind
        L.check(lib.ecgmm_prof_enable(2 if args.dtype == "bf16" else 3), "prof_enable")
    fence()
    PROF_EVERY = 10  # the event pairs bracket every 10th step of the timed region (an event pair costs ~1-2 us of
    n_sampled = 0    # stream time, ~70 pairs per step, and a bracketed step runs ~4 % longer: 1 step in 10 keeps it under 0.5 % of `value`)
    # whole-step hipEvent timing: one event per step boundary on the compute stream (every side stream has been
    # joined to it by the step's last kernel, the fused Adam)
    step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    step_ev[0].record()
    for i in range(args.steps):
        if prof:
            lib.ecgmm_prof_pause(int(i % PROF_EVERY != 0))
            n_sampled += int(i % PROF_EVERY == 0)
        loss = step()
        step_ev[i + 1].record()
    fence()
    elapsed = time.perf_counter() - t0
    step_ms = sorted(step_ev[i].elapsed_time(step_ev[i + 1]) for i in range(args.steps))

    def pct(q):
        return round(step_ms[min(len(step_ms) - 1, int(q * (len(step_ms) - 1) + 0.5))], 3)
    peak = PEAK_BF16_MFMA_TFLOPS if args.dtype == "bf16" else PEAK_F32_MFMA_TFLOPS

    def collect_roofline(nsteps, note):
        nk = len(PROF_KINDS)
        ms, fl, by, cnt = (C.c_double * nk)(), (C.c_double * nk)(), (C.c_double * nk)(), (C.c_int64 * nk)()
        rc = lib.ecgmm_prof_collect(nk, ms, fl, by, cnt)
        kinds = {PROF_KINDS[i]: {"ms": ms[i], "flops": fl[i], "bytes": by[i], "launches": int(cnt[i])} for i in range(nk)}
        # the implicit-GEMM kernel template (fwd + dgrad instantiations) of the run's compute dtype is one kernel class;
        # in a bf16 run the exact-fp32 instantiation (dense tails, priced against another MFMA peak) is listed apart
        kf, kd = ("conv_igemm_fwd", "conv_igemm_dgrad") if args.dtype == "bf16" else ("conv_igemm_f32_fwd", "conv_igemm_f32_dgrad")
        ig_ms = kinds[kf]["ms"] + kinds[kd]["ms"]
        ig_fl = kinds[kf]["flops"] + kinds[kd]["flops"]
        ig_n = kinds[kf]["launches"] + kinds[kd]["launches"]
        ig_by = kinds[kf]["bytes"] + kinds[kd]["bytes"]
        if rc != 0 or ig_ms <= 0:
            return None
        # HBM traffic per launch from the rocprofv3 PMC passes of this same command (cannot be collected from
        # inside the process): tools/final_profiles.sh -> profiles/rNN_igemm_traffic.json (round-end code).  It was
        # measured on the headline configuration only -- unfrozen encoders, 224x224, batch 256, bf16 -- and is
        # attached to no other run (null there: absent, not zero).
        traffic = traffic_src = None
        tpath = latest_traffic_file()
        if (args.dtype == "bf16" and args.workload == "multimodal" and args.batch == 256 and tpath
                and not args.freeze_encoders and args.image_hw.lower() == "224x224"):
            traffic = round(json.load(open(tpath))["traffic_bytes_per_launch"], 1)
            traffic_src = os.path.relpath(tpath, ROOT)
        ach = ig_fl / (ig_ms * 1e-3) / 1e12
        return {"bound": "mfma", "kernel": ("igemm_kernel<bf16> + conv_halo_kernel (conv fwd + dgrad)" if args.dtype == "bf16" else "igemm_kernel<f32> (conv fwd + dgrad)"), "achieved": round(ach, 2), "peak": peak,
                "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": round(ig_by / max(ig_n, 1), 1), "launches": ig_n,
                "avg_launch_ms": round(ig_ms / max(ig_n, 1), 4), "flop_per_launch": round(ig_fl / max(ig_n, 1), 1),
                "by_kind": {k: {"ms_per_step": round(v["ms"] / nsteps, 3),
                                "tflops": round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 1),
                                "launches_per_step": v["launches"] // max(nsteps, 1)} for k, v in kinds.items()},
                "note": note}

    roof = roof_serial = None
    if prof:
        roof = collect_roofline(n_sampled, f"HIP events around each launch inside the timed region, on every {PROF_EVERY}th "
                                f"step ({n_sampled} of {args.steps} steps); kernels of the three "
                                "encoders and the weight-gradient kernels run CONCURRENTLY on separate streams, so a "
                                "launch's duration includes the time it shares the GPU with them")
        # second, untimed pass with the overlap switched off: each kernel alone on the GPU
        lib.ecgmm_side_wgrad(0)
        cfg_obj = getattr(model, "config", None)
        if cfg_obj is not None:
            cfg_obj.overlap_encoders = False
        step(); fence()
        L.check(lib.ecgmm_prof_enable(1), "prof_enable")
        n_serial = min(args.steps, 20)    # (~110 event pairs per step; csrc/prof.hip holds 8192)
        for _ in range(n_serial):
            step()
        fence()
        roof_serial = collect_roofline(n_serial, f"same launches, serialized on one stream (overlap off), untimed extra pass of {n_serial} steps")
        lib.ecgmm_prof_enable(0)
        lib.ecgmm_side_wgrad(1)
        if cfg_obj is not None:
            cfg_obj.overlap_encoders = True

    t = torch.tensor([elapsed], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = t.item()
    if rank == 0:
        total = args.batch * world * args.steps
        value = total / elapsed
        IH, IW = (int(v) for v in args.image_hw.lower().split("x"))
        flop_s = flop_per_sample(args.workload, IH, IW, args.freeze_encoders)
        esz = 1.0 if args.dtype == "bf16" else 2.0
        img_b = BYTES_PER_SAMPLE_BF16["image224"] * (IH * IW) / (224.0 * 224.0) * esz
        bytes_s = {"multimodal": img_b + BYTES_PER_SAMPLE_BF16["signal1"] * esz, "image_only": img_b,
                   "signal12": BYTES_PER_SAMPLE_BF16["signal12"] * esz}[args.workload]
        if args.freeze_encoders:
            bytes_s *= 4.666 / 13.046      # forward share of the minimum-traffic model
        adam_bytes = ADAM_BYTES_PER_PARAM * sum(p.numel() for p in model.parameters() if p.requires_grad)
        out = {
            "metric": "samples/sec fwd+bwd, batch-256 multimodal (img+sig+clin), 1->8 MI355X",
            "value": round(value, 1), "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": {"multimodal": "full multimodal (image 3x224x224 + 5000-pt signal + 16-dim clinical), "
                                                   "fwd+bwd+Adam, encoders unfrozen",
                                    "image_only": "image-only ResNet18 (train_image_only.py), fwd+bwd+Adam",
                                    "signal12": "12-lead ResNet1D_SE (train_signal_12_af.py), focal loss, fwd+bwd+Adam"}[args.workload],
                       "image_hw": args.image_hw, "per_gpu_batch": args.batch, "global_batch": args.batch * world,
                       "parallelism": f"dp{world}", "final_loss": round(float(loss.item()), 5)},
            "step_ms_hipevent": {"median": pct(0.5), "p10": pct(0.1), "p90": pct(0.9), "min": round(step_ms[0], 3),
                                 "max": round(step_ms[-1], 3), "n": len(step_ms)},
            # both whole-step fractions (the bf16 conv stack sits at the ridge): algorithmic FLOPs and
            # algorithmic (minimum-traffic) HBM bytes of one step over the measured step time
            "mfma_roofline_frac_whole_step": round(value / world * flop_s / (peak * 1e12), 4),
            "hbm_roofline_frac_whole_step": round((value / world * bytes_s + adam_bytes / (elapsed / args.steps)) /
                                                  (HBM_ACHIEVABLE_TBS * 1e12), 4),
            "flop_per_sample": flop_s, "algorithmic_bytes_per_sample": bytes_s,
            # `roofline`: the dominant kernel class, every launch ALONE on the GPU (extra untimed pass of this run with
            # the stream overlap off): its avg_launch_ms is what rocprofv3 --kernel-trace reports for
            # `bench.py --serialize`.  `roofline_timed_region`: the same launches bracketed inside the timed region,
            # where an event pair also spans the time a launch waits for CUs held by the other streams (lower bracket)
            "roofline": roof_serial if roof_serial is not None else roof,
            "roofline_timed_region": roof,
        }
        if args.freeze_encoders:
            out["config"]["workload"] += "; ENCODERS FROZEN as train.py:35-40 (forward + head backward only)"
            out["config"]["frozen_encoders"] = True
        if args.serialize:
            out["config"]["serialized_streams"] = True   # profiling configuration, not the headline number
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
