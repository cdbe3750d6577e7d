"""The data-parallel step on the GPU (RCCL, one rank): the stage-hooked backward (gradient ranges all-reduced on the
comm stream while earlier stages still compute, weight-gradient side stream joined only at the last stage group)
produces exactly the gradients of the plain backward.  Multi-rank averaging itself is covered on CPU with gloo
(tests/test_host_cpu.py); the driver runs the real N = 2/4/8 case."""
import os

import pytest
import torch
import torch.distributed as dist

from ecgmm.config import Config
from ecgmm.hip import functional as HF
from ecgmm.multimodal_paper_modal_balance import ECGMultimodalModel
from ecgmm.parallel import DataParallel, flatten, reduction_order
from oracle import fill

from .util import DEV, dev

pytestmark = pytest.mark.gpu


def _model(cd):
    cfg = type("C", (Config,), {})
    cfg.compute_dtype, cfg.clinical_input_dim, cfg.num_classes = cd, 16, 2
    m = fill.hash_fill_module(ECGMultimodalModel(cfg), "mm.")
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    return m.to(DEV).train()


@pytest.mark.parametrize("cd", ["bf16", "fp32"])
def test_stage_hooked_backward_equals_plain_backward(cd):
    img, sig, clin, lab = (dev(t) for t in fill.synthetic_batch(8, salt=11))
    plain = _model(cd)
    _, g_plain = flatten(plain, order=reduction_order(plain))   # bench.py's layout
    out = plain(img, sig, clin)
    (HF.cross_entropy(out[3], lab) + 0.1 * out[4]).backward()
    torch.cuda.synchronize()
    want = g_plain.clone()

    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29571")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
        created = True
    try:
        model = _model(cd)
        flatten(model, order=reduction_order(model))
        ddp = DataParallel(model, force=True)
        assert ddp.overlap and model.image_encoder._spec.stage_hook is not None
        for _ in range(2):                       # second pass: the side-stream events are reused across steps
            ddp.flat_g.zero_()
            HF.release_grads(model)              # what optimizer.zero_grad() does between backward passes
            ddp.prepare_backward()
            out = model(img, sig, clin)
            (HF.cross_entropy(out[3], lab) + 0.1 * out[4]).backward()
            ddp.reduce_gradients()
            torch.cuda.synchronize()
            assert torch.equal(ddp.flat_g, want)
    finally:
        model.image_encoder._spec.stage_hook = None
        model.image_encoder._spec.stage_groups = None
        if created:
            dist.destroy_process_group()


def test_two_ranks_rccl_average_of_shard_gradients_and_identical_parameters():
    """The real multi-rank path: 2 processes, one per GPU, RCCL all-reduce (tests/ddp_rank_worker.py holds the checks).
    Needs two GPUs: skipped on the one-GPU boxes; the world-size-2 logic itself also runs on CPU under gloo
    (tests/test_host_cpu.py)."""
    import socket
    import subprocess
    import sys
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ddp_rank_worker.py")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), worker],
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ddp2 ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
