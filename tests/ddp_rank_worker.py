"""One rank of the 2-rank RCCL data-parallel test (started by torch.distributed.run from
tests/test_parallel_gpu.py::test_two_ranks_rccl...).  Checks SURVEY 8e's parity statement: the average of the
shard gradients equals the gradient of the full batch when BatchNorm statistics are per replica -- here against a
single-process run that feeds each shard through its OWN replica (same per-replica statistics) -- and parameters
stay identical across ranks after Adam."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    from ecgmm.config import Config
    from ecgmm.hip import functional as HF
    from ecgmm.multimodal_paper_modal_balance import ECGMultimodalModel
    from ecgmm.optim import FusedAdam
    from ecgmm.parallel import DataParallel, flatten, reduction_order
    from oracle import fill

    def build(prefix):
        cfg = type("C", (Config,), {})
        cfg.compute_dtype, cfg.clinical_input_dim, cfg.num_classes = "fp32", 16, 2
        m = fill.hash_fill_module(ECGMultimodalModel(cfg), prefix)
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        return m.to(dev).train()

    per = 4
    img, sig, clin, lab = (t.to(dev) for t in fill.synthetic_batch(per * world, img_hw=(64, 64), sig_len=1000, salt=21))
    sl = slice(rank * per, (rank + 1) * per)

    # every rank starts from DIFFERENT parameters: the wrapper must broadcast rank 0's
    model = build("mm." if rank == 0 else f"other{rank}.")
    flatten(model, order=reduction_order(model))      # the layout bench.py uses (stage groups = contiguous ranges)
    ddp = DataParallel(model)
    opt = FusedAdam(model.parameters(), lr=1e-3, grad_scale=ddp.grad_scale)
    opt.zero_grad()
    ddp.prepare_backward()
    out = ddp(img[sl], sig[sl], clin[sl])
    (HF.cross_entropy(out[3], lab[sl]) + 0.1 * out[4]).backward()
    ddp.reduce_gradients()
    torch.cuda.synchronize()
    g_avg = ddp.flat_g.clone() * ddp.grad_scale

    # reference on this one GPU: each shard through its own replica of rank 0's parameters, gradients averaged
    acc = None
    for r in range(world):
        ref = build("mm.")
        _, g = flatten(ref, order=reduction_order(ref))
        s = slice(r * per, (r + 1) * per)
        o = ref(img[s], sig[s], clin[s])
        (HF.cross_entropy(o[3], lab[s]) + 0.1 * o[4]).backward()
        torch.cuda.synchronize()
        acc = g.clone() if acc is None else acc + g
    want = acc / world
    err = ((g_avg - want).norm() / want.norm()).item()
    assert err < 1e-5, f"rank {rank}: averaged shard gradient differs from the per-replica reference by {err}"

    opt.step()
    torch.cuda.synchronize()
    gathered = [torch.empty_like(ddp.flat_p) for _ in range(world)]
    dist.all_gather(gathered, ddp.flat_p)
    assert all(torch.equal(gathered[0], t) for t in gathered), "parameters diverged across ranks after Adam"
    bn = model.image_encoder.bn1.running_mean
    allbn = [torch.empty_like(bn) for _ in range(world)]
    dist.all_gather(allbn, bn)
    assert not torch.equal(allbn[0], allbn[1]), "BatchNorm statistics should stay per replica (different shards)"
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(f"ddp2 ok: rel err {err:.2e}")


if __name__ == "__main__":
    main()
