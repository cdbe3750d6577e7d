"""CPU: host-side logic that needs no GPU -- module trees / state-dict compatibility with the oracle
(== reference key names), config surface, datasets, FusedAdam run planning, and the data-parallel
wrapper under a world-size-2 gloo group."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from ecgmm.config import Config
from ecgmm.dataset import ECGMultimodalDataset, get_dataloaders
from ecgmm.multimodal_paper_modal_balance import ECGMultimodalModel, MultimodalModel, ResNet1D_SE
from ecgmm.signal_model import ECGDataset, FocalLoss
from ecgmm.train_image_only import ImageOnlyClassifier
from oracle import ref_models as O


def test_state_dict_keys_and_shapes_match_reference_layout(golden_dir):
    cfg = type("C16", (Config,), {"clinical_input_dim": 16})
    mine, ref = ECGMultimodalModel(cfg), O.ECGMultimodalModel(2, 16)
    sm, sr = mine.state_dict(), ref.state_dict()
    assert list(sm) == list(sr)
    assert all(sm[k].shape == sr[k].shape and sm[k].dtype == sr[k].dtype for k in sr)
    assert MultimodalModel is ECGMultimodalModel
    mine.load_state_dict(sr, strict=True)
    # the reference's only real checkpoint loads strictly into the signal encoder
    sd = {k: torch.from_numpy(v) for k, v in np.load(f"{golden_dir}/best_ptbxl_tensors.npz").items()}
    ResNet1D_SE(1, 2).load_state_dict(sd, strict=True)
    # partial loader semantics of PMB:309-322 (drops classifier.4.*)
    torch.save(sd, "/tmp/_ptbxl_test.pth")
    before = mine.signal_encoder.classifier[4].weight.clone()
    mine.load_pretrained_signal_encoder("/tmp/_ptbxl_test.pth", load_fc=False)
    assert torch.equal(mine.signal_encoder.classifier[4].weight, before)
    assert torch.equal(mine.signal_encoder.initial[0].weight, sd["initial.0.weight"])
    assert list(ImageOnlyClassifier().state_dict())[0] == "image_encoder.conv1.weight"
    for attr in ("image_encoder", "image_norm", "signal_encoder", "signal_norm", "clinical_encoder", "clinical_norm",
                 "attention_fusion", "fusion_classifier", "image_classifier", "signal_classifier",
                 "clinical_classifier", "modal_dim"):
        assert hasattr(mine, attr)
    assert mine.get_clinical_feature_dim() == 16 and ECGMultimodalModel(Config).get_clinical_feature_dim() == 24


def test_config_surface():
    for k, v in dict(seed=42, img_height=224, img_width=224, num_classes=2, batch_size=16, num_epochs=30, lr=1e-4,
                     patience=5, k_outer=5, k_inner=3, checkpoint_dir="./checkpoints").items():
        assert getattr(Config, k) == v
    assert Config.device in ("cuda", "cpu") and Config.ecg_csv.endswith("ecg_signals.csv")


def test_product_modules_have_no_cpu_path():
    cfg = type("C16", (Config,), {"clinical_input_dim": 16})
    m = ECGMultimodalModel(cfg)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(2, 3, 64, 64), torch.zeros(2, 500), torch.zeros(2, 16))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        FocalLoss()(torch.zeros(2, 2), torch.zeros(2, dtype=torch.long))


def test_datasets():
    ds = ECGMultimodalDataset.synthetic(5, Config)
    img, sig, clin, lab, idx = ds[3]
    assert img.shape == (3, 224, 224) and img.dtype == torch.float32 and img.abs().max() <= 1
    assert sig.shape == (5000,) and clin.shape == (24,) and lab.dtype == torch.int64 and idx == 3
    assert torch.equal(ds[3][1], sig)
    cfg = type("Small", (Config,), {"synthetic_train_size": 32, "batch_size": 8})
    tr, va, te = get_dataloaders(cfg)
    *batch, index = next(iter(tr))
    assert [t.shape[0] for t in batch] == [8, 8, 8, 8] and len(index) == 8
    d = ECGDataset(np.ones((3, 7)), [0, 1, 0])
    assert len(d) == 3 and d[1][0].dtype == torch.float32 and d[1][1].dtype == torch.int64


def test_file_datasets_follow_the_reference_tables(tmp_path):
    """real-file path, host half: filtering, index intersection, stratified split, scalers, raw items"""
    from ecgmm import dataset as D
    from oracle import dataset_ref as DR
    ids = DR.write_tiny_dataset(str(tmp_path), n=40, sig_len=600, hw=(50, 500))
    cfg = type("Files", (Config,), {"synthetic": False, "data_dir": str(tmp_path), "image_dir": str(tmp_path / "images"),
                                    "ecg_csv": str(tmp_path / "ecg_signals.csv"), "label_file": str(tmp_path / "labels.xlsx"),
                                    "clinical_file": str(tmp_path / "clinical.csv"), "batch_size": 4})
    labels_df, ecg, clin = D.load_tables(cfg)                       # labels.xlsx -> labels.csv fallback (no openpyxl here)
    assert len(labels_df) == 38 and "ECG" not in clin.columns       # one Borderline, one subject without a picture
    assert set(labels_df["label"]) == {0, 1} and list(ecg.index) == list(labels_df["index"])
    sets, ecg_scaler, clin_scaler = D.build_file_datasets(cfg)
    assert [len(s) for s in sets] == [30, 4, 4]
    assert not (set(sets[0].labels_df["index"]) & set(sets[2].labels_df["index"]))
    tr_rows = ecg.loc[ecg.index.isin(sets[0].labels_df["index"])].values
    assert np.allclose(ecg_scaler.mean_, tr_rows.mean(0)) and np.allclose(ecg_scaler.scale_, tr_rows.std(0))
    pic, sig, c, lab, index = sets[1][2]
    assert pic.dtype == torch.uint8 and pic.shape == (50, 500, 3) and sig.shape == (600,) and sig.dtype == torch.float32
    assert c.shape == (2,) and lab.dtype == torch.int64 and index in ids
    assert np.allclose(sig.numpy(), ecg.loc[index].values.astype(np.float32))


def test_tabnet_restatement_and_multimodal_py_variant_keys():
    """f3 host side: sparsemax known answer, the pytorch_tabnet key layout, and that ecgmm.multimodal's model (TabNet
    clinical branch, widths 512 / 128 / 32) has the state-dict of the restated multimodal.py model"""
    from oracle import tabnet_ref as T
    from ecgmm.multimodal import ECGMultimodalModel as TabNetVariant
    p = T.sparsemax(torch.tensor([[0.5, 0.2, -1.0], [3.0, 1.0, 0.0]]))
    assert torch.allclose(p, torch.tensor([[0.65, 0.35, 0.0], [1.0, 0.0, 0.0]]), atol=1e-6)
    ref = T.multimodal_tabnet_model(2)
    ours = TabNetVariant(Config)
    assert list(ours.state_dict()) == list(ref.state_dict())
    assert [tuple(v.shape) for v in ours.state_dict().values()] == [tuple(v.shape) for v in ref.state_dict().values()]
    keys = [k for k in ours.state_dict() if k.startswith("clinical_encoder.")]
    assert len(keys) == 120 and "clinical_encoder.tabnet.encoder.initial_splitter.shared.glu_layers.0.fc.weight" in keys
    assert "clinical_encoder.tabnet.final_mapping.weight" in keys
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ours.clinical_encoder(torch.zeros(4, 2))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _ddp_worker(rank, world, port, out):
    import torch.distributed as dist
    from ecgmm.parallel import DataParallel, flatten
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)          # different init per rank: the wrapper must broadcast rank 0's
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    flatten(net)
    ddp = DataParallel(net, bucket_mb=1e-5)            # tiny buckets: exercise the bucket loop
    g = torch.Generator().manual_seed(7)
    X, Y = torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)
    xs, ys = X[rank * 4:(rank + 1) * 4], Y[rank * 4:(rank + 1) * 4]   # contiguous shards
    ddp.prepare_backward()
    loss = ((ddp(xs) - ys) ** 2).mean()
    for p in net.parameters():
        p.grad.zero_()
    loss.backward()                                     # plain torch autograd accumulates into the flat views
    ddp.reduce_gradients()
    flat_p, flat_g = net._ecg_flat[0].clone(), net._ecg_flat[1].clone() * ddp.grad_scale
    if rank == 0:
        ref = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
        with torch.no_grad():
            for pr, pm in zip(ref.parameters(), net.parameters()):
                pr.copy_(pm)
        ((ref(X) - Y) ** 2).mean().backward()
        gref = torch.cat([(p.grad.reshape(-1)) for p in ref.parameters()])
        mine = torch.cat([p.grad.reshape(-1) * ddp.grad_scale for p in net.parameters()])
        out.put((torch.allclose(mine, gref, atol=1e-6), float((mine - gref).abs().max())))
    gathered = [torch.zeros_like(flat_p) for _ in range(world)]
    dist.all_gather(gathered, flat_p)
    if rank == 0:
        out.put(all(torch.equal(gathered[0], t) for t in gathered))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_gloo_world2_matches_full_batch_gradient():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    ok, err = q.get(timeout=10)
    assert ok, f"averaged shard gradients differ from the full-batch gradient by {err}"
    assert q.get(timeout=10), "parameters were not identical across ranks after the initial broadcast"


def test_fused_adam_run_planning_cpu():
    """Run merging over the padded flat buffer needs no GPU: only pointer arithmetic."""
    from ecgmm.optim import FusedAdam
    from ecgmm.parallel import flatten
    net = torch.nn.Sequential(torch.nn.Linear(3, 2), torch.nn.Linear(2, 3))   # sizes 6, 2, 6, 3: all need padding
    flatten(net)
    opt = FusedAdam(net.parameters(), lr=1e-3)
    opt._build_runs()
    assert len(opt._runs) == 1 and opt._runs[0]["n"] == 8 + 4 + 8 + 3 and opt._runs_valid()
    net[0].weight.requires_grad = False
    opt2 = FusedAdam([p for p in net.parameters() if p.requires_grad], lr=1e-3)
    opt2._build_runs()
    assert len(opt2._runs) == 1 and opt2._runs[0]["params"][0] is net[0].bias


def test_bench_gpus_flag_launches_ranks_or_fails_loudly():
    """`bench.py --gpus N` must never silently run one rank: without a launcher environment it spawns N
    ranks itself (here: refuses, no GPUs), and under a launcher it insists on --gpus == WORLD_SIZE."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "only 0 GPU(s) visible" in r.stderr and "{" not in r.stdout
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8"], env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_fused_adam_never_creates_gradients_cpu():
    """torch.optim.Adam skips parameters whose grad is None; FusedAdam's run planning must too (pointer logic only)."""
    from ecgmm.optim import FusedAdam
    net = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.Linear(4, 4))
    net[0].weight.grad = torch.zeros_like(net[0].weight)
    opt = FusedAdam(net.parameters(), lr=1e-3)
    opt._build_runs()
    assert [len(r["params"]) for r in opt._runs] == [1] and opt._runs[0]["params"][0] is net[0].weight
    assert all(p.grad is None for p in list(net.parameters())[1:])
    assert opt._runs_valid()
    net[1].bias.grad = torch.zeros_like(net[1].bias)        # a parameter gains its gradient later: runs are re-planned
    assert not opt._runs_valid()


def test_flat_buffer_in_reduction_order_keeps_stage_groups_contiguous_and_one_adam_run():
    """parallel.reduction_order: [fc + layer4 | layer3 | everything else] -- every data-parallel stage group is one
    contiguous range of the flat gradient buffer, the rest (reduced after the backward) is ONE range, parameters are still
    views of the buffer, and FusedAdam (which plans in address order) still covers it with a single run."""
    from ecgmm.config import Config
    from ecgmm.multimodal_paper_modal_balance import ECGMultimodalModel
    from ecgmm.optim import FusedAdam
    from ecgmm.parallel import flatten, reduction_order
    cfg = type("C", (Config,), {"clinical_input_dim": 16})
    m = ECGMultimodalModel(cfg)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    order = reduction_order(m)
    assert sorted(map(id, order)) == sorted(map(id, m.parameters()))
    flat_p, flat_g = flatten(m, order=order)
    _, _, params, offs = m._ecg_flat
    enc = m.image_encoder
    pos = {id(p): (o, o + p.numel()) for p, o in zip(params, offs)}

    def span(mods):
        ps = [p for mod in mods for p in mod.parameters()]
        return min(pos[id(p)][0] for p in ps), max(pos[id(p)][1] for p in ps), sum((p.numel() + 3) // 4 * 4 for p in ps)

    lo1, hi1, n1 = span([enc.fc, enc.layer4])
    lo2, hi2, n2 = span([enc.layer3])
    assert lo1 == 0 and hi1 - lo1 <= n1 and lo2 >= hi1 and hi2 - lo2 <= n2      # contiguous, in this order
    rest_lo = min(pos[id(p)][0] for p in m.parameters() if not (lo1 <= pos[id(p)][0] < hi2))
    assert rest_lo >= hi2                                                         # everything else behind them: one range
    for k, v in m.state_dict().items():                                           # values unchanged, parameters are views
        assert torch.equal(v, before[k])
    p0 = next(iter(enc.layer4.parameters()))
    o0 = pos[id(p0)][0]
    assert p0.data_ptr() == flat_p.data_ptr() + 4 * o0 and p0.grad.data_ptr() == flat_g.data_ptr() + 4 * o0
    opt = FusedAdam(m.parameters(), lr=1e-3)
    opt._build_runs()
    assert len(opt._runs) == 1 and opt._runs[0]["n"] == flat_p.numel() - (flat_p.numel() - max(e for _, e in pos.values()))
    with pytest.raises(ValueError):
        flatten(ECGMultimodalModel(cfg), order=order[:-1])


@pytest.mark.parametrize("groups", ["3", "2"])
def test_stage_hook_ranges_cover_the_flat_buffer_exactly_once(groups, monkeypatch):
    """on the REAL model's parameter list, the gradient ranges DataParallel's backward-stage hooks ship
    early plus what reduce_gradients() ships afterwards cover the flat buffer exactly once (no element reduced twice --
    a double all-reduce would multiply that gradient by the world size -- and none left out), for both ECGMM_DDP_GROUPS
    settings; every range holds exactly the parameters of the encoder stages whose backward has finished by then."""
    from ecgmm.config import Config
    from ecgmm.multimodal_paper_modal_balance import ECGMultimodalModel
    from ecgmm.parallel import DataParallel, flatten, reduction_order
    monkeypatch.setenv("ECGMM_DDP_GROUPS", groups)
    cfg = type("C", (Config,), {"clinical_input_dim": 16})
    m = ECGMultimodalModel(cfg)
    flatten(m, order=reduction_order(m))
    ddp = DataParallel(m)                 # world 1, CPU: no process group needed; hooks are installed by hand below
    ddp._install_hooks()
    enc = m.image_encoder
    spec = enc._spec
    assert spec.stage_hook is not None and spec.stage_groups[0][0] == 0 and spec.stage_groups[-1][1] == spec.n_stages
    assert all(a[1] == b[0] for a, b in zip(spec.stage_groups, spec.stage_groups[1:]))    # stages 0..9, each once
    _, _, params, offs = m._ecg_flat
    span = {id(p): (o, o + (p.numel() + 3) // 4 * 4) for p, o in zip(params, offs)}
    owners = ([[enc.fc, enc.layer4], [enc.layer3], [enc.layer2, enc.layer1, enc.conv1, enc.bn1]] if groups == "3"
              else [[enc.fc, enc.layer4], [enc.layer3, enc.layer2, enc.layer1, enc.conv1, enc.bn1]])
    assert len(ddp._enc_ranges) == len(owners) == len(spec.stage_groups)
    for (lo, hi), mods in zip(ddp._enc_ranges, owners):
        mine = {id(p) for mod in mods for p in mod.parameters()}
        inside = {i for i, (a, b) in span.items() if a >= lo and b <= hi}
        assert inside == mine and sum(b - a for i, (a, b) in span.items() if i in mine) == hi - lo
    # what the hooks ship early (all groups but the last with ECGMM_DDP_TAIL_ON_COMPUTE=1, the default) + the remainder
    ddp.prepare_backward()
    shipped = []
    ddp._launch = lambda lo, hi, side=False: shipped.append((lo, hi))
    for gi in range(len(owners)):
        ddp._stage_hook(spec, gi)
    assert len(shipped) == len(owners) - 1
    pieces = sorted(shipped + ddp._remaining(ddp._done_ranges))
    assert pieces[0][0] == 0 and pieces[-1][1] == ddp.flat_g.numel()
    assert all(a[1] == b[0] for a, b in zip(pieces, pieces[1:]))                          # disjoint and gap-free
    assert len(ddp._remaining(ddp._done_ranges)) == 1                                     # ONE tail collective
    spec.stage_hook = spec.stage_groups = None


def test_fused_adam_keeps_one_step_count_per_parameter_cpu():
    """a parameter that gains its gradient later than its neighbours must not inherit their step count (its
    first bias correction would be off by ~3x): it starts a run of its own."""
    from ecgmm.optim import FusedAdam
    from ecgmm.parallel import flatten
    net = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.Linear(4, 4))
    flatten(net)
    late = net[1].weight
    late.grad = None
    opt = FusedAdam(net.parameters(), lr=1e-3)
    opt._build_runs()
    for r in opt._runs:                      # what step() records after two updates
        for p in r["params"]:
            opt.state[p]["step"] = 2
    late.grad = late._ecg_grad_view
    assert not opt._runs_valid()
    opt._build_runs()
    steps = {id(p): r["step"] for r in opt._runs for p in r["params"]}
    assert steps[id(late)] == 0 and steps[id(net[0].weight)] == 2 and steps[id(net[1].bias)] == 2
    assert sum(len(r["params"]) for r in opt._runs) == 4 and len(opt._runs) == 3


def test_pretrained_loaders_follow_the_reference(tmp_path):
    """PMB:309-322 / :356-384: the signal loader drops `classifier.4*` unless load_fc, strict=False; the image loader
    drops `fc.*` unless load_fc and RAISES on a tensor of the wrong shape (the reference loads through a temporary
    resnet18, whose load_state_dict raises)."""
    from ecgmm.config import Config
    from ecgmm.multimodal_paper_modal_balance import ECGMultimodalModel
    m = ECGMultimodalModel(type("C", (Config,), {"clinical_input_dim": 16}))
    sd = {k: torch.full_like(v, 0.25) if v.is_floating_point() else v.clone() for k, v in m.image_encoder.state_dict().items()}
    torch.save(sd, tmp_path / "img.pth")
    fc_before = m.image_encoder.fc.weight.clone()
    m.load_pretrained_image_encoder(str(tmp_path / "img.pth"), load_fc=False)
    assert float(m.image_encoder.layer3[1].conv2.weight.mean()) == 0.25 and torch.equal(m.image_encoder.fc.weight, fc_before)
    m.load_pretrained_image_encoder(str(tmp_path / "img.pth"), load_fc=True)
    assert float(m.image_encoder.fc.weight.mean()) == 0.25
    sd["layer1.0.conv1.weight"] = torch.zeros(64, 32, 3, 3)
    torch.save(sd, tmp_path / "bad.pth")
    with pytest.raises(RuntimeError, match="size mismatch"):
        m.load_pretrained_image_encoder(str(tmp_path / "bad.pth"))
    sd = {k: torch.full_like(v, 0.5) if v.is_floating_point() else v.clone() for k, v in m.signal_encoder.state_dict().items()}
    torch.save(sd, tmp_path / "sig.pth")
    last = m.signal_encoder.classifier[4].weight.clone()
    m.load_pretrained_signal_encoder(str(tmp_path / "sig.pth"))
    assert float(m.signal_encoder.layer2.conv1.weight.mean()) == 0.5 and torch.equal(m.signal_encoder.classifier[4].weight, last)
