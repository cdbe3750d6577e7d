"""Generate tests/golden/*.npz -- run ONLY in the build container (needs /root/reference).

TEST INFRASTRUCTURE ONLY.  What it pins:
  g1  reference ResNet1D_SE(1,2) + best_ptbxl.pth, eval, hash inputs   (reference class, imported)
  g2  same net, train mode (dropout p=0), CE loss, fwd+bwd             (reference class, imported)
  g3  reference 12-lead ResNet1D_SE + reference FocalLoss, 3 Adam+OneCycleLR steps (imported)
  g4  reference FocalLoss known answers                                 (imported)
  g5  oracle ECGMultimodalModel (restatement), B=8: eval/train outputs, grads, 3-step Adam losses
  g6  oracle ResNet18 (restatement of torchvision's): per-stage statistics
  g7  reference preprocess_signal / remove_baseline_drift on [12, 5000] float inputs    (imported)
  g9  the REFERENCE's own ECGMultimodalModel classes (multimodal_paper_modal_balance.py and multimodal.py, imported
      with stand-ins for the absent third-party packages / checkpoint files, see _import_reference_module): the
      oracle is asserted bit-identical, the reference objects' outputs, gradients and 3-step Adam losses are stored;
      `python oracle/make_golden.py g9` regenerates only this file
  g8  the image transform: Pillow's own BILINEAR resize (the arithmetic torchvision's Resize runs on a PIL
      picture; torchvision itself is absent here and unpinned in the reference) of formula pictures,
      + float32 ToTensor/Normalize; `python oracle/make_golden.py g8` regenerates only this file
and, before writing g1-g3, that oracle.ref_models.ResNet1D_SE is BIT-IDENTICAL to the reference
class on the same weights/inputs (it is the same sequence of torch ops).

The reference's source never leaves /root/reference: only tensors (inputs by formula, expected
outputs, and a tensor copy of best_ptbxl.pth's 79 entries) are written.
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")

from oracle import fill, ref_models as O  # noqa: E402


def npy(t):
    return t.detach().cpu().numpy()


G8_CASES = [(250, 2500, 224, 224), (300, 400, 224, 224), (100, 120, 224, 224), (37, 53, 16, 20), (500, 224, 224, 224)]


def g8_image():
    from PIL import Image
    from oracle import image_ref as IR
    g8 = {"cases": np.array(G8_CASES, np.int32)}
    for i, (h, w, oh, ow) in enumerate(G8_CASES):
        img = IR.synthetic_ecg_picture(h, w, 31 + i)
        pil = np.asarray(Image.fromarray(img, "RGB").resize((ow, oh), Image.BILINEAR))   # == transforms.Resize((oh, ow))
        assert np.array_equal(pil, IR.resize_bilinear_u8(img, oh, ow)), f"oracle resize differs from Pillow on case {i}"
        g8[f"resized_{i}"] = pil
        g8[f"sha_in_{i}"] = np.frombuffer(__import__("hashlib").sha256(img.tobytes()).digest(), np.uint8)
    # ToTensor + Normalize exactly as torch does it in float32 (checked against torch ops here)
    u = np.arange(256, dtype=np.uint8)
    t = (torch.from_numpy(u).to(torch.float32).div(255) - 0.5) / 0.5
    assert np.array_equal(t.numpy(), IR.to_tensor_normalize(u.reshape(16, 16, 1).repeat(3, 2))[0].reshape(-1))
    g8["normalize_lut"] = t.numpy()
    np.savez_compressed(os.path.join(OUT, "g8_image.npz"), **g8)


def _import_reference_module(name):
    """Import one of the reference's own model files (/root/reference/<name>.py) in this container.  The file is
    executed as it stands; what this image lacks -- third-party packages the file imports at module level and the
    checkpoint files its constructor hard-loads, none of which is reference code -- is stood in for:
      torchvision.models.resnet18()   -> oracle.ref_models.ResNet18 (the restatement; that sub-graph STAYS unpinned)
      pytorch_tabnet TabNetNoEmbeddings -> oracle.tabnet_ref.TabNetNoEmbeddings (restatement; STAYS unpinned)
      seaborn                           -> empty module (plot helper import only)
      torch.load('./checkpoints/...')   -> {} (files not in the tree: multimodal_paper_modal_balance.py:215,234-237,
                                           multimodal.py:350,369-372,388); load_state_dict(strict=False) then keeps the
                                           hash-filled parameters
    So what IS pinned to the reference's own code: AttentionFusion, the clinical MLP, the LayerNorms, branch heads,
    fusion_classifier, forward's composition order and var_loss, and (again) ResNet1D_SE."""
    import importlib
    import types
    from oracle import tabnet_ref as T
    tv, tvm = types.ModuleType("torchvision"), types.ModuleType("torchvision.models")
    tvm.resnet18 = lambda *a, **k: O.ResNet18()
    tvm.ResNet18_Weights = type("ResNet18_Weights", (), {"IMAGENET1K_V1": None})
    tv.models = tvm
    pt, ptn = types.ModuleType("pytorch_tabnet"), types.ModuleType("pytorch_tabnet.tab_network")
    ptn.TabNetNoEmbeddings = T.TabNetNoEmbeddings
    pt.tab_network = ptn
    for k, v in (("torchvision", tv), ("torchvision.models", tvm), ("pytorch_tabnet", pt),
                 ("pytorch_tabnet.tab_network", ptn), ("seaborn", types.ModuleType("seaborn"))):
        sys.modules.setdefault(k, v)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    return importlib.import_module(name)


def g9_reference_composition():
    """g9: the REFERENCE's ECGMultimodalModel objects (both variants), hash-filled, against the oracle: bit-identity of
    the 6-tuple, the train.py loss and every gradient is asserted here; the reference objects' outputs are the golden."""
    import contextlib
    import io
    real_load = torch.load

    def fake_load(f, *a, **k):
        if isinstance(f, str) and f.startswith("./checkpoints/"):
            return {}
        return real_load(f, *a, **k)

    sys.path.insert(0, REF)
    import config as RC                                   # the reference's Config (device = cpu here)
    g9 = {}
    for tag, modname, clin_in, make_oracle in (
            ("pmb", "multimodal_paper_modal_balance", 24, lambda: O.ECGMultimodalModel(RC.Config.num_classes, 24)),
            ("tab", "multimodal", 2, lambda: __import__("oracle.tabnet_ref", fromlist=["x"]).multimodal_tabnet_model(RC.Config.num_classes))):
        mod = _import_reference_module(modname)
        torch.load = fake_load
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                ref = mod.ECGMultimodalModel(RC.Config)
        finally:
            torch.load = real_load
        mine = make_oracle()
        assert list(ref.state_dict()) == list(mine.state_dict()), "state-dict keys differ from the reference object"
        for m in (ref, mine):
            fill.hash_fill_module(m, "mm.")
            O.disable_dropout(m)
        img, sig, clin, lab = fill.synthetic_batch(8, clin_dim=clin_in, salt=9)
        outs = {}
        for name, m in (("ref", ref), ("mine", mine)):
            m.eval()
            with torch.no_grad():
                ev = m(img, sig, clin)
            m.train()
            m.zero_grad()
            tr = m(img, sig, clin)
            loss = F.cross_entropy(tr[3], lab) + 0.1 * tr[4]           # train.py:69-78
            loss.backward()
            outs[name] = (ev, tr, loss.detach(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
        for a, b in zip(outs["ref"][0] + outs["ref"][1], outs["mine"][0] + outs["mine"][1]):
            assert torch.equal(a, b), f"{tag}: oracle output differs from the reference object's"
        assert torch.equal(outs["ref"][2], outs["mine"][2])
        assert outs["ref"][3].keys() == outs["mine"][3].keys()
        for k in outs["ref"][3]:
            assert torch.equal(outs["ref"][3][k], outs["mine"][3][k]), (tag, k)
        names = ("img_logits", "sig_logits", "clin_logits", "fusion_logits", "var_loss", "soft_w")
        for n, o in zip(names, outs["ref"][0]):
            g9[f"{tag}.eval.{n}"] = npy(o)
        for n, o in zip(names, outs["ref"][1]):
            g9[f"{tag}.train.{n}"] = npy(o)
        g9[f"{tag}.train.loss"] = npy(outs["ref"][2])
        for k, g in outs["ref"][3].items():
            g9[f"{tag}.gnorm.{k}"] = np.array(g.norm().item())
            if "encoder" not in k or k.startswith("clinical_encoder"):
                g9[f"{tag}.grad.{k}"] = npy(g)                          # every head / fusion / clinical gradient in full
        # three Adam steps of the reference object (train_paper_modal_balance.py:29: all parameters trainable)
        opt = torch.optim.Adam(ref.parameters(), lr=1e-4)
        ls = []
        for _ in range(3):
            opt.zero_grad()
            o = ref(img, sig, clin)
            l = F.cross_entropy(o[3], lab) + 0.1 * o[4]
            l.backward(); opt.step(); ls.append(l.item())
        g9[f"{tag}.adam3"] = np.array(ls)
        print(f"g9[{tag}]: reference {modname}.ECGMultimodalModel == oracle bit for bit "
              f"({len(outs['ref'][3])} gradients); loss {outs['ref'][2].item():.6f}")
    np.savez_compressed(os.path.join(OUT, "g9_reference_composition.npz"), **g9)


def main():
    if sys.argv[1:] == ["g8"]:
        g8_image()
        print("g8 written")
        return
    if sys.argv[1:] == ["g9"]:
        g9_reference_composition()
        print("g9 written")
        return
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    sys.path.insert(0, REF)
    import train_signal_only as R1          # reference ResNet1D_SE (1-lead script)
    import train_signal_12_af as R12        # reference ResNet1D_SE + FocalLoss (12-lead script)

    # ---------------- g1 / g2: best_ptbxl.pth through the reference class --------------------
    sd = torch.load(os.path.join(REF, "best_ptbxl.pth"), map_location="cpu", weights_only=True)
    np.savez_compressed(os.path.join(OUT, "best_ptbxl_tensors.npz"), **{k: npy(v) for k, v in sd.items()})
    ref = R1.ResNet1D_SE(input_channels=1, num_classes=2)
    ref.load_state_dict(sd, strict=True)
    mine = O.ResNet1D_SE(1, 2)
    mine.load_state_dict(sd, strict=True)
    g1 = {}
    for L in (2476, 5000):
        x = fill.hash_tensor((4, 1, L), 77 + L, 1.5)
        ref.eval(); mine.eval()
        with torch.no_grad():
            yr = ref(x)
            ym = mine(x)
            st = mine.stages(x)
        assert torch.equal(yr, ym), "oracle ResNet1D_SE is not bit-identical to the reference class"
        g1[f"logits_{L}"] = npy(yr)
        for i, s in enumerate(st):
            g1[f"stage{i}_mean_{L}"] = npy(s.mean(dim=(0, 2)))
            g1[f"stage{i}_absmean_{L}"] = npy(s.abs().mean(dim=(0, 2)))
        g1[f"pooled_{L}"] = npy(st[-1].mean(dim=2))
    np.savez_compressed(os.path.join(OUT, "g1_ptbxl_eval.npz"), **g1)

    g2 = {}
    for net in (ref, mine):
        net.load_state_dict(sd, strict=True)
        net.train()
        for m in net.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
    x = fill.hash_tensor((4, 1, 2476), 91, 1.5)
    y = torch.tensor([0, 1, 1, 0])
    outs = []
    for net in (ref, mine):
        net.zero_grad()
        lo = net(x)
        loss = F.cross_entropy(lo, y)
        loss.backward()
        outs.append((lo.detach(), loss.detach(), {k: p.grad.clone() for k, p in net.named_parameters()},
                     {k: v.clone() for k, v in net.state_dict().items() if "running" in k}))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    for k in outs[0][2]:
        assert torch.equal(outs[0][2][k], outs[1][2][k]), k
    g2["logits"], g2["loss"] = npy(outs[0][0]), npy(outs[0][1])
    for k, v in outs[0][2].items():
        g2["grad." + k] = npy(v)
    for k, v in outs[0][3].items():
        g2["buf." + k] = npy(v)
    np.savez_compressed(os.path.join(OUT, "g2_ptbxl_train.npz"), **g2)

    # ---------------- g3: 12-lead, reference FocalLoss, Adam + OneCycleLR ---------------------
    ref12 = R12.ResNet1D_SE(input_channels=12)
    mine12 = O.ResNet1D_SE(12, 2)
    fill.hash_fill_module(ref12, "sig12.")
    fill.hash_fill_module(mine12, "sig12.")
    x12 = fill.hash_tensor((8, 12, 5000), 555, 1.5)
    y12 = torch.tensor([0, 1, 1, 0, 1, 0, 0, 1])
    traj = []
    for net, crit in ((ref12, R12.FocalLoss(alpha=1.0, gamma=2.0)), (mine12, O.FocalLoss(1.0, 2.0))):
        net.train()
        for m in net.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        opt = torch.optim.Adam(net.parameters(), lr=1e-3)
        sch = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-3, steps_per_epoch=4, epochs=30)
        losses, lrs, b1s = [], [], []
        for _ in range(3):
            lrs.append(opt.param_groups[0]["lr"]); b1s.append(opt.param_groups[0]["betas"][0])
            opt.zero_grad()
            loss = crit(net(x12), y12)
            loss.backward()
            opt.step(); sch.step()
            losses.append(loss.item())
        net.eval()
        with torch.no_grad():
            fin = net(x12)
        traj.append((losses, fin, lrs, b1s))
    assert traj[0][0] == traj[1][0] and torch.equal(traj[0][1], traj[1][1])
    np.savez_compressed(os.path.join(OUT, "g3_sig12_steps.npz"), losses=np.array(traj[0][0]),
                        final_logits=npy(traj[0][1]), lrs=np.array(traj[0][2]), beta1s=np.array(traj[0][3]))

    # ---------------- g4: FocalLoss known answers ---------------------------------------------
    lg = torch.tensor([[2.0, -1.0], [0.3, 0.1]]); tg = torch.tensor([0, 1])
    v_ref = R12.FocalLoss()(lg, tg)
    assert abs(v_ref.item() - 0.12070029228925705) < 1e-9            # SURVEY 8c known answer
    lg2 = fill.hash_tensor((16, 2), 4242, 3.0); tg2 = torch.from_numpy((fill.hash_uniform(16, 4243) > 0).astype(np.int64))
    v2 = R12.FocalLoss(alpha=0.25, gamma=2.0)(lg2, tg2)
    v2u = R12.FocalLoss(reduce=False)(lg2, tg2)
    assert torch.equal(v2, O.FocalLoss(0.25, 2.0)(lg2, tg2))
    np.savez_compressed(os.path.join(OUT, "g4_focal.npz"), kat=npy(v_ref), logits=npy(lg2), targets=npy(tg2),
                        loss_a025=npy(v2), loss_unreduced=npy(v2u))

    # ---------------- g5: full multimodal oracle (restatement) --------------------------------
    g5 = {}
    img, sig, clin, lab = fill.synthetic_batch(8, salt=5)
    model = O.disable_dropout(fill.hash_fill_module(O.ECGMultimodalModel(2, 16), "mm."))
    model.eval()
    with torch.no_grad():
        out = model(img, sig, clin)
    for n, o in zip(("img_logits", "sig_logits", "clin_logits", "fusion_logits", "var_loss", "soft_w"), out):
        g5["eval." + n] = npy(o)
    model.train()
    model.zero_grad()
    out = model(img, sig, clin)
    loss = O.multimodal_loss(out, lab)
    loss.backward()
    for n, o in zip(("img_logits", "sig_logits", "clin_logits", "fusion_logits", "var_loss", "soft_w"), out):
        g5["train." + n] = npy(o)
    g5["train.loss"] = npy(loss)
    for k, p in model.named_parameters():
        g5["gnorm." + k] = np.array(0.0 if p.grad is None else p.grad.norm().item())
    for k in ("image_encoder.conv1.weight", "image_encoder.layer4.1.conv2.weight", "image_encoder.fc.weight"):
        g5["gslice." + k] = npy(dict(model.named_parameters())[k].grad.flatten()[:512])
    for k in ("fusion_classifier.0.weight", "attention_fusion.weights", "signal_encoder.layer3.se.fc.0.weight",
              "clinical_encoder.0.weight", "signal_encoder.initial.0.weight"):
        g5["grad." + k] = npy(dict(model.named_parameters())[k].grad)
    for frozen in (False, True):
        m = O.disable_dropout(fill.hash_fill_module(O.ECGMultimodalModel(2, 16), "mm."))
        if frozen:
            O.freeze_encoders(m)
        m.train()
        opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-4)
        ls = []
        for _ in range(3):
            opt.zero_grad()
            l = O.multimodal_loss(m(img, sig, clin), lab)
            l.backward(); opt.step(); ls.append(l.item())
        g5["adam3.frozen" if frozen else "adam3.unfrozen"] = np.array(ls)
    np.savez_compressed(os.path.join(OUT, "g5_multimodal.npz"), **g5)

    # ---------------- g6: ResNet18 restatement, per-stage statistics --------------------------
    g6 = {}
    r18 = fill.hash_fill_module(O.ResNet18(num_classes=256), "r18.")
    assert sum(p.numel() for p in r18.parameters()) == 11307840
    assert len(O.ResNet18().state_dict()) == 122
    for tag, shape in (("224", (2, 3, 224, 224)), ("250x2500", (1, 3, 250, 2500))):
        x = fill.hash_tensor(shape, 606)
        r18.eval()
        with torch.no_grad():
            st = r18.stages(x)
            g6[f"feat_{tag}"] = npy(r18(x))
        for i, s in enumerate(st):
            g6[f"shape{i}_{tag}"] = np.array(s.shape)
            g6[f"stage{i}_mean_{tag}"] = npy(s.mean(dim=(0, 2, 3)))
    r18.train()
    x = fill.hash_tensor((4, 3, 64, 64), 607)
    f = r18(x)
    f.square().mean().backward()
    g6["train_feat_64"] = npy(f)
    g6["train_gnorm_conv1_64"] = np.array(r18.conv1.weight.grad.norm().item())
    g6["train_gnorm_l4_64"] = np.array(r18.layer4[1].conv2.weight.grad.norm().item())
    g6["train_bn1_rm_64"] = npy(r18.bn1.running_mean)
    np.savez_compressed(os.path.join(OUT, "g6_resnet18.npz"), **g6)
    # ---------------- g7: signal pre-processing, the REFERENCE's own function ---------------------
    from oracle import preprocess_ref as PR
    x7 = fill.hash_tensor((12, 5000), 707, 2.0).numpy().astype(np.float64)
    x7 += np.linspace(-1.5, 2.0, 5000)[None, :] + 0.8 * np.sin(np.arange(5000) / 37.0)[None, :]   # drift + rhythm
    y_ref = R12.preprocess_signal(x7)                   # train_signal_12_af.py:30-34
    assert np.array_equal(y_ref, PR.preprocess_signal(x7)), "oracle preprocess_signal differs from the reference's"
    x7s = x7[:3, :1000]
    np.savez_compressed(os.path.join(OUT, "g7_preprocess.npz"), x=x7.astype(np.float32),
                        y=R12.preprocess_signal(x7.astype(np.float32).astype(np.float64)),
                        y_short=R12.preprocess_signal(x7s.astype(np.float32).astype(np.float64)),
                        baseline_removed=R12.remove_baseline_drift(x7.astype(np.float32).astype(np.float64)))
    g8_image()
    g9_reference_composition()
    print("goldens written to", OUT)
    for f_ in sorted(os.listdir(OUT)):
        print(f"  {f_}: {os.path.getsize(os.path.join(OUT, f_)) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
