"""CPU: the oracle restatement reproduces the committed goldens (g1-g4 were generated from the
REFERENCE's own classes in the build container by oracle/make_golden.py)."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import fill, ref_models as O


def _sd(golden_dir):
    return {k: torch.from_numpy(v) for k, v in np.load(f"{golden_dir}/best_ptbxl_tensors.npz").items()}


def test_hash_fill_is_machine_independent():
    v = fill.hash_uniform(5, 77)
    assert v.dtype == np.float32 and np.all(np.abs(v) <= 1)
    assert np.array_equal(v, fill.hash_uniform(5, 77))
    # frozen values: any change to the hash would silently invalidate every golden
    assert np.array_equal(fill.hash_uniform(3, 1),
                          np.array([0.30292809009552, -0.631522536277771, 0.32319724559783936], dtype=np.float32))
    assert abs(float(fill.hash_uniform(1000, 3).mean())) < 0.1


def test_resnet1d_eval_matches_reference_golden(golden_dir):
    g1 = np.load(f"{golden_dir}/g1_ptbxl_eval.npz")
    net = O.ResNet1D_SE(1, 2)
    net.load_state_dict(_sd(golden_dir), strict=True)
    net.eval()
    for L in (2476, 5000):
        with torch.no_grad():
            out = net(fill.hash_tensor((4, 1, L), 77 + L, 1.5))
        assert torch.allclose(out, torch.from_numpy(g1[f"logits_{L}"]), atol=1e-5)


def test_resnet1d_train_step_matches_reference_golden(golden_dir):
    g2 = np.load(f"{golden_dir}/g2_ptbxl_train.npz")
    net = O.disable_dropout(O.ResNet1D_SE(1, 2))
    net.load_state_dict(_sd(golden_dir), strict=True)
    net.train()
    lo = net(fill.hash_tensor((4, 1, 2476), 91, 1.5))
    loss = F.cross_entropy(lo, torch.tensor([0, 1, 1, 0]))
    loss.backward()
    assert torch.allclose(lo.detach(), torch.from_numpy(g2["logits"]), atol=1e-5)
    assert abs(loss.item() - float(g2["loss"])) < 1e-6
    g = dict(net.named_parameters())["layer3.conv2.weight"].grad
    assert torch.allclose(g, torch.from_numpy(g2["grad.layer3.conv2.weight"]), rtol=1e-3, atol=1e-7)


def test_focal_loss_known_answers(golden_dir):
    g4 = np.load(f"{golden_dir}/g4_focal.npz")
    v = O.FocalLoss()(torch.tensor([[2.0, -1.0], [0.3, 0.1]]), torch.tensor([0, 1]))
    assert abs(v.item() - 0.12070029228925705) < 1e-7
    assert abs(v.item() - float(g4["kat"])) < 1e-7
    v2 = O.FocalLoss(0.25, 2.0)(torch.from_numpy(g4["logits"]), torch.from_numpy(g4["targets"]))
    assert abs(v2.item() - float(g4["loss_a025"])) < 1e-7
    v3 = O.FocalLoss(reduce=False)(torch.from_numpy(g4["logits"]), torch.from_numpy(g4["targets"]))
    assert torch.allclose(v3, torch.from_numpy(g4["loss_unreduced"]), atol=1e-7)


def test_sig12_adam_onecycle_trajectory(golden_dir):
    g3 = np.load(f"{golden_dir}/g3_sig12_steps.npz")
    net = O.disable_dropout(fill.hash_fill_module(O.ResNet1D_SE(12, 2), "sig12.")).train()
    x, y = fill.hash_tensor((8, 12, 5000), 555, 1.5), torch.tensor([0, 1, 1, 0, 1, 0, 0, 1])
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    sch = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-3, steps_per_epoch=4, epochs=30)
    crit = O.FocalLoss(1.0, 2.0)
    for i in range(3):
        assert abs(opt.param_groups[0]["lr"] - g3["lrs"][i]) < 1e-12
        assert abs(opt.param_groups[0]["betas"][0] - g3["beta1s"][i]) < 1e-9   # cycle_momentum rewrites beta1
        opt.zero_grad()
        loss = crit(net(x), y)
        loss.backward()
        opt.step(); sch.step()
        assert abs(loss.item() - g3["losses"][i]) < 2e-4


def test_structure_of_the_resnet18_restatement():
    r = O.ResNet18()
    keys = list(r.state_dict())
    assert len(keys) == 122 and keys[0] == "conv1.weight" and keys[-1] == "fc.bias"
    assert "layer2.0.downsample.1.num_batches_tracked" in keys and "layer1.0.downsample.0.weight" not in keys
    assert sum(p.numel() for p in r.parameters()) == 11689512
    m = O.ECGMultimodalModel(2, 16)
    assert sum(p.numel() for p in m.parameters()) == 11911975
    assert sum(p.numel() for p in O.ResNet1D_SE(12, 2).parameters()) == 471390


def test_multimodal_oracle_matches_g5(golden_dir):
    g5 = np.load(f"{golden_dir}/g5_multimodal.npz")
    model = O.disable_dropout(fill.hash_fill_module(O.ECGMultimodalModel(2, 16), "mm.")).eval()
    img, sig, clin, lab = fill.synthetic_batch(8, salt=5)
    with torch.no_grad():
        out = model(img, sig, clin)
    assert torch.allclose(out[3], torch.from_numpy(g5["eval.fusion_logits"]), atol=2e-5)
    assert abs(out[4].item() - float(g5["eval.var_loss"])) < 1e-5


def test_preprocess_oracle_and_native_filter_design(golden_dir):
    """g7 comes from the reference's own preprocess_signal; the host-side filter design (no scipy at run time)
    reproduces scipy's butter / lfilter_zi and the SURVEY 8c known answer."""
    from oracle import preprocess_ref as PR
    from ecgmm.preprocess import butter_lowpass, lfilter_zi
    from scipy.signal import butter, lfilter_zi as sp_zi
    g7 = np.load(f"{golden_dir}/g7_preprocess.npz")
    x = g7["x"].astype(np.float64)
    assert np.allclose(PR.preprocess_signal(x), g7["y"], rtol=0, atol=1e-12)
    assert np.allclose(PR.remove_baseline_drift(x), g7["baseline_removed"], rtol=0, atol=1e-12)
    b, a = butter_lowpass(5, 0.1)
    kat_b = [5.97957804e-05, 2.98978902e-04, 5.97957804e-04, 5.97957804e-04, 2.98978902e-04, 5.97957804e-05]
    kat_a = [1, -3.98454312, 6.43486709, -5.25361517, 2.16513291, -0.35992825]
    assert np.allclose(b, kat_b, rtol=1e-8) and np.allclose(a, kat_a, rtol=1e-8)
    bs, as_ = butter(5, 0.1)
    assert np.allclose(b, bs, rtol=1e-13, atol=0) and np.allclose(a, as_, rtol=1e-13, atol=0)
    assert np.allclose(lfilter_zi(b, a), sp_zi(bs, as_), rtol=1e-9)
    for order, wn in ((3, 0.2), (7, 0.05), (1, 0.5)):
        b2, a2 = butter_lowpass(order, wn)
        bs2, as2 = butter(order, wn)
        assert np.allclose(b2, bs2, rtol=1e-10) and np.allclose(a2, as2, rtol=1e-10)


def test_image_transform_oracle_matches_pillow_golden(golden_dir):
    """g8 holds Pillow's own BILINEAR outputs; the restatement reproduces them bit for bit, and (Pillow being
    installed wherever the tests run) again live on a shape outside the fixture."""
    import hashlib
    from oracle import image_ref as IR
    g8 = np.load(f"{golden_dir}/g8_image.npz")
    for i, (h, w, oh, ow) in enumerate(g8["cases"]):
        img = IR.synthetic_ecg_picture(int(h), int(w), 31 + i)
        assert hashlib.sha256(img.tobytes()).digest() == g8[f"sha_in_{i}"].tobytes()   # formula picture is stable
        assert np.array_equal(IR.resize_bilinear_u8(img, int(oh), int(ow)), g8[f"resized_{i}"])
    lut = IR.to_tensor_normalize(np.arange(256, dtype=np.uint8).reshape(16, 16, 1).repeat(3, 2))[1].reshape(-1)
    assert np.array_equal(lut, g8["normalize_lut"])
    from PIL import Image
    img = IR.synthetic_ecg_picture(91, 333, 5)
    assert np.array_equal(np.asarray(Image.fromarray(img, "RGB").resize((64, 48), Image.BILINEAR)),
                          IR.resize_bilinear_u8(img, 48, 64))


def test_library_resize_tables_match_the_oracle_coefficients():
    """host-side half of the C ABI (no GPU needed): the coefficient table the kernel consumes"""
    import ctypes as C
    from ecgmm.hip import lib as L
    from oracle import image_ref as IR
    lib = L.lib()
    for (h, w, oh, ow) in [(250, 2500, 224, 224), (100, 120, 224, 224), (37, 53, 16, 20)]:
        nb = lib.ecgmm_image_resize_tables_bytes(h, w, oh, ow)
        buf = np.zeros(nb // 4, np.int32)
        assert lib.ecgmm_image_resize_tables(h, w, oh, ow, buf.ctypes.data_as(C.c_void_p), nb) == 0
        hb, hk = IR._coeffs(w, ow)
        vb, vk = IR._coeffs(h, oh)
        want = np.concatenate([hb.reshape(-1), hk.reshape(-1), vb.reshape(-1), vk.reshape(-1)])
        assert np.array_equal(buf, want)
        assert lib.ecgmm_image_resize_tables(h, w, oh, ow, buf.ctypes.data_as(C.c_void_p), nb - 4) != 0


def test_oracle_reproduces_the_reference_composition_golden_g9(golden_dir):
    """g9 holds what the REFERENCE's own ECGMultimodalModel objects (multimodal_paper_modal_balance.py and multimodal.py,
    imported in the build container by oracle/make_golden.py, where bit-identity with the oracle is asserted) return
    for hash-filled weights: the oracle must reproduce them on this machine too."""
    from oracle import tabnet_ref as T
    g9 = np.load(f"{golden_dir}/g9_reference_composition.npz")
    names = ("img_logits", "sig_logits", "clin_logits", "fusion_logits", "var_loss", "soft_w")
    for tag, clin_in, make in (("pmb", 24, lambda: O.ECGMultimodalModel(2, 24)), ("tab", 2, lambda: T.multimodal_tabnet_model(2))):
        m = O.disable_dropout(fill.hash_fill_module(make(), "mm."))
        img, sig, clin, lab = fill.synthetic_batch(8, clin_dim=clin_in, salt=9)
        m.eval()
        with torch.no_grad():
            ev = m(img, sig, clin)
        for n, o in zip(names, ev):
            assert torch.allclose(o, torch.from_numpy(g9[f"{tag}.eval.{n}"]), atol=2e-5), (tag, n)
        m.train()
        tr = m(img, sig, clin)
        loss = F.cross_entropy(tr[3], lab) + 0.1 * tr[4]
        loss.backward()
        assert abs(loss.item() - float(g9[f"{tag}.train.loss"])) < 2e-5
        for k, p in m.named_parameters():
            key = f"{tag}.grad.{k}"
            if key in g9.files:
                ref = torch.from_numpy(g9[key])
                assert (p.grad - ref).norm() <= 2e-4 * ref.norm() + 1e-7, (tag, k)
