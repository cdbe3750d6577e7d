"""Every BASELINE GPU configuration against the CPU oracle AT ITS REAL SIZE (one training-mode forward + backward):

  cfg 3   full multimodal, batch 256, 3x224x224 + 5000 + 16          (BASELINE.json configs[2]; the benchmarked step)
  cfg 2   ImageOnlyClassifier (fc -> 2), batch 128                    (configs[1], train_image_only.py:92-99)
  cfg 5   12-lead ResNet1D_SE + FocalLoss, batch 512, 12x5000         (configs[4], train_signal_12_af.py:238-275)

fp32 compute path: logits within 1e-3 of the oracle (the north-star bar), loss, and the gradients of first / middle /
last layers.  bf16 compute path (the benchmarked dtype): judged against torch's own CPU bf16 autocast of the same
oracle -- the HIP path may deviate from the fp32 oracle by at most 1.3x what autocast does (+2 %), the yardstick of
tests/test_models_gpu.py, here at full batch size where the BatchNorm reductions run over up to 3.2 M values per channel
and the conv kernels run their big-M tile paths (32-bit offsets, persistent halo tiles, split-K weight gradients).
The oracle runs on the host cores (a few seconds per configuration); weights / inputs come from the integer hash fill."""
import pytest
import torch

from ecgmm.config import Config
from ecgmm.hip import functional as HF
from ecgmm.multimodal_paper_modal_balance import ECGMultimodalModel, ResNet1D_SE
from ecgmm.train_image_only import ImageOnlyClassifier
from oracle import fill, ref_models as O

from .util import DEV, dev, rel_err

pytestmark = pytest.mark.gpu


def _nodrop(m):
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    return m


def _run_ref(make_ref, inputs, loss_of, autocast):
    ref = make_ref()
    with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
        out = ref(*inputs)
    logits = (out[3] if isinstance(out, tuple) else out).float()
    extra = out[4].float() if isinstance(out, tuple) else None
    loss = loss_of(logits, extra, False)
    loss.backward()
    return logits.detach(), float(loss), {k: p.grad.clone() for k, p in ref.named_parameters() if p.grad is not None}


def _run_gpu(make_net, cd, inputs, loss_of):
    net = make_net(cd)
    out = net(*[dev(t) for t in inputs])
    logits = out[3] if isinstance(out, tuple) else out
    loss = loss_of(logits, out[4] if isinstance(out, tuple) else None, True)
    loss.backward()
    torch.cuda.synchronize()
    g = {k: p.grad.detach().cpu() for k, p in net.named_parameters() if p.grad is not None}
    res = logits.detach().cpu(), float(loss), g
    del net, out, loss
    torch.cuda.empty_cache()
    return res


def _check(make_ref, make_net, inputs, loss_of, keys, tol_grad_fp32, bf16_logit_abs, bf16_grad_rel):
    """bf16_logit_abs / bf16_grad_rel: ABSOLUTE ceilings of the bf16 compute path beside the autocast-relative bar --
    max |logit - fp32 oracle logit| over the batch, and the largest relative L2 error of the listed gradients (set at
    about twice what the shipped kernels measure on MI355X; the measured values are printed)."""
    l32, loss32, g32 = _run_ref(make_ref, inputs, loss_of, False)
    # ---- fp32 compute path vs the fp32 oracle
    lg, lossg, gg = _run_gpu(make_net, "fp32", inputs, loss_of)
    assert (lg - l32).abs().max() < 1e-3, float((lg - l32).abs().max())
    assert abs(lossg - loss32) < 1e-3
    # (two fp32 implementations summing up to 3.2 M products per weight-gradient element in different orders: the
    # stem's gradient differs by ~5e-3 relative at batch 256; the dense layers by ~3e-7)
    errs = {k: round(rel_err(gg[k], g32[k]), 7) for k in keys}
    print("fp32 gradient rel. errors:", errs)
    assert all(e < tol_grad_fp32 for e in errs.values()), sorted(errs.items(), key=lambda kv: -kv[1])[:4]
    # ---- bf16 compute path vs the same oracle, with torch's CPU bf16 autocast as the yardstick
    l16, _, g16 = _run_ref(make_ref, inputs, loss_of, True)
    lb, lossb, gb = _run_gpu(make_net, "bf16", inputs, loss_of)
    print("bf16 logits rel. error: HIP %.4f, torch autocast %.4f" % (rel_err(lb, l32), rel_err(l16, l32)))
    print("bf16 gradient rel. errors (HIP, torch autocast):", {k: (round(rel_err(gb[k], g32[k]), 4), round(rel_err(g16[k], g32[k]), 4)) for k in keys})
    abs_hip, abs_ac = float((lb - l32).abs().max()), float((l16 - l32).abs().max())
    worst = max(rel_err(gb[k], g32[k]) for k in keys)
    print("bf16 ABSOLUTE errors: max |logit - oracle| HIP %.5f (ceiling %.3g; torch autocast %.5f), |loss - oracle| %.5f, "
          "worst gradient rel. error %.4f (ceiling %.3g)" % (abs_hip, bf16_logit_abs, abs_ac, abs(lossb - loss32), worst,
                                                            bf16_grad_rel))
    assert abs_hip < bf16_logit_abs, abs_hip
    assert worst < bf16_grad_rel, worst
    assert rel_err(lb, l32) < 1.3 * rel_err(l16, l32) + 0.02
    assert abs(lossb - loss32) < 0.05 * max(1.0, abs(loss32))
    bad = {k: (rel_err(gb[k], g32[k]), rel_err(g16[k], g32[k])) for k in keys
           if not rel_err(gb[k], g32[k]) < 1.3 * rel_err(g16[k], g32[k]) + 0.02}
    assert not bad, bad


def test_cfg3_full_multimodal_batch256_vs_oracle():
    B = 256
    sd = fill.hash_fill_module(O.ECGMultimodalModel(2, 16), "mm.").state_dict()
    img, sig, clin, lab = fill.synthetic_batch(B, salt=17)

    def make_ref():
        m = O.ECGMultimodalModel(2, 16); m.load_state_dict(sd); return O.disable_dropout(m).train()

    def make_net(cd):
        cfg = type("Full", (Config,), {"compute_dtype": cd, "clinical_input_dim": 16})
        m = ECGMultimodalModel(cfg); m.load_state_dict(sd); return _nodrop(m).to(DEV).train()

    def loss_of(logits, var, gpu):          # train.py:69-78
        if gpu:
            return HF.cross_entropy_plus(logits, dev(lab), var, 0.1)
        return torch.nn.functional.cross_entropy(logits, lab) + 0.1 * var

    keys = ["image_encoder.conv1.weight", "image_encoder.layer1.0.conv1.weight", "image_encoder.layer2.0.downsample.0.weight",
            "image_encoder.layer4.1.conv2.weight", "image_encoder.fc.weight", "signal_encoder.initial.0.weight",
            "signal_encoder.layer3.conv2.weight", "clinical_encoder.0.weight", "fusion_classifier.0.weight",
            "attention_fusion.weights"]
    _check(make_ref, make_net, (img, sig, clin), loss_of, keys, 1e-2, 3.5e-2, 0.6)     # measured 0.0156 / 0.42 (torch autocast: 0.0172)


def test_cfg2_image_only_batch128_vs_oracle():
    B = 128
    sd = fill.hash_fill_module(O.ImageOnlyClassifier(2), "io.").state_dict()    # train_image_only.py:92-99: fc -> Linear(512, 2)
    img = fill.synthetic_batch(B, salt=23)[0]
    lab = torch.arange(B) % 2

    def make_ref():
        m = O.ImageOnlyClassifier(2); m.load_state_dict(sd); return m.train()

    def make_net(cd):
        m = ImageOnlyClassifier(compute_dtype=cd); m.load_state_dict(sd); return m.to(DEV).train()

    def loss_of(logits, _var, gpu):
        return HF.cross_entropy(logits, dev(lab)) if gpu else torch.nn.functional.cross_entropy(logits, lab)

    keys = ["image_encoder.conv1.weight", "image_encoder.layer1.0.conv1.weight", "image_encoder.layer3.0.conv1.weight",
            "image_encoder.layer4.1.conv2.weight", "image_encoder.fc.weight", "image_encoder.fc.bias"]
    _check(make_ref, make_net, (img,), loss_of, keys, 1e-2, 4.5e-2, 0.6)              # measured 0.0229 / 0.44 (torch autocast: 0.0233)


def test_cfg5_signal12_batch512_focal_vs_oracle():
    B = 512
    sd = fill.hash_fill_module(O.ResNet1D_SE(12, 2), "s12.").state_dict()
    x = fill.hash_tensor((B, 12, 5000), 29, 1.0)
    lab = (torch.arange(B) * 7 // 3) % 2

    def make_ref():
        m = O.ResNet1D_SE(12, 2); m.load_state_dict(sd); return O.disable_dropout(m).train()

    def make_net(cd):
        m = ResNet1D_SE(12, 2, compute_dtype=cd); m.load_state_dict(sd); return _nodrop(m).to(DEV).train()

    focal = O.FocalLoss()

    def loss_of(logits, _var, gpu):         # train_signal_12_af.py:248: FocalLoss(alpha=1, gamma=2)
        return HF.focal_loss(logits, dev(lab), 1.0, 2.0) if gpu else focal(logits, lab)

    keys = ["initial.0.weight", "layer1.conv1.weight", "layer2.downsample.0.weight", "layer3.conv2.weight",
            "layer3.se.fc.0.weight", "classifier.1.weight", "classifier.4.weight"]
    _check(make_ref, make_net, (x,), loss_of, keys, 1e-2, 4e-3, 0.3)                  # measured 0.0012 / 0.15 (torch autocast: 0.0053)
