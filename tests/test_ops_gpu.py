"""Per-op parity on the MI355X: each HIP entry point vs the same torch-CPU fp32 op the reference runs.
fp32 instantiations use exact-f32 MFMA -> tight tolerances; bf16 is checked against a bf16-rounded
oracle with the tolerance stated in each test."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from ecgmm.hip import lib as L
from ecgmm.hip.functional import ptr, stream
from oracle import fill

from .util import DEV, TDT, bf16_round, conv_desc, dev, from_nhwc, pack_weight, rel_err, to_nhwc

pytestmark = pytest.mark.gpu

CONV_CASES = [
    # N, H, W, Cin, Cout, R, S, stride, ph, pw
    (2, 14, 14, 64, 64, 3, 3, 1, 1, 1),      # resnet layer1-style
    (3, 15, 13, 64, 128, 3, 3, 2, 1, 1),     # stride-2, odd sizes
    (2, 9, 9, 128, 256, 1, 1, 2, 0, 0),      # downsample 1x1 / 2
    (1, 7, 7, 512, 512, 3, 3, 1, 1, 1),      # layer4 (K = 4608)
    (5, 1, 313, 128, 256, 1, 3, 2, 0, 1),    # Conv1d k3 s2 (H = 1), ragged L
    (4, 1, 157, 64, 128, 1, 1, 2, 0, 0),     # Conv1d k1 s2
    (7, 1, 1, 96, 40, 1, 1, 1, 0, 0),        # Linear-shaped, Cout not a tile multiple, Cin partial stage
    # the weights-resident halo kernel (bf16, Cin/Cout <= 64, stride 1): tiles spanning images, 1-D, partial channels,
    # and more tiles than workgroups (persistent loop, double-buffered halo)
    (3, 9, 11, 64, 64, 3, 3, 1, 1, 1),
    (6, 1, 300, 64, 64, 1, 3, 1, 0, 1),
    (2, 20, 17, 32, 48, 3, 3, 1, 1, 1),
    (16, 48, 48, 64, 64, 3, 3, 1, 1, 1),
    # conv_halo.hip (bf16, stride-1 3x3 / 1x3, whole 256-pixel tiles): one tile with 16 images inside, one slice / several
    # slices, both channel-tile widths (forward and dgrad swap Cin / Cout), layer-2 geometry (W = 28), layer-4 geometry
    (16, 4, 4, 64, 64, 3, 3, 1, 1, 1),
    (4, 8, 8, 64, 128, 3, 3, 1, 1, 1),
    (8, 16, 16, 128, 256, 3, 3, 1, 1, 1),
    (16, 28, 28, 128, 128, 3, 3, 1, 1, 1),
    (256, 7, 7, 192, 128, 3, 3, 1, 1, 1),
    (8, 1, 160, 64, 64, 1, 3, 1, 0, 1),
    (4, 1, 320, 128, 256, 1, 3, 1, 0, 1),
]


def _conv_ref(x, w, b, stride, ph, pw):
    return F.conv2d(x, w, b, stride=stride, padding=(ph, pw))


@pytest.fixture
def halo_everywhere():
    """conv_halo.hip / wgrad_ring_kernel for every shape they can serve (the defaults only pick them where faster)"""
    L.lib().ecgmm_conv_halo_enable(2)
    L.lib().ecgmm_conv_wgrad_ring_enable(2)
    L.lib().ecgmm_bn_fuse_min_pixels(0)
    yield
    L.lib().ecgmm_bn_fuse_min_pixels(-1)
    L.lib().ecgmm_conv_halo_enable(1)
    L.lib().ecgmm_conv_wgrad_ring_enable(1)


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("dt", [L.F32, L.BF16])
def test_conv_fwd_dgrad_wgrad(case, dt, halo_everywhere):
    N, H, W, Cin, Cout, R, S, st, ph, pw = case
    lib = L.lib()
    x = fill.hash_tensor((N, Cin, H, W), 11)
    w = fill.hash_tensor((Cout, Cin, R, S), 12, (2.0 / (Cin * R * S)) ** 0.5 * 1.7)
    b = fill.hash_tensor((Cout,), 13, 0.5)
    if dt == L.BF16:
        x, w = bf16_round(x), bf16_round(w)
    x.requires_grad_(True)
    w.requires_grad_(True)
    y_ref = _conv_ref(x, w, b, st, ph, pw)
    dy = fill.hash_tensor(tuple(y_ref.shape), 14)
    if dt == L.BF16:
        dy = bf16_round(dy)
    y_ref.backward(dy)
    OH, OW = y_ref.shape[2], y_ref.shape[3]

    d = conv_desc(N, H, W, Cin, Cout, R, S, st, ph, pw)
    xg, dyg = to_nhwc(x.detach(), dt), to_nhwc(dy, dt)
    wf, wd = pack_weight(w.detach(), dt)
    bg = dev(b)
    M = N * OH * OW
    rows = lib.ecgmm_conv_stats_rows(M)
    stats = torch.zeros(rows, 2, Cout, device=DEV)
    yg = torch.empty(M * Cout, device=DEV, dtype=TDT[dt])
    L.check(lib.ecgmm_conv_fwd(dt, C.byref(d), ptr(xg), ptr(wf), ptr(bg), ptr(yg), ptr(stats), 0, stream()))
    y = from_nhwc(yg, dt, y_ref.shape)
    tol = 2e-5 if dt == L.F32 else 6e-3   # bf16: output rounding 2^-9 relative
    assert rel_err(y, y_ref.detach()) < tol
    # fused BatchNorm partial sums (computed from the fp32 accumulators)
    s = stats.sum(0).cpu()
    ref1 = y_ref.detach().sum(dim=(0, 2, 3))
    ref2 = (y_ref.detach() ** 2).sum(dim=(0, 2, 3))
    assert torch.allclose(s[0], ref1, rtol=1e-4, atol=1e-3 * M ** 0.5)
    assert torch.allclose(s[1], ref2, rtol=1e-4, atol=1e-3)

    # dgrad (+ fused addend)
    add = fill.hash_tensor((N, Cin, H, W), 15)
    if dt == L.BF16:
        add = bf16_round(add)
    addg = to_nhwc(add, dt)
    dxg = torch.empty(N * H * W * Cin, device=DEV, dtype=TDT[dt])
    L.check(lib.ecgmm_conv_bwd_data(dt, C.byref(d), ptr(dyg), ptr(wd), ptr(addg), ptr(dxg), stream()))
    dx = from_nhwc(dxg, dt, x.shape)
    assert rel_err(dx, x.grad + add) < tol

    # wgrad
    nb = lib.ecgmm_conv_bwd_weight_workspace(dt, C.byref(d))
    ws = torch.empty(nb, device=DEV, dtype=torch.uint8)
    dw = torch.full(w.shape, 7.0, device=DEV)
    L.check(lib.ecgmm_conv_bwd_weight(dt, C.byref(d), ptr(xg), ptr(dyg), ptr(dw), 0, ptr(ws), nb, stream()))
    torch.cuda.synchronize()
    assert rel_err(dw.cpu(), w.grad) < (2e-5 if dt == L.F32 else 1e-5)  # inputs exact in bf16, fp32 accumulate
    L.check(lib.ecgmm_conv_bwd_weight(dt, C.byref(d), ptr(xg), ptr(dyg), ptr(dw), 1, ptr(ws), nb, stream()))
    torch.cuda.synchronize()
    assert rel_err(dw.cpu(), 2 * w.grad) < 3e-5


@pytest.mark.parametrize("dt", [L.F32, L.BF16])
@pytest.mark.parametrize("case", [(4, 16, 16, 64, 128, 3), (3, 14, 10, 128, 256, 3), (2, 9, 7, 64, 64, 3),
                                  (5, 1, 62, 64, 128, 1), (3, 1, 45, 128, 256, 1)])
def test_stride2_dgrad_with_folded_downsample_branch(case, dt):
    """dx of a stage-entry block's two stride-2 branches in one launch == torch autograd of conv3x3/s2(x) and conv1x1/s2(x)
    (torchvision BasicBlock.downsample; BasicBlock1D, multimodal_paper_modal_balance.py:71-93)"""
    N, H, W, Cin, Cout, R = case
    lib = L.lib()
    ph = 1 if R == 3 else 0
    x = fill.hash_tensor((N, Cin, H, W), 61)
    w = fill.hash_tensor((Cout, Cin, R, 3), 62, (2.0 / (Cin * R * 3)) ** 0.5)
    wd_ = fill.hash_tensor((Cout, Cin, 1, 1), 63, (2.0 / Cin) ** 0.5)
    if dt == L.BF16:
        x, w, wd_ = bf16_round(x), bf16_round(w), bf16_round(wd_)
    x.requires_grad_(True)
    y1 = F.conv2d(x, w, None, 2, (ph, 1))
    y2 = F.conv2d(x, wd_, None, 2, 0)
    assert y1.shape == y2.shape
    dy1, dy2 = fill.hash_tensor(tuple(y1.shape), 64), fill.hash_tensor(tuple(y1.shape), 65)
    if dt == L.BF16:
        dy1, dy2 = bf16_round(dy1), bf16_round(dy2)
    ((y1 * dy1).sum() + (y2 * dy2).sum()).backward()
    d = conv_desc(N, H, W, Cin, Cout, R, 3, 2, ph, 1)
    _, wpk = pack_weight(w.detach(), dt)
    _, wpk2 = pack_weight(wd_.detach(), dt)
    dxg = torch.empty(N * H * W * Cin, device=DEV, dtype=TDT[dt])
    g1, g2 = to_nhwc(dy1, dt), to_nhwc(dy2, dt)
    L.check(lib.ecgmm_conv_bwd_data_with_downsample(dt, C.byref(d), ptr(g1), ptr(wpk), ptr(g2), ptr(wpk2), ptr(dxg), None,
                                                     stream()))
    dx = from_nhwc(dxg, dt, x.shape)
    assert rel_err(dx, x.grad) < (2e-5 if dt == L.F32 else 6e-3)


def test_conv_halo_kernel_agrees_with_the_general_kernel():
    """same launch through both bf16 kernels (ecgmm_conv_halo_enable): identical products, other fp32 summation order"""
    lib = L.lib()
    N, H, W, Cin, Cout = 8, 16, 16, 128, 128
    d = conv_desc(N, H, W, Cin, Cout, 3, 3, 1, 1, 1)
    x = to_nhwc(bf16_round(fill.hash_tensor((N, Cin, H, W), 41)), L.BF16)
    wf, wd = pack_weight(bf16_round(fill.hash_tensor((Cout, Cin, 3, 3), 42, 0.05)), L.BF16)
    M = N * H * W
    outs = []
    try:
        for on in (2, 0):
            lib.ecgmm_conv_halo_enable(on)
            y = torch.empty(M * Cout, device=DEV, dtype=torch.bfloat16)
            dx = torch.empty(M * Cin, device=DEV, dtype=torch.bfloat16)
            st = torch.zeros(lib.ecgmm_conv_stats_rows(M), 2, Cout, device=DEV)
            L.check(lib.ecgmm_conv_fwd(L.BF16, C.byref(d), ptr(x), ptr(wf), None, ptr(y), ptr(st), 1, stream()))
            L.check(lib.ecgmm_conv_bwd_data(L.BF16, C.byref(d), ptr(y), ptr(wd), ptr(x), ptr(dx), stream()))
            torch.cuda.synchronize()
            outs.append((y.float().cpu(), dx.float().cpu(), st.sum(0).cpu()))
    finally:
        lib.ecgmm_conv_halo_enable(1)
    for a, b in zip(*outs):
        assert rel_err(a, b) < 3e-3          # bf16 output rounding of differently ordered fp32 sums
    assert (outs[0][0] >= 0).all()           # act = ReLU in the epilogue


@pytest.mark.parametrize("sep_mask", [False, True])
@pytest.mark.parametrize("case", [(8, 16, 16, 128, 128), (16, 8, 8, 256, 64), (4, 8, 8, 64, 128)])
def test_dgrad_with_fused_batchnorm_backward_reduction(case, sep_mask):
    """conv dgrad whose epilogue also reduces for the BatchNorm backward that consumes the gradient
    (ecgmm_conv_bwd_data_bnred + ecgmm_bn_bwd_from_rows) == plain dgrad + ecgmm_bn_bwd == torch autograd of
    relu(bn(y) [+ res]) -> conv.  sep_mask: the ReLU followed a residual add, so its mask comes from the block output
    (BasicBlock tail) and the stored gradient is the masked one."""
    N, H, W, Cin, Cout = case
    lib = L.lib()
    dt = L.BF16
    M = N * H * W
    d = conv_desc(N, H, W, Cin, Cout, 3, 3, 1, 1, 1)
    y = bf16_round(fill.hash_tensor((N, Cin, H, W), 51, 2.0) + 0.3)          # raw conv output that BN normalises
    res = bf16_round(fill.hash_tensor((N, Cin, H, W), 52))
    w = bf16_round(fill.hash_tensor((Cout, Cin, 3, 3), 53, 0.05))
    dy = bf16_round(fill.hash_tensor((N, Cout, H, W), 54))
    gam, bet = 1 + 0.2 * fill.hash_tensor((Cin,), 55), 0.1 * fill.hash_tensor((Cin,), 56)
    yg, dyg = to_nhwc(y, dt), to_nhwc(dy, dt)
    _, wd = pack_weight(w, dt)
    # forward coefficients + activated tensor on the GPU (bn_finalize + bn_act), exactly what the plans feed the backward
    rows = lib.ecgmm_col_stats_rows(dt, M, Cin)
    part = torch.zeros(rows + 64, 2, Cin, device=DEV)
    L.check(lib.ecgmm_col_stats(dt, ptr(yg), M, Cin, ptr(part), stream()))
    coef = torch.empty(4, Cin, device=DEV)
    gg, bg = dev(gam), dev(bet)
    L.check(lib.ecgmm_bn_finalize(ptr(part), rows, Cin, float(M), ptr(gg), ptr(bg), None, None, None, 0.1, 1e-5, ptr(coef), stream()))
    resg = to_nhwc(res, dt) if sep_mask else None
    outg = torch.empty(M * Cin, device=DEV, dtype=torch.bfloat16)
    L.check(lib.ecgmm_bn_act(dt, ptr(yg), ptr(coef), ptr(resg), None, None, 1, 1, ptr(outg), M, Cin, stream()))
    mask = outg if sep_mask else yg
    scratch = torch.empty(lib.ecgmm_bn_bwd_scratch(dt, M, Cin), device=DEV, dtype=torch.uint8)

    def run(fused):
        dx = torch.empty(M * Cin, device=DEV, dtype=torch.bfloat16)
        dyo = torch.empty(M * Cin, device=DEV, dtype=torch.bfloat16)
        dgam, dbet = torch.empty(Cin, device=DEV), torch.empty(Cin, device=DEV)
        if fused:
            rows_buf = torch.full((512, 2, Cin), float("nan"), device=DEV)
            n = C.c_int(0)
            L.check(lib.ecgmm_conv_bwd_data_bnred(dt, C.byref(d), ptr(dyg), ptr(wd), None, ptr(dx), ptr(yg), ptr(mask), ptr(coef),
                                                  ptr(rows_buf), C.byref(n), stream()))
            assert 1 <= n.value <= 512
            L.check(lib.ecgmm_bn_bwd_from_rows(dt, ptr(dx), None if sep_mask else ptr(yg), ptr(yg), ptr(coef), ptr(gg), ptr(dgam),
                                               ptr(dbet), ptr(dyo), ptr(rows_buf), n.value, M, Cin, ptr(scratch), stream()))
        else:
            dz = torch.empty(M * Cin, device=DEV, dtype=torch.bfloat16)
            L.check(lib.ecgmm_conv_bwd_data(dt, C.byref(d), ptr(dyg), ptr(wd), None, ptr(dx), stream()))
            L.check(lib.ecgmm_bn_bwd(dt, ptr(dx), ptr(mask), None, None, 1, ptr(yg), ptr(coef), ptr(gg), ptr(dgam), ptr(dbet),
                                     ptr(dyo), ptr(dz) if sep_mask else None, None, M, Cin, ptr(scratch), stream()))
        torch.cuda.synchronize()
        return from_nhwc(dyo, dt, y.shape), dgam.cpu(), dbet.cpu()

    try:
        lib.ecgmm_conv_halo_enable(2)
        a = run(True)
        b = run(False)
    finally:
        lib.ecgmm_conv_halo_enable(1)
    for u, v in zip(a, b):
        assert rel_err(u, v) < 2e-3
    # and against torch autograd (bf16 storage of the conv gradient is part of both GPU paths: loose tolerance)
    yr, gr, br = y.clone().requires_grad_(True), gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    z = F.batch_norm(yr, None, None, gr, br, True, 0.1, 1e-5)
    act = F.relu(z + res) if sep_mask else F.relu(z)
    F.conv2d(act, w, None, stride=1, padding=1).backward(dy)
    assert rel_err(a[0], yr.grad) < 2e-2 and rel_err(a[1], gr.grad) < 1e-2 and rel_err(a[2], br.grad) < 1e-2


@pytest.mark.parametrize("cap", [0, 64, 7])
def test_fused_reduction_row_count_follows_the_halo_cu_cap(cap):
    """with the persistent conv kernel capped to fewer CUs (ECGMM_HALO_CUS / ecgmm_conv_halo_cus) a launch writes
    fewer partial rows of the fused BatchNorm-backward reduction.  The count a plan uses when the rows are consumed by a
    LATER call (csrc/plan_resnet18.hip: bn2's reduction rides on the next block's dgrad) is
    ecgmm_conv_bwd_data_bnred_rows: it must equal what the launch wrote, rows past it stay untouched, and the finished
    backward must not depend on the cap beyond fp32 summation order."""
    N, H, W, Cin, Cout = 32, 16, 16, 64, 64        # 32 pixel tiles: more tiles than 7 workgroups, fewer than 64
    lib = L.lib()
    dt = L.BF16
    M = N * H * W
    d = conv_desc(N, H, W, Cin, Cout, 3, 3, 1, 1, 1)
    y = bf16_round(fill.hash_tensor((N, Cin, H, W), 61, 2.0) + 0.3)
    w = bf16_round(fill.hash_tensor((Cout, Cin, 3, 3), 63, 0.05))
    dy = bf16_round(fill.hash_tensor((N, Cout, H, W), 64))
    gam, bet = dev(1 + 0.2 * fill.hash_tensor((Cin,), 65)), dev(0.1 * fill.hash_tensor((Cin,), 66))
    yg, dyg = to_nhwc(y, dt), to_nhwc(dy, dt)
    _, wd = pack_weight(w, dt)
    rows = lib.ecgmm_col_stats_rows(dt, M, Cin)
    part = torch.zeros(rows + 64, 2, Cin, device=DEV)
    L.check(lib.ecgmm_col_stats(dt, ptr(yg), M, Cin, ptr(part), stream()))
    coef = torch.empty(4, Cin, device=DEV)
    L.check(lib.ecgmm_bn_finalize(ptr(part), rows, Cin, float(M), ptr(gam), ptr(bet), None, None, None, 0.1, 1e-5, ptr(coef), stream()))
    scratch = torch.empty(lib.ecgmm_bn_bwd_scratch(dt, M, Cin), device=DEV, dtype=torch.uint8)

    def run():
        dx = torch.empty(M * Cin, device=DEV, dtype=torch.bfloat16)
        dyo = torch.empty(M * Cin, device=DEV, dtype=torch.bfloat16)
        dgam, dbet = torch.empty(Cin, device=DEV), torch.empty(Cin, device=DEV)
        rows_buf = torch.full((512, 2, Cin), float("nan"), device=DEV)
        n = C.c_int(0)
        want = lib.ecgmm_conv_bwd_data_bnred_rows(dt, C.byref(d))
        L.check(lib.ecgmm_conv_bwd_data_bnred(dt, C.byref(d), ptr(dyg), ptr(wd), None, ptr(dx), ptr(yg), ptr(yg), ptr(coef),
                                              ptr(rows_buf), C.byref(n), stream()))
        torch.cuda.synchronize()
        assert n.value == want >= 1, (n.value, want)
        assert torch.isfinite(rows_buf[:n.value]).all() and torch.isnan(rows_buf[n.value:]).all()
        L.check(lib.ecgmm_bn_bwd_from_rows(dt, ptr(dx), ptr(yg), ptr(yg), ptr(coef), ptr(gam), ptr(dgam), ptr(dbet), ptr(dyo),
                                           ptr(rows_buf), want, M, Cin, ptr(scratch), stream()))
        torch.cuda.synchronize()
        return n.value, dx.clone(), dyo.float().cpu(), dgam.cpu(), dbet.cpu()

    try:
        lib.ecgmm_conv_halo_enable(2)
        lib.ecgmm_conv_halo_cus(0)
        full = run()
        lib.ecgmm_conv_halo_cus(cap)
        capped = run()
    finally:
        lib.ecgmm_conv_halo_cus(0)
        lib.ecgmm_conv_halo_enable(1)
    assert capped[0] <= full[0] and (cap == 0 or capped[0] <= 2 * cap)   # (64-channel tiles: up to two workgroups per CU)
    assert torch.equal(capped[1], full[1])                                  # the gradient itself: pure scheduling
    for u, v in zip(capped[2:], full[2:]):
        assert rel_err(u, v) < 1e-5


@pytest.mark.parametrize("dt", [L.F32, L.BF16])
@pytest.mark.parametrize("shape", [(4, 64, 16, 16, 256), (2, 128, 8, 8, 37), (3, 256, 5, 4, 512), (2, 512, 4, 4, 64)])
def test_finalize_folded_into_the_consumer_pass(dt, shape):
    """ecgmm_bn_act_from_rows == ecgmm_bn_finalize + ecgmm_bn_act, and ecgmm_bn_bwd with the fold (default) == without
    (ecgmm_bn_fold(0)): outputs, coefficients, running statistics, dgamma / dbeta.  Every workgroup of the consumer folds the
    partial rows itself (csrc/elementwise.hip); the rows here are synthetic multi-row splits of the true column sums."""
    N, Cn, H, W, rows = shape
    lib = L.lib()
    M = N * H * W
    x = fill.hash_tensor((N, Cn, H, W), 71, 1.5) + 0.2
    res = fill.hash_tensor((N, Cn, H, W), 72)
    if dt == L.BF16:
        x, res = bf16_round(x), bf16_round(res)
    gam, bet = dev(1 + 0.3 * fill.hash_tensor((Cn,), 73)), dev(0.2 * fill.hash_tensor((Cn,), 74))
    xg, rg = to_nhwc(x, dt), to_nhwc(res, dt)
    # partial rows: the column sums split over `rows` rows with hash weights that add up to 1
    wgt = (fill.hash_tensor((rows, 1), 75).abs() + 0.1)
    wgt = dev(wgt / wgt.sum())
    col = torch.stack([x.sum(dim=(0, 2, 3)), (x * x).sum(dim=(0, 2, 3))]).to(DEV)            # [2][C]
    part = (wgt[:, :, None] * col[None]).contiguous()                                            # [rows][2][C]
    part = torch.cat([part, torch.zeros(64, 2, Cn, device=DEV)])

    lib.ecgmm_bn_fold(1)       # (ecgmm_bn_act_from_rows falls back to two launches when the fold is switched off)

    def fwd(folded):
        rm, rv, nbt = torch.zeros(Cn, device=DEV), torch.ones(Cn, device=DEV), torch.zeros((), dtype=torch.int64, device=DEV)
        coef = torch.full((4, Cn), float("nan"), device=DEV)
        out = torch.empty(M * Cn, device=DEV, dtype=TDT[dt])
        if folded:
            L.check(lib.ecgmm_bn_act_from_rows(dt, ptr(xg), ptr(part), rows, float(M), ptr(gam), ptr(bet), ptr(rm), ptr(rv),
                                               ptr(nbt), 0.1, 1e-5, ptr(coef), ptr(rg), None, None, 1, 1, ptr(out), M, Cn, stream()))
        else:
            L.check(lib.ecgmm_bn_finalize(ptr(part), rows, Cn, float(M), ptr(gam), ptr(bet), ptr(rm), ptr(rv), ptr(nbt), 0.1,
                                          1e-5, ptr(coef), stream()))
            L.check(lib.ecgmm_bn_act(dt, ptr(xg), ptr(coef), ptr(rg), None, None, 1, 1, ptr(out), M, Cn, stream()))
        torch.cuda.synchronize()
        return out.float().cpu(), coef.cpu(), rm.cpu(), rv.cpu(), int(nbt)

    a, b = fwd(True), fwd(False)
    assert a[4] == b[4] == 1
    for u, v in zip(a[1:4], b[1:4]):
        assert torch.allclose(u, v, rtol=1e-6, atol=1e-7), float((u - v).abs().max())
    assert rel_err(a[0], b[0]) < (1e-6 if dt == L.F32 else 2e-3)
    # backward: the same call with and without the fold
    dy = fill.hash_tensor((N, Cn, H, W), 76)
    if dt == L.BF16:
        dy = bf16_round(dy)
    dyg = to_nhwc(dy, dt)
    coef = dev(b[1])
    scratch = torch.empty(lib.ecgmm_bn_bwd_scratch(dt, M, Cn), device=DEV, dtype=torch.uint8)

    def bwd(fold):
        lib.ecgmm_bn_fold(fold)
        dx = torch.empty(M * Cn, device=DEV, dtype=TDT[dt])
        dgam, dbet = torch.full((Cn,), float("nan"), device=DEV), torch.full((Cn,), float("nan"), device=DEV)
        L.check(lib.ecgmm_bn_bwd(dt, ptr(dyg), ptr(xg), None, None, 1, ptr(xg), ptr(coef), ptr(gam), ptr(dgam), ptr(dbet),
                                 ptr(dx), None, None, M, Cn, ptr(scratch), stream()))
        torch.cuda.synchronize()
        return dx.float().cpu(), dgam.cpu(), dbet.cpu()

    try:
        c, d = bwd(1), bwd(0)
    finally:
        lib.ecgmm_bn_fold(0)
    assert torch.equal(c[1], d[1]) or torch.allclose(c[1], d[1], rtol=1e-6, atol=1e-6)
    assert torch.allclose(c[2], d[2], rtol=1e-6, atol=1e-6)
    assert rel_err(c[0], d[0]) < (1e-6 if dt == L.F32 else 2e-3)


@pytest.mark.parametrize("case", [(64, 16, 16, 64, 64), (300, 8, 8, 64, 64), (16, 16, 16, 128, 128), (8, 1, 160, 128, 128)])
def test_conv_forward_statistics_one_row_per_workgroup(case):
    """ecgmm_conv_fwd_wgrows (the plans' form: partial sums accumulated per workgroup -- in registers across its tiles for the
    64-channel tiles, in LDS for the 128-channel ones) == ecgmm_conv_fwd: identical output bits, same column sums."""
    N, H, W, Cin, Cout = case
    lib = L.lib()
    dt = L.BF16
    R = 3 if H > 1 else 1
    d = conv_desc(N, H, W, Cin, Cout, R, 3, 1, R // 2, 1)
    M = N * H * W
    x = to_nhwc(bf16_round(fill.hash_tensor((N, Cin, H, W), 81)), dt)
    wf, _ = pack_weight(bf16_round(fill.hash_tensor((Cout, Cin, R, 3), 82, 0.05)), dt)
    rows = lib.ecgmm_conv_stats_rows(M)
    try:
        lib.ecgmm_conv_halo_enable(2)
        y0 = torch.empty(M * Cout, device=DEV, dtype=torch.bfloat16)
        s0 = torch.zeros(rows + 64, 2, Cout, device=DEV)
        L.check(lib.ecgmm_conv_fwd(dt, C.byref(d), ptr(x), ptr(wf), None, ptr(y0), ptr(s0), 0, stream()))
        y1 = torch.empty_like(y0)
        s1 = torch.full((rows + 64, 2, Cout), float("nan"), device=DEV)
        n = C.c_int(0)
        L.check(lib.ecgmm_conv_fwd_wgrows(dt, C.byref(d), ptr(x), ptr(wf), None, ptr(y1), ptr(s1), C.byref(n), 0, stream()))
        torch.cuda.synchronize()
    finally:
        lib.ecgmm_conv_halo_enable(1)
    assert 1 <= n.value <= 512 and torch.isfinite(s1[:n.value]).all()
    assert torch.equal(y0.view(torch.int16), y1.view(torch.int16))
    a, b = s0[:rows].sum(0), s1[:n.value].sum(0)
    assert torch.allclose(a, b, rtol=2e-5, atol=1e-5 * float(a.abs().max()) + 1e-3), float((a - b).abs().max())


def test_conv_rejects_bad_shapes():
    lib = L.lib()
    d = conv_desc(1, 8, 8, 6, 64, 3, 3, 1, 1, 1)   # Cin not a multiple of the 16-byte vector
    t = torch.zeros(16, device=DEV)
    rc = lib.ecgmm_conv_fwd(L.BF16, C.byref(d), ptr(t), ptr(t), None, ptr(t), None, 0, stream())
    assert rc == 1 and b"multiple" in lib.ecgmm_last_error()
    d = conv_desc(1, 8, 8, 64, 64, 5, 5, 1, 2, 2)
    rc = lib.ecgmm_conv_bwd_weight(L.BF16, C.byref(d), ptr(t), ptr(t), ptr(t), 0, ptr(t), 1 << 30, stream())
    assert rc == 1
    d = conv_desc(1, 8, 8, 64, 64, 3, 3, 1, 1, 1)
    rc = lib.ecgmm_conv_bwd_weight(L.BF16, C.byref(d), ptr(t), ptr(t), ptr(t), 0, ptr(t), 16, stream())
    assert rc == 3 and b"workspace" in lib.ecgmm_last_error()


STEM_CASES = [
    (2, 3, 64, 64, 7),       # 2-D stem
    (1, 3, 50, 83, 7),       # ragged, non-square
    (3, 1, 1, 500, 1),       # 1-D stem, 1 lead
    (2, 12, 1, 333, 1),      # 1-D stem, 12 leads, odd length
]


@pytest.mark.parametrize("case", STEM_CASES)
@pytest.mark.parametrize("dt", [L.F32, L.BF16])
def test_stem_fwd_wgrad(case, dt):
    N, Cin, H, W, R = case
    lib = L.lib()
    x = fill.hash_tensor((N, Cin, H, W), 21)
    w = fill.hash_tensor((64, Cin, R, 7), 22, 0.3)
    b = fill.hash_tensor((64,), 23, 0.2)
    xr, wr = (bf16_round(x), bf16_round(w)) if dt == L.BF16 else (x, w)
    wr = wr.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, b, stride=2, padding=(R // 2, 3))
    dy = fill.hash_tensor(tuple(y_ref.shape), 24)
    if dt == L.BF16:
        dy = bf16_round(dy)
    y_ref.backward(dy)
    OH, OW = y_ref.shape[2], y_ref.shape[3]

    xg, wg, bg = dev(x), dev(w), dev(b)
    pk = torch.empty(lib.ecgmm_stem_packed_elems(Cin, R), device=DEV, dtype=TDT[dt])
    L.check(lib.ecgmm_stem_pack(dt, ptr(wg), ptr(pk), Cin, R, stream()))
    rows = lib.ecgmm_stem_stats_rows(N, Cin, H, W, R)
    stats = torch.zeros(rows, 2, 64, device=DEV)
    yg = torch.empty(N * OH * OW * 64, device=DEV, dtype=TDT[dt])
    L.check(lib.ecgmm_stem_fwd(dt, ptr(xg), ptr(pk), ptr(bg), ptr(yg), ptr(stats), N, Cin, H, W, R, stream()))
    y = from_nhwc(yg, dt, y_ref.shape)
    assert rel_err(y, y_ref.detach()) < (2e-5 if dt == L.F32 else 6e-3)
    s = stats.sum(0).cpu()
    assert torch.allclose(s[0], y_ref.detach().sum(dim=(0, 2, 3)), rtol=1e-4, atol=2e-2)

    if dt == L.BF16:   # per-workgroup statistics rows (what the plans call): same output bits, same column sums
        rows2 = lib.ecgmm_stem_wg_stats_rows(N, Cin, H, W, R)
        stats2 = torch.full((rows2 + 64, 2, 64), float("nan"), device=DEV)
        yg2 = torch.empty_like(yg)
        L.check(lib.ecgmm_stem_fwd_wgrows(dt, ptr(xg), ptr(pk), ptr(bg), ptr(yg2), ptr(stats2), N, Cin, H, W, R, stream()))
        torch.cuda.synchronize()
        assert torch.equal(yg2.view(torch.int16), yg.view(torch.int16))
        s2 = stats2[:rows2].sum(0).cpu()
        assert torch.allclose(s2, s, rtol=2e-5, atol=1e-5 * float(s.abs().max()) + 1e-3)
    dyg = to_nhwc(dy, dt)
    nb = lib.ecgmm_stem_bwd_weight_workspace(N, Cin, H, W, R)
    ws = torch.empty(nb, device=DEV, dtype=torch.uint8)
    dw = torch.zeros(w.shape, device=DEV)
    L.check(lib.ecgmm_stem_bwd_weight(dt, ptr(xg), ptr(dyg), ptr(dw), 0, ptr(ws), nb, N, Cin, H, W, R, stream()))
    torch.cuda.synchronize()
    assert rel_err(dw.cpu(), wr.grad) < (2e-5 if dt == L.F32 else 1e-5)


@pytest.mark.parametrize("case", [(2, 3, 64, 64), (1, 3, 50, 83), (3, 3, 37, 100), (2, 2, 40, 40), (5, 1, 33, 47),
                                  (40, 3, 224, 224)])
def test_stem_by_recompute_matches_the_two_pass_route(case):
    """csrc/conv_stem_fused.hip (bf16): conv7x7/2 -> bn -> relu -> maxpool(3,2,1) and its backward WITHOUT the
    full-resolution conv output, against the route that stores it (ecgmm_stem_fwd + bnrelu_maxpool; pool_bn_bwd +
    stem_bwd_weight -- themselves held to F.conv2d / torch autograd above): the pooled tensor and the arg-max bytes are
    bit-identical (same MFMA order per output, same rounding points), dgamma / dbeta too (same reduction kernel), the conv
    weight gradient up to fp32 summation order.  torchvision resnet18 stem, multimodal_paper_modal_balance.py:210."""
    N, Cin, H, W = case
    dt, R = L.BF16, 7
    lib = L.lib()
    x, w = dev(fill.hash_tensor((N, Cin, H, W), 31)), dev(fill.hash_tensor((64, Cin, R, 7), 32, 0.3))
    gam, bet = dev(1 + 0.3 * fill.hash_tensor((64,), 33)), dev(0.2 * fill.hash_tensor((64,), 34))
    gam[5] = -gam[5]                                     # a negative BatchNorm scale: max of relu(bn(y)), not bn(max y)
    OH, OW = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    PH, PW = (OH - 1) // 2 + 1, (OW - 1) // 2 + 1
    M = N * OH * OW
    pk = torch.empty(lib.ecgmm_stem_packed_elems(Cin, R), device=DEV, dtype=torch.bfloat16)
    L.check(lib.ecgmm_stem_pack(dt, ptr(w), ptr(pk), Cin, R, stream()))
    # ---- two-pass route
    rows = lib.ecgmm_stem_stats_rows(N, Cin, H, W, R)
    stats = torch.zeros(rows + 64, 2, 64, device=DEV)
    y = torch.empty(M * 64, device=DEV, dtype=torch.bfloat16)
    L.check(lib.ecgmm_stem_fwd(dt, ptr(x), ptr(pk), None, ptr(y), ptr(stats), N, Cin, H, W, R, stream()))
    col = stats[:rows].sum(0)
    coef = torch.empty(4, 64, device=DEV)
    rm, rv, nbt = torch.zeros(64, device=DEV), torch.ones(64, device=DEV), torch.zeros((), dtype=torch.int64, device=DEV)
    L.check(lib.ecgmm_bn_finalize(ptr(stats), rows, 64, float(M), ptr(gam), ptr(bet), ptr(rm), ptr(rv), ptr(nbt), 0.1, 1e-5,
                                  ptr(coef), stream()))
    p_ref = torch.empty(N * PH * PW * 64, device=DEV, dtype=torch.bfloat16)
    i_ref = torch.empty(N * PH * PW * 64, device=DEV, dtype=torch.uint8)
    L.check(lib.ecgmm_bnrelu_maxpool(dt, ptr(y), ptr(coef), ptr(p_ref), ptr(i_ref), N, OH, OW, 64, stream()))
    # ---- recompute: statistics pass
    rows2 = lib.ecgmm_stem_stats_only_rows(N, Cin, H, W, R)
    stats2 = torch.full((rows2 + 64, 2, 64), float("nan"), device=DEV)
    L.check(lib.ecgmm_stem_stats_only(dt, ptr(x), ptr(pk), None, ptr(stats2), N, Cin, H, W, R, stream()))
    col2 = stats2[:rows2].sum(0)
    assert torch.isfinite(col2).all()
    assert torch.allclose(col2, col, rtol=2e-5, atol=1e-5 * float(col.abs().max()) + 1e-3), float((col2 - col).abs().max())
    coef2 = torch.empty(4, 64, device=DEV)
    rm2, rv2, nbt2 = torch.zeros(64, device=DEV), torch.ones(64, device=DEV), torch.zeros((), dtype=torch.int64, device=DEV)
    L.check(lib.ecgmm_bn_finalize(ptr(stats2), rows2, 64, float(M), ptr(gam), ptr(bet), ptr(rm2), ptr(rv2), ptr(nbt2), 0.1,
                                  1e-5, ptr(coef2), stream()))
    assert torch.allclose(coef2, coef, rtol=1e-4, atol=1e-5) and torch.allclose(rv2, rv, rtol=1e-4)
    # ---- recompute: pooled forward (the SAME coefficients, so every bit can be compared)
    p_new = torch.full((N * PH * PW * 64,), float("nan"), device=DEV, dtype=torch.bfloat16)
    i_new = torch.full((N * PH * PW * 64,), 77, device=DEV, dtype=torch.uint8)
    L.check(lib.ecgmm_stem_pool_fwd(ptr(x), ptr(pk), ptr(coef), ptr(p_new), ptr(i_new), N, Cin, H, W, stream()))
    torch.cuda.synchronize()
    assert torch.equal(p_new.view(torch.int16), p_ref.view(torch.int16)), int((p_new.view(torch.int16) != p_ref.view(torch.int16)).sum())
    assert torch.equal(i_new, i_ref), int((i_new != i_ref).sum())
    # ---- backward
    dp = to_nhwc(bf16_round(fill.hash_tensor((N, 64, PH, PW), 35)), dt)
    scratch = torch.empty(lib.ecgmm_bn_bwd_scratch(dt, M, 64), device=DEV, dtype=torch.uint8)
    dy = torch.empty_like(y)
    dgam, dbet = torch.zeros(64, device=DEV), torch.zeros(64, device=DEV)
    L.check(lib.ecgmm_pool_bn_bwd(dt, ptr(dp), ptr(p_ref), ptr(i_ref), ptr(y), ptr(coef), ptr(gam), ptr(dgam), ptr(dbet),
                                  ptr(dy), None, N, OH, OW, 64, ptr(scratch), stream()))
    nb = lib.ecgmm_stem_bwd_weight_workspace(N, Cin, H, W, R)
    ws = torch.empty(nb, device=DEV, dtype=torch.uint8)
    dw_ref = torch.zeros_like(w)
    L.check(lib.ecgmm_stem_bwd_weight(dt, ptr(x), ptr(dy), ptr(dw_ref), 0, ptr(ws), nb, N, Cin, H, W, R, stream()))
    nb2 = lib.ecgmm_stem_pool_bwd_workspace(N, Cin, H, W)
    ws2 = torch.empty(nb2, device=DEV, dtype=torch.uint8)
    dw = torch.full_like(w, float("nan"))
    dgam2, dbet2 = torch.zeros(64, device=DEV), torch.zeros(64, device=DEV)
    L.check(lib.ecgmm_stem_pool_bwd(ptr(x), ptr(pk), ptr(coef), ptr(gam), ptr(dp), ptr(p_ref), ptr(i_ref), ptr(dgam2),
                                    ptr(dbet2), ptr(dw), ptr(ws2), nb2, N, Cin, H, W, stream()))
    torch.cuda.synchronize()
    assert torch.equal(dgam2, dgam) and torch.equal(dbet2, dbet)
    err = rel_err(dw.cpu(), dw_ref.cpu())
    print("stem weight gradient, recompute vs two-pass: rel err %.2e, bit-identical %s" % (err, torch.equal(dw, dw_ref)))
    assert err < 1e-5, err
    # frozen conv weight (dw = NULL): the BatchNorm gradients alone
    dgam3, dbet3 = torch.zeros(64, device=DEV), torch.zeros(64, device=DEV)
    L.check(lib.ecgmm_stem_pool_bwd(ptr(x), ptr(pk), ptr(coef), ptr(gam), ptr(dp), ptr(p_ref), ptr(i_ref), ptr(dgam3),
                                    ptr(dbet3), None, ptr(ws2), nb2, N, Cin, H, W, stream()))
    assert torch.equal(dgam3, dgam)
    assert lib.ecgmm_stem_pool_fwd(ptr(x), ptr(pk), ptr(coef), ptr(p_new), ptr(i_new), N, 4, H, W, stream()) != 0   # Cin > 3


@pytest.mark.parametrize("dt", [L.F32, L.BF16])
@pytest.mark.parametrize("shape", [(3, 64, 9, 11), (2, 128, 1, 77), (5, 256, 4, 4)])
def test_batchnorm_train_fwd_bwd_with_residual_and_gate(dt, shape):
    """bn_finalize + bn_act + bn_bwd vs F.batch_norm autograd, incl. running stats, ReLU mask,
    residual add, SE-style gate and additive per-sample term."""
    N, Cn, H, W = shape
    lib = L.lib()
    M, R = N * H * W, H * W
    y = fill.hash_tensor(shape, 31, 2.0) + 0.5
    res = fill.hash_tensor(shape, 32)
    gate = 0.5 + 0.4 * fill.hash_tensor((N, Cn), 33)
    addc = 0.01 * fill.hash_tensor((N, Cn), 34)
    gam = 1 + 0.2 * fill.hash_tensor((Cn,), 35)
    bet = 0.1 * fill.hash_tensor((Cn,), 36)
    dout = fill.hash_tensor(shape, 37)
    if dt == L.BF16:
        y, res, dout = bf16_round(y), bf16_round(res), bf16_round(dout)
    rm, rv = torch.zeros(Cn), torch.ones(Cn)
    yr = y.clone().requires_grad_(True)
    gr, br = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    z = F.batch_norm(yr, rm, rv, gr, br, True, 0.1, 1e-5)
    pre = z * gate[:, :, None, None] + res
    out_ref = F.relu(pre)
    # loss: <dout, out> + <addc, mean_HW z>  -> dz gets the gate, the mask and the additive term
    (out_ref * dout).sum().backward(retain_graph=True)
    g_main = yr.grad.clone(), gr.grad.clone(), br.grad.clone()
    yr.grad = None; gr.grad = None; br.grad = None
    ((out_ref * dout).sum() + (z.mean(dim=(2, 3)) * addc).sum() * R).backward()

    yg, resg, doutg = to_nhwc(y, dt), to_nhwc(res, dt), to_nhwc(dout, dt)
    # partial stats through col_stats (the conv epilogue path is covered in the conv test)
    rows = lib.ecgmm_col_stats_rows(dt, M, Cn)
    partial = torch.empty(rows, 2, Cn, device=DEV)
    L.check(lib.ecgmm_col_stats(dt, ptr(yg), M, Cn, ptr(partial), stream()))
    coef = torch.empty(4, Cn, device=DEV)
    rmg, rvg, nbt = dev(torch.zeros(Cn)), dev(torch.ones(Cn)), torch.zeros((), dtype=torch.int64, device=DEV)
    gg, bg = dev(gam), dev(bet)
    L.check(lib.ecgmm_bn_finalize(ptr(partial), rows, Cn, float(M), ptr(gg), ptr(bg), ptr(rmg), ptr(rvg), ptr(nbt),
                                  0.1, 1e-5, ptr(coef), stream()))
    outg = torch.empty_like(yg)
    gateg, addg = dev(gate), dev(addc)
    L.check(lib.ecgmm_bn_act(dt, ptr(yg), ptr(coef), ptr(resg), None, ptr(gateg), R, 1, ptr(outg), M, Cn, stream()))
    out = from_nhwc(outg, dt, shape)
    tol = 1e-5 if dt == L.F32 else 8e-3
    assert (out - out_ref.detach()).abs().max() < tol * 4
    torch.cuda.synchronize()
    assert torch.allclose(rmg.cpu(), rm, atol=1e-5) and torch.allclose(rvg.cpu(), rv, rtol=1e-4, atol=1e-5)
    assert int(nbt.item()) == 1

    dyg, dzg = torch.empty_like(yg), torch.empty_like(yg)
    dgam, dbet, dbias = torch.zeros(Cn, device=DEV), torch.zeros(Cn, device=DEV), torch.zeros(Cn, device=DEV)
    nb = lib.ecgmm_bn_bwd_scratch(dt, M, Cn)
    scratch = torch.empty(nb, device=DEV, dtype=torch.uint8)
    # use the oracle's post-ReLU output as the mask reference so both sides agree on borderline elements
    maskg = to_nhwc(out_ref.detach(), dt)
    L.check(lib.ecgmm_bn_bwd(dt, ptr(doutg), ptr(maskg), ptr(gateg), ptr(addg), R, ptr(yg), ptr(coef), ptr(gg),
                             ptr(dgam), ptr(dbet), ptr(dyg), ptr(dzg), ptr(dbias), M, Cn, ptr(scratch), stream()))
    dyv = from_nhwc(dyg, dt, shape)
    dzv = from_nhwc(dzg, dt, shape)
    gtol = 2e-4 if dt == L.F32 else 2e-2
    assert rel_err(dyv, yr.grad) < gtol
    assert rel_err(dgam.cpu(), gr.grad) < gtol and rel_err(dbet.cpu(), br.grad) < gtol
    assert rel_err(dzv, dout * (out_ref.detach() > 0)) < (1e-6 if dt == L.F32 else 1e-6)
    assert dbias.abs().max().item() < (1e-3 if dt == L.F32 else 0.5)  # analytically zero


@pytest.mark.parametrize("dt", [L.F32, L.BF16])
@pytest.mark.parametrize("shape", [(2, 64, 12, 10), (3, 64, 1, 41), (1, 128, 7, 7)])
def test_bnrelu_maxpool_fwd_bwd(dt, shape):
    N, Cn, H, W = shape
    lib = L.lib()
    y = fill.hash_tensor(shape, 41, 2.0)
    if dt == L.BF16:
        y = bf16_round(y)
    sc = 1 + 0.3 * fill.hash_tensor((Cn,), 42)
    sh = 0.2 * fill.hash_tensor((Cn,), 43)
    a = F.relu(y * sc[None, :, None, None] + sh[None, :, None, None])
    if dt == L.BF16:
        a = bf16_round(a)
    a.requires_grad_(True)
    if H == 1:
        p_ref = F.max_pool1d(a[:, :, 0], 3, 2, 1)[:, :, None]
    else:
        p_ref = F.max_pool2d(a, 3, 2, 1)
    dp = fill.hash_tensor(tuple(p_ref.shape), 44)
    if dt == L.BF16:
        dp = bf16_round(dp)
    p_ref.backward(dp)
    coef = torch.zeros(4, Cn)
    coef[0], coef[1] = sc, sh
    yg, coefg = to_nhwc(y, dt), dev(coef)
    OH, OW = p_ref.shape[2], p_ref.shape[3]
    pg = torch.empty(N * OH * OW * Cn, device=DEV, dtype=TDT[dt])
    idx = torch.empty(N * OH * OW * Cn, device=DEV, dtype=torch.uint8)
    L.check(lib.ecgmm_bnrelu_maxpool(dt, ptr(yg), ptr(coefg), ptr(pg), ptr(idx), N, H, W, Cn, stream()))
    p = from_nhwc(pg, dt, p_ref.shape)
    assert (p - p_ref.detach()).abs().max() < (1e-5 if dt == L.F32 else 1e-2)
    dpg = to_nhwc(dp, dt)
    dzg = torch.empty(N * H * W * Cn, device=DEV, dtype=TDT[dt])
    L.check(lib.ecgmm_maxpool_relu_bwd(dt, ptr(dpg), ptr(pg), ptr(idx), ptr(dzg), N, H, W, Cn, stream()))
    dz = from_nhwc(dzg, dt, shape)
    # torch's grad w.r.t. a, then through the ReLU
    ref = a.grad * (a.detach() > 0)
    if dt == L.F32:
        assert rel_err(dz, ref) < 1e-6
    else:  # bf16 ties may pick a different (equal-valued) argmax: compare the total routed mass per window instead
        assert abs(dz.sum().item() - ref.sum().item()) < 1e-2 * ref.abs().sum().item()
        assert rel_err(dz, ref) < 0.05


@pytest.mark.parametrize("dt", [L.F32, L.BF16])
@pytest.mark.parametrize("shape", [(2, 64, 12, 10), (3, 64, 1, 41), (2, 64, 9, 7), (4, 64, 28, 28)])
def test_stem_backward_over_pooled_tensors_matches_the_two_pass_form(dt, shape):
    """ecgmm_pool_bn_bwd == ecgmm_maxpool_relu_bwd + ecgmm_bn_bwd == torch autograd of maxpool(relu(bn(y)))
    (resnet18 stem bn1/relu/maxpool; ResNet1D_SE.initial[1:4], multimodal_paper_modal_balance.py:100-104)"""
    N, Cn, H, W = shape
    lib = L.lib()
    M = N * H * W
    y = fill.hash_tensor(shape, 51, 2.0)
    if dt == L.BF16:
        y = bf16_round(y)
    gam = 1 + 0.3 * fill.hash_tensor((Cn,), 52)
    bet = 0.2 * fill.hash_tensor((Cn,), 53)
    yr, gr, br = y.clone().requires_grad_(True), gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    z = F.relu(F.batch_norm(yr, None, None, gr, br, True, 0.1, 1e-5))
    p_ref = F.max_pool1d(z[:, :, 0], 3, 2, 1)[:, :, None] if H == 1 else F.max_pool2d(z, 3, 2, 1)
    dp = fill.hash_tensor(tuple(p_ref.shape), 54)
    if dt == L.BF16:
        dp = bf16_round(dp)
    p_ref.backward(dp)

    yg = to_nhwc(y, dt)
    rows = lib.ecgmm_col_stats_rows(dt, M, Cn)
    partial = torch.empty(rows, 2, Cn, device=DEV)
    L.check(lib.ecgmm_col_stats(dt, ptr(yg), M, Cn, ptr(partial), stream()))
    coef = torch.empty(4, Cn, device=DEV)
    rmg, rvg, nbt = dev(torch.zeros(Cn)), dev(torch.ones(Cn)), torch.zeros((), dtype=torch.int64, device=DEV)
    gg, bg = dev(gam), dev(bet)
    L.check(lib.ecgmm_bn_finalize(ptr(partial), rows, Cn, float(M), ptr(gg), ptr(bg), ptr(rmg), ptr(rvg), ptr(nbt),
                                  0.1, 1e-5, ptr(coef), stream()))
    OH, OW = p_ref.shape[2], p_ref.shape[3]
    pg = torch.empty(N * OH * OW * Cn, device=DEV, dtype=TDT[dt])
    idx = torch.empty(N * OH * OW * Cn, device=DEV, dtype=torch.uint8)
    L.check(lib.ecgmm_bnrelu_maxpool(dt, ptr(yg), ptr(coef), ptr(pg), ptr(idx), N, H, W, Cn, stream()))
    dpg = to_nhwc(dp, dt)
    scratch = torch.empty(lib.ecgmm_bn_bwd_scratch(dt, M, Cn), device=DEV, dtype=torch.uint8)

    def two_pass():
        dzg, dyg = torch.empty_like(yg), torch.empty_like(yg)
        dgam, dbet, dbias = (torch.zeros(Cn, device=DEV) for _ in range(3))
        L.check(lib.ecgmm_maxpool_relu_bwd(dt, ptr(dpg), ptr(pg), ptr(idx), ptr(dzg), N, H, W, Cn, stream()))
        L.check(lib.ecgmm_bn_bwd(dt, ptr(dzg), None, None, None, 1, ptr(yg), ptr(coef), ptr(gg), ptr(dgam), ptr(dbet),
                                 ptr(dyg), None, ptr(dbias), M, Cn, ptr(scratch), stream()))
        return from_nhwc(dyg, dt, shape), dgam.cpu(), dbet.cpu(), dbias.cpu()

    def pooled_form():
        dyg = torch.empty_like(yg)
        dgam, dbet, dbias = (torch.zeros(Cn, device=DEV) for _ in range(3))
        L.check(lib.ecgmm_pool_bn_bwd(dt, ptr(dpg), ptr(pg), ptr(idx), ptr(yg), ptr(coef), ptr(gg), ptr(dgam), ptr(dbet),
                                      ptr(dyg), ptr(dbias), N, H, W, Cn, ptr(scratch), stream()))
        return from_nhwc(dyg, dt, shape), dgam.cpu(), dbet.cpu(), dbias.cpu()

    a, b = two_pass(), pooled_form()
    tol = 2e-5 if dt == L.F32 else 1.5e-2
    assert rel_err(b[0], a[0]) < tol and rel_err(b[1], a[1]) < tol and rel_err(b[2], a[2]) < tol
    assert b[3].abs().max().item() < (1e-3 if dt == L.F32 else 0.5)  # analytically zero
    gtol = 2e-4 if dt == L.F32 else 6e-2   # bf16: torch pools the unrounded activations, ties route differently
    assert rel_err(b[0], yr.grad) < gtol and rel_err(b[1], gr.grad) < gtol and rel_err(b[2], br.grad) < gtol
    # no dbias, no parameter gradients (frozen BatchNorm parameters): same dy
    dyg = torch.empty_like(yg)
    L.check(lib.ecgmm_pool_bn_bwd(dt, ptr(dpg), ptr(pg), ptr(idx), ptr(yg), ptr(coef), ptr(gg), None, None, ptr(dyg),
                                  None, N, H, W, Cn, ptr(scratch), stream()))
    assert torch.equal(from_nhwc(dyg, dt, shape), b[0])


def test_avgpool_bcast_and_se_gate_grad():
    lib = L.lib()
    N, R, Cn = 3, 37, 128
    for dt in (L.F32, L.BF16):
        x = fill.hash_tensor((N, Cn, 1, R), 51)
        if dt == L.BF16:
            x = bf16_round(x)
        xg = to_nhwc(x, dt)
        out = torch.empty(N, Cn, device=DEV)
        L.check(lib.ecgmm_avgpool(dt, ptr(xg), ptr(out), N, R, Cn, None, stream()))
        torch.cuda.synchronize()
        assert torch.allclose(out.cpu(), x.mean(dim=(2, 3)), atol=1e-5)
        v = dev(fill.hash_tensor((N, Cn), 52))
        b = torch.empty(N * R * Cn, device=DEV, dtype=TDT[dt])
        L.check(lib.ecgmm_bcast_rows(dt, ptr(v), ptr(b), N, R, Cn, 0.25, stream()))
        bb = from_nhwc(b, dt, (N, Cn, 1, R))
        assert torch.allclose(bb, (v.cpu() * 0.25)[:, :, None, None].expand(N, Cn, 1, R), atol=4e-3)
        dout = fill.hash_tensor((N, Cn, 1, R), 53)
        mref = fill.hash_tensor((N, Cn, 1, R), 54)
        if dt == L.BF16:
            dout, mref = bf16_round(dout), bf16_round(mref)
        coef = torch.zeros(4, Cn)
        coef[0], coef[1] = 1 + 0.1 * fill.hash_tensor((Cn,), 55), 0.1 * fill.hash_tensor((Cn,), 56)
        dg = torch.empty(N, Cn, device=DEV)
        doutg, mrefg, coefg = to_nhwc(dout, dt), to_nhwc(mref, dt), dev(coef)   # keep alive across the launch
        L.check(lib.ecgmm_se_gate_grad(dt, ptr(doutg), ptr(mrefg), ptr(xg), ptr(coefg), ptr(dg), N, R, Cn, stream()))
        torch.cuda.synchronize()
        z = x * coef[0][None, :, None, None] + coef[1][None, :, None, None]
        ref = (dout * (mref > 0) * z).sum(dim=(2, 3))
        assert rel_err(dg.cpu(), ref) < 1e-5


def test_adam_matches_torch():
    lib = L.lib()
    n = 4099
    p0, g0 = fill.hash_tensor((n,), 61), fill.hash_tensor((n,), 62, 0.1)
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=1e-3, betas=(0.95, 0.999))
    pg, m, v = dev(p0.clone()), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step in range(1, 4):
        g = g0 * step
        pr.grad = g.clone()
        opt.step()
        gg = dev(g * 2.0)  # gscale = 0.5 undoes the factor
        L.check(lib.ecgmm_adam(ptr(pg), ptr(gg), ptr(m), ptr(v), n, 1e-3, 0.95, 0.999, 1e-8, 0.0, step, 0.5, stream()))
    torch.cuda.synchronize()
    assert torch.allclose(pg.cpu(), pr.detach(), rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("B,H,W,cap,act,use_bias", [(16, 56, 56, 0, 0, 0), (20, 56, 56, 7, 0, 0), (4, 32, 32, 1, 0, 0), (12, 56, 56, 5, 1, 1), (40, 56, 56, 100, 0, 0), (8, 8, 60, 3, 0, 0)])
def test_stream_form_of_64_channel_halo_tiles_matches_tile_at_a_time(B, H, W, cap, act, use_bias):
    """conv_halo_kernel's stream form (64 -> 64 channel 3x3 tiles: K loop continuous across tile boundaries, previous tile's
    epilogue inside the next tile's steps, address table rebuilt in place) against the tile-at-a-time form of the same kernel and
    against the fp32 reference: one tile per workgroup (no boundary at all), many tiles on few workgroups (CU cap 1 / 3 / 7),
    tile counts that are not a multiple of the workgroup count, bias + ReLU.  Output, statistics rows and input gradient must be
    bit-identical between the two forms."""
    lib = L.lib()
    lib.ecgmm_conv_halo_enable(2)
    d = L.ConvDesc(B, H, W, 64, 64, 3, 3, 1, 1, 1)
    g = torch.Generator(device=DEV).manual_seed(B * 131 + H)
    n = B * H * W * 64
    x = torch.randn(n, device=DEV, generator=g).to(torch.bfloat16)
    w = (torch.randn(64 * 64 * 9, device=DEV, generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(64, device=DEV, generator=g) if use_bias else None
    add = torch.randn(n, device=DEV, generator=g).to(torch.bfloat16)
    res = {}
    try:
        lib.ecgmm_conv_halo_cus(cap)
        for on in (0, 1):
            lib.ecgmm_conv_halo_stream(on)
            y = torch.full((n,), 7.0, device=DEV).to(torch.bfloat16)
            dx = torch.full((n,), 7.0, device=DEV).to(torch.bfloat16)
            y2 = torch.full((n,), 7.0, device=DEV).to(torch.bfloat16)
            st = torch.zeros(600 * 2 * 64, device=DEV)
            rows = C.c_int(0)
            L.check(lib.ecgmm_conv_fwd_wgrows(L.BF16, C.byref(d), ptr(x), ptr(w), ptr(bias) if use_bias else None, ptr(y), ptr(st), C.byref(rows), act, stream()))
            L.check(lib.ecgmm_conv_bwd_data(L.BF16, C.byref(d), ptr(x), ptr(w), None, ptr(dx), stream()))
            L.check(lib.ecgmm_conv_fwd(L.BF16, C.byref(d), ptr(x), ptr(w), None, ptr(y2), None, act, stream()))
            dxa = torch.full((n,), 7.0, device=DEV).to(torch.bfloat16)
            L.check(lib.ecgmm_conv_bwd_data(L.BF16, C.byref(d), ptr(x), ptr(w), ptr(add), ptr(dxa), stream()))
            torch.cuda.synchronize()
            res[on] = (y, st[: rows.value * 2 * 64].clone(), dx, y2, rows.value, dxa)
    finally:
        lib.ecgmm_conv_halo_cus(0)
        lib.ecgmm_conv_halo_stream(1)
        lib.ecgmm_conv_halo_enable(1)
    assert res[0][4] == res[1][4] and res[1][4] >= 1
    for i, name in ((0, "output"), (1, "statistics rows"), (2, "input gradient"), (3, "output without statistics"), (5, "input gradient + residual addend")):
        a, b = res[0][i], res[1][i]
        assert torch.equal(a.view(torch.int16) if a.dtype == torch.bfloat16 else a, b.view(torch.int16) if b.dtype == torch.bfloat16 else b), name
    # ... and both are the convolution: fp32 reference on the same bf16 operands
    xr = x.float().view(B, H, W, 64).permute(0, 3, 1, 2)
    wr = w.float().view(64, 3, 3, 64).permute(0, 3, 1, 2)
    ref = F.conv2d(xr, wr, bias, padding=1)
    if act:
        ref = ref.relu()
    got = res[1][0].float().view(B, H, W, 64).permute(0, 3, 1, 2)
    assert rel_err(got, ref) < 6e-3
    # the addend form = the plain input gradient + the addend, rounded once
    want = (res[1][2].float() * 0 + (res[1][5].float() - add.float()))
    assert rel_err(want, res[1][2].float()) < 1.5e-2
    tot = res[1][1].view(-1, 2, 64).sum(0)
    assert rel_err(tot[0], ref.sum((0, 2, 3))) < 2e-3 or ref.sum((0, 2, 3)).abs().max() < 1.0
    assert rel_err(tot[1], (ref * ref).sum((0, 2, 3))) < 2e-3
