"""on-device signal pre-processing vs the reference's own preprocess_signal (golden g7) and,
live, vs the scipy restatement on other shapes (short signals, odd lengths, StandardScaler in front)."""
import numpy as np
import pytest
import torch

from ecgmm import preprocess as PP
from oracle import fill, preprocess_ref as PR

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_matches_reference_function_golden(golden_dir):
    g7 = np.load(f"{golden_dir}/g7_preprocess.npz")
    x = torch.from_numpy(g7["x"]).to(DEV)
    y = PP.preprocess_signal(x)
    torch.cuda.synchronize()
    assert y.dtype == torch.float32 and y.shape == x.shape
    assert np.abs(y.cpu().numpy().astype(np.float64) - g7["y"]).max() < 2e-6          # fp32 output rounding only
    ys = PP.preprocess_signal(x[:3, :1000].contiguous())
    assert np.abs(ys.cpu().numpy().astype(np.float64) - g7["y_short"]).max() < 2e-6
    bl = PP.remove_baseline_drift(x)
    assert np.abs(bl.cpu().numpy().astype(np.float64) - g7["baseline_removed"]).max() < 2e-6


@pytest.mark.parametrize("shape", [(256, 5000), (5, 12, 2476), (3, 201), (70, 777)])
def test_matches_scipy_restatement_live(shape):
    x = fill.hash_tensor(shape, 909, 1.5).numpy().astype(np.float64)
    x += 0.5 * np.cos(np.arange(shape[-1]) / 11.0)
    Ln = shape[-1]
    mean = 0.1 * fill.hash_uniform(Ln, 910).astype(np.float64)
    scale = 1.0 + 0.3 * np.abs(fill.hash_uniform(Ln, 911)).astype(np.float64)
    x32 = x.astype(np.float32)
    ref = PR.preprocess_signal(PR.standard_scale(x32.astype(np.float64), mean.astype(np.float32).astype(np.float64),
                                                 scale.astype(np.float32).astype(np.float64)))
    y = PP.preprocess_signal(torch.from_numpy(x32).to(DEV), scaler_mean=mean, scaler_scale=scale)
    torch.cuda.synchronize()
    assert np.abs(y.cpu().numpy().astype(np.float64) - ref).max() < 3e-6 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("shape,order,cutoff", [((4, 12000), 5, 0.05), ((6, 3000), 3, 0.1), ((6, 999), 7, 0.08),
                                                ((2, 4000), 1, 0.2), ((3, 20000), 2, 0.05)])
def test_other_orders_and_long_records(shape, order, cutoff):
    """records too long for one CU's LDS take the workspace kernel; every supported order has its own instantiation"""
    x = fill.hash_tensor(shape, 313, 1.0).numpy().astype(np.float32)
    x += np.sin(np.arange(shape[-1]) / 37.0).astype(np.float32)
    ref = PR.lowpass_filter(PR.remove_baseline_drift(x.astype(np.float64)), cutoff=cutoff, order=order)
    y = PP.preprocess_signal(torch.from_numpy(x).to(DEV), cutoff=cutoff, order=order)
    torch.cuda.synchronize()
    assert np.abs(y.cpu().numpy().astype(np.float64) - ref).max() < 3e-6 * max(1.0, np.abs(ref).max())


def test_rejects_bad_arguments():
    x = torch.zeros(2, 100, device=DEV)
    with pytest.raises(RuntimeError, match="window"):
        PP.preprocess_signal(x)                       # shorter than the 200-tap window, as np.convolve 'same' would mis-size
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        PP.preprocess_signal(torch.zeros(2, 5000))
