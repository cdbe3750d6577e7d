"""ECG signal pre-processing on the GPU with the reference's function names:
``remove_baseline_drift`` + ``lowpass_filter`` = ``preprocess_signal`` (dataset.py:81-95;
train_signal_12_af.py:19-34), optionally preceded by the per-column StandardScaler of
dataset.py:195-200.  The reference runs these per sample in DataLoader workers with numpy/scipy in
float64; here a whole batch is filtered by one HIP kernel (fp64 arithmetic, fp32 result).

Filter design is done natively on the host (no scipy at run time): digital Butterworth low-pass by
bilinear transform of the analog prototype, and ``lfilter_zi``'s steady-state initial conditions.
"""
import ctypes as C

import numpy as np
import torch

from .hip import lib as L
from .hip.functional import _Scratch, _require_cuda, ptr, stream


def butter_lowpass(order, wn):
    """== scipy.signal.butter(order, wn, 'low') (b, a): poles of the analog prototype on the unit circle,
    frequency pre-warping, bilinear transform with fs = 2."""
    k = np.arange(order)
    p = np.exp(1j * np.pi * (2 * k + order + 1) / (2 * order))
    fs = 2.0
    warped = 2 * fs * np.tan(np.pi * wn / fs)
    p = warped * p
    gain = warped ** order
    fs2 = 2 * fs
    p_d = (fs2 + p) / (fs2 - p)
    k_d = gain * np.real(1 / np.prod(fs2 - p))
    b = k_d * np.real(np.poly(-np.ones(order)))
    a = np.real(np.poly(p_d))
    return b, a


def lfilter_zi(b, a):
    """== scipy.signal.lfilter_zi: initial state of the transposed direct-form-II filter for a unit step."""
    b, a = np.asarray(b, dtype=np.float64) / a[0], np.asarray(a, dtype=np.float64) / a[0]
    n = len(a)
    comp = np.zeros((n - 1, n - 1))
    comp[0] = -a[1:]
    comp[1:, :-1] = np.eye(n - 2)
    return np.linalg.solve(np.eye(n - 1) - comp.T, b[1:] - a[1:] * b[0])


def preprocess_signal(raw_signal, window_size=200, cutoff=0.05, fs=1.0, order=5, scaler_mean=None, scaler_scale=None):
    """raw_signal: CUDA float tensor [..., L] (e.g. [B, L] or [B, leads, L]); returns the same shape, float32."""
    _require_cuda(raw_signal, "preprocess_signal")
    shape = raw_signal.shape
    x = raw_signal.reshape(-1, shape[-1]).float().contiguous()
    S, Ln = x.shape
    b, a = butter_lowpass(order, cutoff / (0.5 * fs))
    zi = lfilter_zi(b, a)
    out = torch.empty_like(x)
    lib = L.lib()
    nb = lib.ecgmm_signal_preprocess_workspace(S, Ln, order)
    ws = _Scratch.get("preprocess", nb, x.device)
    dbl = lambda v: (C.c_double * len(v))(*[float(t) for t in v])
    sm = None if scaler_mean is None else torch.as_tensor(scaler_mean, dtype=torch.float32, device=x.device).contiguous()
    ss = None if scaler_scale is None else torch.as_tensor(scaler_scale, dtype=torch.float32, device=x.device).contiguous()
    L.check(lib.ecgmm_signal_preprocess(ptr(x), ptr(out), S, Ln, ptr(sm), ptr(ss), int(window_size), dbl(b), dbl(a),
                                        dbl(zi), int(order), ptr(ws), ws.numel(), stream()), "signal_preprocess")
    return out.reshape(shape)


def remove_baseline_drift(signal, window_size=200):
    """signal - moving average ('same' convolution): the first half of preprocess_signal (order-1 pass-through
    is not expressible, so this helper runs the kernel with an identity filter b=[1,0], a=[1,0])."""
    _require_cuda(signal, "remove_baseline_drift")
    shape = signal.shape
    x = signal.reshape(-1, shape[-1]).float().contiguous()
    S, Ln = x.shape
    out = torch.empty_like(x)
    lib = L.lib()
    ws = _Scratch.get("preprocess", lib.ecgmm_signal_preprocess_workspace(S, Ln, 1), x.device)
    one = (C.c_double * 2)(1.0, 0.0)
    zi = (C.c_double * 1)(0.0)
    L.check(lib.ecgmm_signal_preprocess(ptr(x), ptr(out), S, Ln, None, None, int(window_size), one, one, zi, 1,
                                        ptr(ws), ws.numel(), stream()), "remove_baseline_drift")
    return out.reshape(shape)
