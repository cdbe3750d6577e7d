"""CPU restatement (numpy / scipy) of the reference's signal pre-processing -- TEST INFRASTRUCTURE ONLY.
Follows dataset.py:81-95 (1-D) == train_signal_12_af.py:19-34 ([leads, time]) line by line; pinned
bit-for-bit to the reference's own ``preprocess_signal`` by oracle/make_golden.py (g7)."""
import numpy as np
from scipy.signal import butter, filtfilt


def remove_baseline_drift(signal, window_size=200):
    conv = lambda x: np.convolve(x, np.ones(window_size) / window_size, mode="same")
    baseline = np.apply_along_axis(conv, -1, signal)
    return signal - baseline


def lowpass_filter(signal, cutoff=0.05, fs=1.0, order=5):
    b, a = butter(order, cutoff / (0.5 * fs), btype="low", analog=False)
    return np.apply_along_axis(lambda x: filtfilt(b, a, x), -1, signal)


def preprocess_signal(raw_signal):
    return lowpass_filter(remove_baseline_drift(np.asarray(raw_signal, dtype=np.float64))).copy()


def standard_scale(x, mean, scale):
    """sklearn StandardScaler.transform over time columns (dataset.py:195-200)."""
    return (np.asarray(x, dtype=np.float64) - mean) / scale
