"""MI355X-native (gfx950) implementation of the ECG multimodal training hot path.

Python host code on PyTorch-ROCm (device memory, streams, torch.distributed) over hand-written HIP
kernels reached through the C ABI of ``libecgmm_hip.so`` (``include/ecgmm.h``).  Module names mirror
the reference's flat scripts: ``config``, ``multimodal_paper_modal_balance``, ``multimodal``,
``signal_model``, ``dataset``, ``train``, ``train_image_only``, ``train_signal_12_af``.
"""
__version__ = "0.1.0"
