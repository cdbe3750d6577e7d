"""Multimodal ECG model (image + 1-D signal + clinical) on MI355X HIP kernels.

Same classes, constructor/forward signatures, sub-module attribute names and state_dict keys as the
reference's ``multimodal_paper_modal_balance.py`` (AttentionFusion :31-46, SEBlock :49-64,
BasicBlock1D :67-93, ResNet1D_SE :96-125, ECGMultimodalModel :197-384).  Differences, all additive:
  * pretrained checkpoints are optional (the reference hard-loads files that are not in its tree,
    :215, :234-237) -- pass paths through ``config.pretrained_*`` or call ``load_pretrained_*``;
  * dims / clinical width / compute dtype come from ``config`` with the reference's values as defaults.
Everything numeric runs in libecgmm_hip.so; there is no CPU path.
"""
import torch
import torch.nn as nn

from .hip import encoders as E
from .hip import functional as HF
from .hip import nn as hnn
from .image_encoder import ResNet18, resnet18


class AttentionFusion(nn.Module):
    """softmax over 3 learnable scalars -> scale each modality -> concat -> LayerNorm (one HIP kernel)."""

    def __init__(self, dims):
        super().__init__()
        self.dims = list(dims)
        self.weights = nn.Parameter(torch.ones(3))  # image, signal, clinical
        self.norm = hnn.LayerNorm(sum(dims))

    def forward(self, img_feat, signal_feat, clinical_feat):
        fused, soft_weights = HF.attention_fusion(img_feat, signal_feat, clinical_feat, self.weights,
                                                  self.norm.weight, self.norm.bias, self.norm.eps)
        return fused, soft_weights


class SEBlock(nn.Module):
    def __init__(self, channels, reduction=16):
        super().__init__()
        self.pool = nn.AdaptiveAvgPool1d(1)
        self.fc = nn.Sequential(
            hnn.Linear(channels, channels // reduction), nn.ReLU(),
            hnn.Linear(channels // reduction, channels), nn.Sigmoid(),
        )

    def forward(self, x):
        """x * sigmoid(fc(mean_L x)) on the per-op kernels (ecgmm/hip/blocks.py).  Stand-alone use only -- inside
        ``ResNet1D_SE`` the encoder's launch plan runs the block; no autograd graph is built here."""
        from .hip import blocks
        return blocks.se_block_forward(self, x)


class BasicBlock1D(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1):
        super().__init__()
        if kernel_size != 3:
            raise ValueError("BasicBlock1D: the HIP plan implements the reference's kernel_size=3")
        padding = kernel_size // 2
        self.conv1 = hnn.Conv1d(in_channels, out_channels, kernel_size, stride=stride, padding=padding)
        self.bn1 = hnn.BatchNorm1d(out_channels)
        self.relu = nn.ReLU()
        self.conv2 = hnn.Conv1d(out_channels, out_channels, kernel_size, padding=padding)
        self.bn2 = hnn.BatchNorm1d(out_channels)
        self.se = SEBlock(out_channels)
        self.downsample = None
        if in_channels != out_channels or stride != 1:
            self.downsample = nn.Sequential(
                hnn.Conv1d(in_channels, out_channels, kernel_size=1, stride=stride),
                hnn.BatchNorm1d(out_channels),
            )

    compute_dtype = "fp32"     # stand-alone forward (ResNet1D_SE hands its own dtype to its blocks)

    def forward(self, x):
        """relu(se(bn2(conv2(relu(bn1(conv1 x))))) + identity) on the per-op kernels (ecgmm/hip/blocks.py): the values
        the encoder's launch plan computes for this block, for inspection / feature extraction.  No autograd graph
        (training goes through ``ResNet1D_SE.forward``); in train mode the BatchNorm running statistics update."""
        from .hip import blocks
        return blocks.basic_block1d_forward(self, x, self.compute_dtype)


class ResNet1D_SE(nn.Module):
    def __init__(self, input_channels=1, num_classes=2, base_filters=64, compute_dtype="bf16"):
        super().__init__()
        if base_filters != 64:
            raise ValueError("ResNet1D_SE: the HIP plan implements the reference's base_filters=64")
        self.input_channels = input_channels
        self.compute_dtype = compute_dtype
        self.initial = nn.Sequential(
            hnn.Conv1d(input_channels, base_filters, kernel_size=7, stride=2, padding=3),
            hnn.BatchNorm1d(base_filters),
            nn.ReLU(),
            nn.MaxPool1d(kernel_size=3, stride=2, padding=1),
        )
        self.layer1 = BasicBlock1D(base_filters, base_filters)
        self.layer2 = BasicBlock1D(base_filters, base_filters * 2, stride=2)
        self.layer3 = BasicBlock1D(base_filters * 2, base_filters * 4, stride=2)
        self.global_pool = nn.AdaptiveAvgPool1d(1)
        self.classifier = nn.Sequential(
            nn.Flatten(),
            hnn.Linear(base_filters * 4, 64),
            nn.ReLU(),
            nn.Dropout(0.3),
            hnn.Linear(64, num_classes),
        )
        self._spec = E.ResNet1DSpec()
        for blk in (self.layer1, self.layer2, self.layer3):
            blk.compute_dtype = compute_dtype

    def forward(self, x):
        spec = self._spec
        spec.dtype = E.dtype_code(self.compute_dtype)
        spec.training = self.training
        bn = self.initial[1]
        spec.momentum, spec.eps = bn.momentum, bn.eps
        spec.cin = self.input_channels
        spec.out_dim = self.classifier[4].weight.shape[0]
        spec.dropout_p = float(self.classifier[3].p) if self.classifier[3].training else 0.0
        spec.buffers = list(self.buffers())
        params = list(self.parameters())
        if x.is_cuda:
            # weight gradients on the plan's own side stream only when this encoder has the GPU to itself: beside the
            # image encoder (ECGMultimodalModel sets _beside_image_encoder) a fifth busy stream oversubscribes the
            # four hardware queues (csrc/plan_resnet1d.hip, ecgmm_resnet1d_side_wgrad)
            from .hip import lib as _L
            _L.lib().ecgmm_resnet1d_side_wgrad(0 if getattr(self, "_beside_image_encoder", False) else 1)
        if len(params) != 52 or len(spec.buffers) != 27:
            raise RuntimeError(f"ResNet1D_SE expects 52 parameters / 27 buffers, found {len(params)} / {len(spec.buffers)}")
        return E.run_plan(x, spec, params)


import os as _os
_OVERLAP_ENV = _os.environ.get("ECGMM_OVERLAP_ENCODERS", "1") != "0"   # A/B switch: 0 = the three encoders on one stream


class ECGMultimodalModel(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        cd = getattr(config, "compute_dtype", "bf16")

        self.modal_dim = getattr(config, "modal_dim", 256)
        self.image_dim, self.signal_dim, self.clinical_dim = self._branch_dims()

        # image encoder (ResNet18) + LayerNorm
        self.image_encoder = resnet18(compute_dtype=cd)
        self.image_encoder.fc = hnn.Linear(self.image_encoder.fc.in_features, self.image_dim)
        self.image_norm = hnn.LayerNorm(self.image_dim)

        # signal encoder (ResNet1D_SE) + LayerNorm
        self.signal_encoder = ResNet1D_SE(input_channels=1, num_classes=self.signal_dim, compute_dtype=cd)
        self.signal_norm = hnn.LayerNorm(self.signal_dim)

        # clinical encoder + LayerNorm
        p = getattr(config, "dropout", 0.3)
        self.clinical_encoder = self._build_clinical_encoder(p)
        self.clinical_norm = hnn.LayerNorm(self.clinical_dim)

        # branch classifiers
        self.image_classifier = hnn.Linear(self.image_dim, config.num_classes)
        self.signal_classifier = hnn.Linear(self.signal_dim, config.num_classes)
        self.clinical_classifier = hnn.Linear(self.clinical_dim, config.num_classes)

        # attention fusion + fusion classifier
        self.attention_fusion = AttentionFusion(dims=[self.image_dim, self.signal_dim, self.clinical_dim])
        self.fusion_classifier = hnn.Sequential(
            hnn.Linear(self.image_dim + self.signal_dim + self.clinical_dim, 128),
            nn.ReLU(),
            nn.Dropout(p),
            hnn.Linear(128, config.num_classes),
        )

        img_ckpt = getattr(config, "pretrained_image_encoder", None)
        if img_ckpt:
            self.load_pretrained_image_encoder(img_ckpt, load_fc=False)
        sig_ckpt = getattr(config, "pretrained_signal_encoder", None)
        if sig_ckpt:
            self.load_pretrained_signal_encoder(sig_ckpt, load_fc=False)

    def get_clinical_feature_dim(self):
        return getattr(self.config, "clinical_input_dim", 24)

    # ---- the two hooks in which multimodal.py's variant of this model differs (ecgmm/multimodal.py)
    def _branch_dims(self):
        return self.modal_dim, self.modal_dim, self.modal_dim

    def _build_clinical_encoder(self, p):
        return hnn.Sequential(
            hnn.Linear(self.get_clinical_feature_dim(), 64),
            hnn.BatchNorm1d(64),
            nn.ReLU(),
            nn.Dropout(p),
            hnn.Linear(64, self.clinical_dim),
        )

    def _clinical_forward(self, clinical):
        return self.clinical_encoder(clinical)

    def load_pretrained_signal_encoder(self, weight_path, load_fc=False):
        checkpoint = torch.load(weight_path, map_location="cpu")
        if not load_fc:
            checkpoint = {k: v for k, v in checkpoint.items() if not k.startswith("classifier.4")}
        missing, unexpected = self.signal_encoder.load_state_dict(checkpoint, strict=False)
        print(f"Loaded pretrained signal encoder from {weight_path}")
        if missing:
            print(f"Missing keys: {missing}")
        if unexpected:
            print(f"Unexpected keys: {unexpected}")

    def load_pretrained_image_encoder(self, weight_path: str, load_fc: bool = False):
        """PMB:356-384: the reference loads the file into a temporary resnet18 whose fc has the checkpoint's width, so a
        tensor of the wrong shape RAISES there (PMB:369-371) -- it does here too; only `fc.*` is dropped when
        ``load_fc`` is false (PMB:378), and keys the encoder does not have are ignored (strict=False)."""
        saved_state = torch.load(weight_path, map_location="cpu")
        current = self.image_encoder.state_dict()
        new_state = {k: v for k, v in saved_state.items() if k in current}
        # (checked on every matching key, `fc.*` included even when it is dropped below: the reference's temporary
        # resnet has an fc of this model's image_dim and load_state_dict raises on it first)
        bad = [(k, tuple(v.shape), tuple(current[k].shape)) for k, v in new_state.items() if v.shape != current[k].shape]
        if bad:
            raise RuntimeError("load_pretrained_image_encoder: size mismatch for " +
                               ", ".join(f"{k}: checkpoint {a} vs model {b}" for k, a, b in bad))
        if not load_fc:
            new_state = {k: v for k, v in new_state.items() if not k.startswith("fc.")}
        current.update(new_state)
        self.image_encoder.load_state_dict(current)
        print(f"Image encoder weights loaded from {weight_path} (load_fc={load_fc}, {len(new_state)} tensors)")

    def forward(self, image, ecg_signal, clinical):
        ecg_signal = ecg_signal.unsqueeze(1)
        if image.is_cuda and getattr(self.config, "overlap_encoders", True) and _OVERLAP_ENV:
            # The three encoders are independent until the fusion: run the (small-kernel) signal and
            # clinical branches on a side HIP stream underneath the image encoder's launches.  autograd
            # replays each backward node on the stream its forward ran on, so the backward overlaps too.
            main = torch.cuda.current_stream(image.device)
            side = self._side_stream = getattr(self, "_side_stream", None) or torch.cuda.Stream(image.device)
            side.wait_stream(main)
            self.signal_encoder._beside_image_encoder = True
            with torch.cuda.stream(side):
                signal_raw = self.signal_encoder(ecg_signal)
                clinical_raw = self._clinical_forward(clinical)
            image_raw = self.image_encoder(image)
            main.wait_stream(side)
            signal_raw.record_stream(main)
            clinical_raw.record_stream(main)
        else:
            self.signal_encoder._beside_image_encoder = False
            image_raw = self.image_encoder(image)
            signal_raw = self.signal_encoder(ecg_signal)
            clinical_raw = self._clinical_forward(clinical)
        spec = self._head_spec()
        if spec is not None and image_raw.is_cuda:
            # everything after the encoders as ONE native call per direction (csrc/plan_head.hip): the same kernels as
            # the module-by-module path below, without ~85 autograd nodes' worth of host work between the encoders'
            # forward and backward, where no stream has anything else queued
            return E.run_head(image_raw, signal_raw, clinical_raw, spec, self._head_params())
        return self._head_by_modules(image_raw, signal_raw, clinical_raw)

    def _head_params(self):
        fc = self.fusion_classifier
        return [self.image_norm.weight, self.image_norm.bias, self.signal_norm.weight, self.signal_norm.bias,
                self.clinical_norm.weight, self.clinical_norm.bias,
                self.image_classifier.weight, self.image_classifier.bias, self.signal_classifier.weight,
                self.signal_classifier.bias, self.clinical_classifier.weight, self.clinical_classifier.bias,
                self.attention_fusion.weights, self.attention_fusion.norm.weight, self.attention_fusion.norm.bias,
                fc[0].weight, fc[0].bias, fc[3].weight, fc[3].bias]

    def _head_spec(self):
        """launch description of the fused head, or None when the module tree is not the reference's
        Linear-ReLU-Dropout-Linear / LayerNorm layout (then the modules run one by one)"""
        fc = self.fusion_classifier
        norms = (self.image_norm, self.signal_norm, self.clinical_norm, self.attention_fusion.norm)
        ok = (len(fc) == 4 and isinstance(fc[0], hnn.Linear) and isinstance(fc[1], nn.ReLU)
              and isinstance(fc[2], nn.Dropout) and isinstance(fc[3], hnn.Linear)
              and all(isinstance(n, hnn.LayerNorm) and n.eps == norms[0].eps for n in norms)
              and all(getattr(m, "bias", None) is not None for m in (fc[0], fc[3], self.image_classifier,
                                                                      self.signal_classifier, self.clinical_classifier))
              and getattr(self.config, "fused_head", True))
        if not ok:
            return None
        drop = fc[2]
        return E.HeadSpec(fc[0].out_features, fc[3].out_features, bool(self.training and drop.training), norms[0].eps,
                          float(drop.p))

    def _head_by_modules(self, image_raw, signal_raw, clinical_raw):
        img_feat = self.image_norm(image_raw)
        signal_feat = self.signal_norm(signal_raw)
        clinical_feat = self.clinical_norm(clinical_raw)

        img_logits = self.image_classifier(img_feat)
        signal_logits = self.signal_classifier(signal_feat)
        clinical_logits = self.clinical_classifier(clinical_feat)

        fused, soft_weights = self.attention_fusion(img_feat, signal_feat, clinical_feat)
        fusion_logits = self.fusion_classifier(fused)

        # variance regularisation (chunk-wise), one fused op
        var_loss = HF.var_loss(img_feat, signal_feat, clinical_feat)

        return img_logits, signal_logits, clinical_logits, fusion_logits, var_loss, soft_weights


MultimodalModel = ECGMultimodalModel  # name used by the north-star text / the reference README
