"""CPU oracle: a plain-torch fp32 restatement of the reference's multimodal hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it, and only as the
checker / the timed CPU baseline.  The shipped path (``ecg-multimodal-model_amd/``) never imports
this package and fails loudly when its HIP library is missing.

Parity pinning (see DESIGN.md "Oracle"):
  * ``ResNet1D_SE`` / ``SEBlock`` / ``BasicBlock1D`` / ``FocalLoss`` are pinned bit-for-bit against the
    reference's own classes (imported from /root/reference in the build container by
    ``oracle/make_golden.py``) and its ``best_ptbxl.pth`` weights -> ``tests/golden/g1..g4``.
  * ``ResNet18`` restates torchvision's ``resnet18()`` (torchvision is not installed anywhere in this
    pipeline and the reference does not vendor it): parity there is structural only
    (state-dict keys/shapes, parameter counts) -- "parity unpinned" for that sub-graph.

Every class cites the reference lines it follows (paths relative to /root/reference).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------
# signal branch: multimodal_paper_modal_balance.py:49-125 (== signal_model.py:12-88)
# --------------------------------------------------------------------------------------------
class SEBlock(nn.Module):
    """multimodal_paper_modal_balance.py:49-64 -- gate = sigmoid(W2 relu(W1 mean_L(x)))."""

    def __init__(self, channels, reduction=16):
        super().__init__()
        self.pool = nn.AdaptiveAvgPool1d(1)
        self.fc = nn.Sequential(
            nn.Linear(channels, channels // reduction),
            nn.ReLU(),
            nn.Linear(channels // reduction, channels),
            nn.Sigmoid(),
        )

    def forward(self, x):
        b, c, _ = x.size()
        gate = self.fc(self.pool(x).view(b, c)).view(b, c, 1)
        return x * gate


class BasicBlock1D(nn.Module):
    """multimodal_paper_modal_balance.py:67-93 (Conv1d keeps bias=True in front of BN)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1):
        super().__init__()
        pad = kernel_size // 2
        self.conv1 = nn.Conv1d(in_channels, out_channels, kernel_size, stride=stride, padding=pad)
        self.bn1 = nn.BatchNorm1d(out_channels)
        self.relu = nn.ReLU()
        self.conv2 = nn.Conv1d(out_channels, out_channels, kernel_size, padding=pad)
        self.bn2 = nn.BatchNorm1d(out_channels)
        self.se = SEBlock(out_channels)
        self.downsample = None
        if in_channels != out_channels or stride != 1:
            self.downsample = nn.Sequential(
                nn.Conv1d(in_channels, out_channels, kernel_size=1, stride=stride),
                nn.BatchNorm1d(out_channels),
            )

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.se(self.bn2(self.conv2(out)))
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


class ResNet1D_SE(nn.Module):
    """multimodal_paper_modal_balance.py:96-125."""

    def __init__(self, input_channels=1, num_classes=2, base_filters=64):
        super().__init__()
        f = base_filters
        self.initial = nn.Sequential(
            nn.Conv1d(input_channels, f, kernel_size=7, stride=2, padding=3),
            nn.BatchNorm1d(f),
            nn.ReLU(),
            nn.MaxPool1d(kernel_size=3, stride=2, padding=1),
        )
        self.layer1 = BasicBlock1D(f, f)
        self.layer2 = BasicBlock1D(f, f * 2, stride=2)
        self.layer3 = BasicBlock1D(f * 2, f * 4, stride=2)
        self.global_pool = nn.AdaptiveAvgPool1d(1)
        self.classifier = nn.Sequential(
            nn.Flatten(), nn.Linear(f * 4, 64), nn.ReLU(), nn.Dropout(0.3), nn.Linear(64, num_classes)
        )

    def stages(self, x):
        s0 = self.initial(x)
        s1 = self.layer1(s0)
        s2 = self.layer2(s1)
        s3 = self.layer3(s2)
        return s0, s1, s2, s3

    def forward(self, x):
        return self.classifier(self.global_pool(self.stages(x)[-1]))


class FocalLoss(nn.Module):
    """signal_model.py:91-106: alpha (1-exp(-CE))^gamma CE, mean."""

    def __init__(self, alpha=1.0, gamma=2.0, logits=True, reduce=True):
        super().__init__()
        self.alpha, self.gamma, self.logits, self.reduce = alpha, gamma, logits, reduce

    def forward(self, inputs, targets):
        if self.logits:
            ce = F.cross_entropy(inputs, targets, reduction="none")
        else:
            ce = F.nll_loss(inputs, targets, reduction="none")
        pt = torch.exp(-ce)
        fl = self.alpha * (1 - pt) ** self.gamma * ce
        return fl.mean() if self.reduce else fl


# --------------------------------------------------------------------------------------------
# image branch: torchvision resnet18 restated (instantiated at
# multimodal_paper_modal_balance.py:210, fc replaced at :221; train_image_only.py:92-99)
# --------------------------------------------------------------------------------------------
class BasicBlock2D(nn.Module):
    def __init__(self, inplanes, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU()
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(
                nn.Conv2d(inplanes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes)
            )

    def forward(self, x):
        idt = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.downsample is not None:
            idt = self.downsample(x)
        return self.relu(out + idt)


class ResNet18(nn.Module):
    """BasicBlock [2,2,2,2]; state-dict keys identical to torchvision's."""

    def __init__(self, num_classes=1000):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU()
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = nn.Sequential(BasicBlock2D(64, 64), BasicBlock2D(64, 64))
        self.layer2 = nn.Sequential(BasicBlock2D(64, 128, 2), BasicBlock2D(128, 128))
        self.layer3 = nn.Sequential(BasicBlock2D(128, 256, 2), BasicBlock2D(256, 256))
        self.layer4 = nn.Sequential(BasicBlock2D(256, 512, 2), BasicBlock2D(512, 512))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def stages(self, x):
        s0 = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        s1 = self.layer1(s0)
        s2 = self.layer2(s1)
        s3 = self.layer3(s2)
        s4 = self.layer4(s3)
        return s0, s1, s2, s3, s4

    def forward(self, x):
        return self.fc(torch.flatten(self.avgpool(self.stages(x)[-1]), 1))


class ImageOnlyClassifier(nn.Module):
    """train_image_only.py:92-99 (weights: random init; the reference's ImageNet fetch is offline)."""

    def __init__(self, num_classes=2):
        super().__init__()
        self.image_encoder = ResNet18()
        self.image_encoder.fc = nn.Linear(512, num_classes)

    def forward(self, x):
        return self.image_encoder(x)


# --------------------------------------------------------------------------------------------
# fusion: multimodal_paper_modal_balance.py:31-46, 197-354
# --------------------------------------------------------------------------------------------
class AttentionFusion(nn.Module):
    def __init__(self, dims):
        super().__init__()
        self.weights = nn.Parameter(torch.ones(3))
        self.norm = nn.LayerNorm(sum(dims))

    def forward(self, img, sig, clin):
        w = torch.softmax(self.weights, dim=0)
        return self.norm(torch.cat([w[0] * img, w[1] * sig, w[2] * clin], dim=1)), w


class ECGMultimodalModel(nn.Module):
    """multimodal_paper_modal_balance.py:197-354 with the hard-coded checkpoint loads (:215,:234)
    removed and the clinical input width (:291-292 returns 24) made a parameter."""

    def __init__(self, num_classes=2, clinical_in=16, modal_dim=256,
                 image_dim=None, signal_dim=None, clinical_dim=None):
        super().__init__()
        self.modal_dim = modal_dim
        self.image_dim = image_dim or modal_dim
        self.signal_dim = signal_dim or modal_dim
        self.clinical_dim = clinical_dim or modal_dim
        self.image_encoder = ResNet18()
        self.image_encoder.fc = nn.Linear(512, self.image_dim)
        self.image_norm = nn.LayerNorm(self.image_dim)
        self.signal_encoder = ResNet1D_SE(input_channels=1, num_classes=self.signal_dim)
        self.signal_norm = nn.LayerNorm(self.signal_dim)
        self.clinical_encoder = nn.Sequential(
            nn.Linear(clinical_in, 64), nn.BatchNorm1d(64), nn.ReLU(), nn.Dropout(0.3),
            nn.Linear(64, self.clinical_dim),
        )
        self.clinical_norm = nn.LayerNorm(self.clinical_dim)
        self.image_classifier = nn.Linear(self.image_dim, num_classes)
        self.signal_classifier = nn.Linear(self.signal_dim, num_classes)
        self.clinical_classifier = nn.Linear(self.clinical_dim, num_classes)
        self.attention_fusion = AttentionFusion([self.image_dim, self.signal_dim, self.clinical_dim])
        self.fusion_classifier = nn.Sequential(
            nn.Linear(self.image_dim + self.signal_dim + self.clinical_dim, 128),
            nn.ReLU(), nn.Dropout(0.3), nn.Linear(128, num_classes),
        )

    def forward(self, image, ecg_signal, clinical):
        img = self.image_norm(self.image_encoder(image))
        sig = self.signal_norm(self.signal_encoder(ecg_signal.unsqueeze(1)))
        clin = self.clinical_norm(self.clinical_encoder(clinical))
        img_logits = self.image_classifier(img)
        sig_logits = self.signal_classifier(sig)
        clin_logits = self.clinical_classifier(clin)
        fused, w = self.attention_fusion(img, sig, clin)
        fusion_logits = self.fusion_classifier(fused)
        vi = torch.var(img, dim=1).mean()
        vs = torch.var(sig, dim=1).mean()
        vc = torch.var(clin, dim=1).mean()
        var_loss = torch.abs(vi - vs) + torch.abs(vi - vc) + torch.abs(vs - vc)
        return img_logits, sig_logits, clin_logits, fusion_logits, var_loss, w


def disable_dropout(model):
    """Parity runs neutralise dropout (torch's CPU Philox stream cannot be matched on device)."""
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.0
    return model


def multimodal_loss(outputs, labels):
    """train.py:69-78: total = CE(fusion_logits) + 0.1 * var_loss."""
    return F.cross_entropy(outputs[3], labels) + 0.1 * outputs[4]


def freeze_encoders(model):
    """train.py:35-40."""
    for enc in (model.image_encoder, model.signal_encoder, model.clinical_encoder):
        for p in enc.parameters():
            p.requires_grad = False
    return model
