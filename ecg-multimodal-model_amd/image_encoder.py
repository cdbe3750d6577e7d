"""ResNet18 image encoder (the role torchvision.models.resnet18 plays in the reference:
multimodal_paper_modal_balance.py:210,221; train_image_only.py:92-99).

Same module tree / state_dict keys as torchvision's (conv1, bn1, layer{1..4}.{0,1}.{conv1,bn1,conv2,
bn2,downsample.{0,1}}, fc) so ``image_encoder.*`` checkpoints of the reference load unchanged.  The
forward and backward run as one native launch plan of HIP kernels (csrc/plan_resnet18.hip).
"""
import torch
import torch.nn as nn

from .hip import encoders as E
from .hip import nn as hnn


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1):
        super().__init__()
        self.conv1 = hnn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = hnn.BatchNorm2d(planes)
        self.conv2 = hnn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = hnn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(hnn.Conv2d(inplanes, planes, 1, stride, 0, bias=False),
                                            hnn.BatchNorm2d(planes))

    def forward(self, x):
        raise RuntimeError("BasicBlock is executed by ResNet18's fused launch plan; call the encoder")


class ResNet18(nn.Module):
    def __init__(self, num_classes=1000, compute_dtype="bf16"):
        super().__init__()
        self.compute_dtype = compute_dtype
        self.conv1 = hnn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = hnn.BatchNorm2d(64)
        self.layer1 = nn.Sequential(BasicBlock(64, 64), BasicBlock(64, 64))
        self.layer2 = nn.Sequential(BasicBlock(64, 128, 2), BasicBlock(128, 128))
        self.layer3 = nn.Sequential(BasicBlock(128, 256, 2), BasicBlock(256, 256))
        self.layer4 = nn.Sequential(BasicBlock(256, 512, 2), BasicBlock(512, 512))
        self.fc = hnn.Linear(512, num_classes)
        # torchvision's init: kaiming-normal(fan_out) convs, BN gamma=1 beta=0
        for m in self.modules():
            if isinstance(m, hnn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        self._spec = E.ResNet18Spec()

    def forward(self, x):
        spec = self._spec
        spec.dtype = E.dtype_code(self.compute_dtype)
        spec.training = self.training
        spec.momentum, spec.eps = self.bn1.momentum, self.bn1.eps
        spec.out_dim = self.fc.weight.shape[0]
        spec.buffers = list(self.buffers())
        params = list(self.parameters())
        if len(params) != 62 or len(spec.buffers) != 60:
            raise RuntimeError(f"ResNet18 expects 62 parameters / 60 buffers, found {len(params)} / {len(spec.buffers)}")
        return E.run_plan(x, spec, params)


def resnet18(num_classes=1000, compute_dtype="bf16", **_ignored):
    """Drop-in for ``torchvision.models.resnet18()`` (random init; a ``weights=`` download is not
    available offline -- load a local state_dict instead)."""
    return ResNet18(num_classes=num_classes, compute_dtype=compute_dtype)
