"""fusion_classifier.py:5-11 -- wraps ``model.fusion_classifier`` so SHAP-style tools can call it."""
import torch.nn as nn


class FusionClassifierWrapper(nn.Module):
    def __init__(self, fusion_classifier):
        super().__init__()
        self.fusion_classifier = fusion_classifier

    def forward(self, x):
        return self.fusion_classifier(x)
