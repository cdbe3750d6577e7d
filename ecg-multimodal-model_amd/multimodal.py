"""``multimodal.py`` is what the reference's train.py:14 imports.  Its ``ECGMultimodalModel`` (multimodal.py:332-460)
differs from the paper_modal_balance variant in the branch widths -- image 512, signal 128, clinical 32, fused
672 -> 128 -> num_classes -- and in the clinical branch: a TabNet encoder over the two scaled numeric columns
(``ClinicalTabNetEncoder``, multimodal.py:109-148; ``get_clinical_feature_dim() == 2``, :421-422) whose mask loss is
dropped in forward (:446).  Everything else (encoders, LayerNorms, AttentionFusion, branch classifiers, var_loss,
the 6-tuple) is shared with ``multimodal_paper_modal_balance``.
"""
from . import multimodal_paper_modal_balance as _pmb
from .multimodal_paper_modal_balance import AttentionFusion, BasicBlock1D, ResNet1D_SE, SEBlock  # noqa: F401
from .tabnet import ClinicalTabNetEncoder  # noqa: F401


class ECGMultimodalModel(_pmb.ECGMultimodalModel):
    def _branch_dims(self):
        return 512, 128, 32

    def get_clinical_feature_dim(self):
        return 2

    def _build_clinical_encoder(self, p):
        return ClinicalTabNetEncoder(input_dim=self.get_clinical_feature_dim(), latent_dim=32,
                                     device=getattr(self.config, "device", None))

    def _clinical_forward(self, clinical):
        clinical_feat, _m_loss = self.clinical_encoder(clinical)
        return clinical_feat

    def load_pretrained_clinical_encoder(self, weight_path):
        return self.clinical_encoder.load_pretrained_partial(weight_path)

    def extract_clinical_features(self, clinical):
        z, _ = self.clinical_encoder(clinical)
        return z


MultimodalModel = ECGMultimodalModel
