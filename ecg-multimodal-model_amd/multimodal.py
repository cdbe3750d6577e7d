"""``multimodal.py`` is what the reference's train.py:14 imports.  Its ``ECGMultimodalModel`` differs
from the paper_modal_balance variant only in the clinical branch (a third-party TabNet,
multimodal.py:109-148, source and version absent -> SURVEY 8f "next") and in the dims 512/128/32.
Until the TabNet encoder is built, this module exposes the MLP-clinical model under the same names so
``from multimodal import ECGMultimodalModel`` keeps working."""
from .multimodal_paper_modal_balance import (AttentionFusion, BasicBlock1D, ECGMultimodalModel,  # noqa: F401
                                             MultimodalModel, ResNet1D_SE, SEBlock)
