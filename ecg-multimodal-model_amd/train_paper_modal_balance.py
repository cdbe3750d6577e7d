"""Entry point of the paper_modal_balance variant (reference: train_paper_modal_balance.py:13,20-324): the same
loop as ``train.main`` on ``multimodal_paper_modal_balance.ECGMultimodalModel`` (clinical MLP, all branches 256
wide), with every parameter trainable (``Adam(model.parameters())``, train_paper_modal_balance.py:29 -- no freeze)."""
from . import train as _train
from .config import Config
from .multimodal_paper_modal_balance import ECGMultimodalModel


def main(config=Config, num_epochs=None, quiet=False, freeze_encoders=False):
    return _train.main(config, freeze_encoders=freeze_encoders, num_epochs=num_epochs, quiet=quiet,
                       model_cls=ECGMultimodalModel)


if __name__ == "__main__":
    main()
