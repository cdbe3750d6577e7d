"""Global hyper-parameters / paths -- same attribute names and defaults as the reference's
``Config`` (config.py:6-46), plus the fields the MI355X build adds (marked NEW)."""
import os

import torch


class Config:
    # Random seed
    seed = 42

    # Image input
    img_height = 224
    img_width = 224

    # Data paths (the private hospital data is not part of either repo)
    data_dir = "./data"
    image_dir = os.path.join(data_dir, "images")
    ecg_csv = os.path.join(data_dir, "ecg_signals.csv")
    label_file = os.path.join(data_dir, "labels.xlsx")
    clinical_file = os.path.join(data_dir, "clinical.csv")
    af_label_file = os.path.join(data_dir, "af_labels.xlsx")
    arrhythmia_label_file = os.path.join(data_dir, "arrhythmia_labels.xlsx")
    physionet_dir = "./data/physionet"
    physionet_data_dir = "./data/physionet/training2017"
    physionet_label_file = os.path.join(physionet_dir, "REFERENCE.csv")

    # Model
    num_classes = 2

    # Training hyperparameters
    batch_size = 16
    num_epochs = 30
    lr = 1e-4
    patience = 5

    # CV settings
    k_outer = 5
    k_inner = 3

    checkpoint_dir = "./checkpoints"

    device = "cuda" if torch.cuda.is_available() else "cpu"

    # ---- NEW (no counterpart in the reference) -------------------------------------------------
    compute_dtype = "bf16"          # conv-trunk compute dtype: "bf16" (MFMA bf16) or "fp32" (exact f32 MFMA)
    modal_dim = 256                 # multimodal_paper_modal_balance.py:203
    clinical_input_dim = 24         # get_clinical_feature_dim(), multimodal_paper_modal_balance.py:291-292
    signal_length = 5000
    signal_leads = 1
    dropout = 0.3                   # nn.Dropout(0.3) at :115, :260, :287
    # the reference hard-loads these inside the model constructor (:215, :234-237); here optional
    pretrained_image_encoder = None
    pretrained_signal_encoder = None
    # synthetic data (stands in for ./data, which neither repo ships)
    synthetic = True
    synthetic_train_size = 256
    synthetic_val_size = 32
    synthetic_test_size = 32
    num_workers = 0
