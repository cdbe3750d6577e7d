"""ORACLE (test infrastructure only): torch-CPU restatement of ``pytorch_tabnet.tab_network.TabNetNoEmbeddings`` as the
reference instantiates it (multimodal.py:113-123: input_dim=2, output_dim=n_d=n_a=32, n_steps=3, gamma=1.5,
n_independent=2, n_shared=2; library defaults virtual_batch_size=128, momentum=0.02, epsilon=1e-15, sparsemax mask).

PARITY UNPINNED: pytorch_tabnet is a third-party dependency that is neither vendored in /root/reference nor installed
here, and the reference pins no version (README.md:67) and holds no test or golden for it.  This file restates the
published algorithm (Arik & Pfister 2019; dreamquark-ai/tabnet ``tab_network.py`` / ``sparsemax.py``, v3-v4 layout:
module and parameter names are kept so that a checkpoint of that library would load):
  initial BatchNorm1d(momentum 0.01) -> FeatTransformer (shared GLU block of n_shared layers + independent block) ->
  per step: AttentiveTransformer (Linear -> GhostBN -> * prior -> sparsemax) = mask M; prior *= (gamma - M);
  masked input -> FeatTransformer -> ReLU(first n_d) summed over steps -> final Linear; the second half feeds the
  next step's attention; M_loss = mean over steps of mean_batch sum_features M log(M + eps).
GLU layer = Linear(no bias, 2*out) -> GhostBN -> a * sigmoid(b); blocks add residuals scaled by sqrt(0.5) (the first
layer of a "first" block has no residual).
"""
import math

import numpy as np
import torch
import torch.nn as nn


def sparsemax(x):
    x = x - x.max(dim=-1, keepdim=True).values
    srt, _ = torch.sort(x, dim=-1, descending=True)
    cum = srt.cumsum(-1) - 1
    rho = torch.arange(1, x.shape[-1] + 1, dtype=x.dtype).view(*([1] * (x.dim() - 1)), -1)
    support = rho * srt > cum
    k = support.sum(-1, keepdim=True)
    tau = cum.gather(-1, k - 1) / k.to(x.dtype)
    return torch.clamp(x - tau, min=0)


def initialize_non_glu(module, input_dim, output_dim):
    nn.init.xavier_normal_(module.weight, gain=np.sqrt((input_dim + output_dim) / np.sqrt(4 * input_dim)))


def initialize_glu(module, input_dim, output_dim):
    nn.init.xavier_normal_(module.weight, gain=np.sqrt((input_dim + output_dim) / np.sqrt(input_dim)))


class GBN(nn.Module):
    def __init__(self, input_dim, virtual_batch_size=128, momentum=0.01):
        super().__init__()
        self.virtual_batch_size = virtual_batch_size
        self.bn = nn.BatchNorm1d(input_dim, momentum=momentum)

    def forward(self, x):
        chunks = x.chunk(int(math.ceil(x.shape[0] / self.virtual_batch_size)), 0)
        return torch.cat([self.bn(c) for c in chunks], dim=0)


class GLU_Layer(nn.Module):
    def __init__(self, input_dim, output_dim, fc=None, virtual_batch_size=128, momentum=0.02):
        super().__init__()
        self.output_dim = output_dim
        self.fc = fc if fc is not None else nn.Linear(input_dim, 2 * output_dim, bias=False)
        initialize_glu(self.fc, input_dim, 2 * output_dim)
        self.bn = GBN(2 * output_dim, virtual_batch_size, momentum)

    def forward(self, x):
        x = self.bn(self.fc(x))
        return x[:, :self.output_dim] * torch.sigmoid(x[:, self.output_dim:])


class GLU_Block(nn.Module):
    def __init__(self, input_dim, output_dim, n_glu=2, first=False, shared_layers=None, virtual_batch_size=128,
                 momentum=0.02):
        super().__init__()
        self.first, self.n_glu = first, n_glu
        self.glu_layers = nn.ModuleList()
        for i in range(n_glu):
            fc = shared_layers[i] if shared_layers else None
            self.glu_layers.append(GLU_Layer(input_dim if i == 0 else output_dim, output_dim, fc, virtual_batch_size,
                                             momentum))

    def forward(self, x):
        scale = math.sqrt(0.5)
        start = 0
        if self.first:
            x = self.glu_layers[0](x)
            start = 1
        for i in range(start, self.n_glu):
            x = (x + self.glu_layers[i](x)) * scale
        return x


class FeatTransformer(nn.Module):
    def __init__(self, input_dim, output_dim, shared_layers, n_glu_independent, virtual_batch_size=128, momentum=0.02):
        super().__init__()
        if shared_layers is None:
            self.shared, is_first = nn.Identity(), True
        else:
            self.shared = GLU_Block(input_dim, output_dim, len(shared_layers), True, shared_layers, virtual_batch_size,
                                    momentum)
            is_first = False
        if n_glu_independent == 0:
            self.specifics = nn.Identity()
        else:
            self.specifics = GLU_Block(input_dim if is_first else output_dim, output_dim, n_glu_independent, is_first,
                                       None, virtual_batch_size, momentum)

    def forward(self, x):
        return self.specifics(self.shared(x))


class AttentiveTransformer(nn.Module):
    def __init__(self, input_dim, group_dim, virtual_batch_size=128, momentum=0.02):
        super().__init__()
        self.fc = nn.Linear(input_dim, group_dim, bias=False)
        initialize_non_glu(self.fc, input_dim, group_dim)
        self.bn = GBN(group_dim, virtual_batch_size, momentum)

    def forward(self, priors, processed_feat):
        return sparsemax(self.bn(self.fc(processed_feat)) * priors)


class TabNetEncoder(nn.Module):
    def __init__(self, input_dim, output_dim, n_d, n_a, n_steps, gamma, n_independent, n_shared, epsilon=1e-15,
                 virtual_batch_size=128, momentum=0.02):
        super().__init__()
        self.input_dim, self.n_d, self.n_a, self.n_steps, self.gamma, self.epsilon = input_dim, n_d, n_a, n_steps, gamma, epsilon
        self.initial_bn = nn.BatchNorm1d(input_dim, momentum=0.01)
        self.group_attention_matrix = torch.eye(input_dim)     # plain attribute in the library, not a buffer
        shared = None
        if n_shared > 0:
            shared = nn.ModuleList([nn.Linear(input_dim if i == 0 else n_d + n_a, 2 * (n_d + n_a), bias=False)
                                    for i in range(n_shared)])
        self.initial_splitter = FeatTransformer(input_dim, n_d + n_a, shared, n_independent, virtual_batch_size, momentum)
        self.feat_transformers = nn.ModuleList()
        self.att_transformers = nn.ModuleList()
        for _ in range(n_steps):
            self.feat_transformers.append(FeatTransformer(input_dim, n_d + n_a, shared, n_independent,
                                                          virtual_batch_size, momentum))
            self.att_transformers.append(AttentiveTransformer(n_a, input_dim, virtual_batch_size, momentum))

    def forward(self, x):
        x = self.initial_bn(x)
        prior = torch.ones(x.shape[0], self.input_dim, dtype=x.dtype)
        m_loss = 0
        att = self.initial_splitter(x)[:, self.n_d:]
        steps_output = []
        for step in range(self.n_steps):
            M = self.att_transformers[step](prior, att)
            m_loss = m_loss + torch.mean(torch.sum(M * torch.log(M + self.epsilon), dim=1))
            prior = (self.gamma - M) * prior
            out = self.feat_transformers[step](torch.matmul(M, self.group_attention_matrix) * x)
            steps_output.append(torch.relu(out[:, :self.n_d]))
            att = out[:, self.n_d:]
        return steps_output, m_loss / self.n_steps


class TabNetNoEmbeddings(nn.Module):
    def __init__(self, input_dim, output_dim, n_d=8, n_a=8, n_steps=3, gamma=1.3, n_independent=2, n_shared=2,
                 epsilon=1e-15, virtual_batch_size=128, momentum=0.02):
        super().__init__()
        self.encoder = TabNetEncoder(input_dim, output_dim, n_d, n_a, n_steps, gamma, n_independent, n_shared, epsilon,
                                     virtual_batch_size, momentum)
        self.final_mapping = nn.Linear(n_d, output_dim, bias=False)
        initialize_non_glu(self.final_mapping, n_d, output_dim)

    def forward(self, x):
        steps_output, m_loss = self.encoder(x)
        return self.final_mapping(torch.sum(torch.stack(steps_output, dim=0), dim=0)), m_loss


class ClinicalTabNetEncoder(nn.Module):
    """multimodal.py:109-148"""

    def __init__(self, input_dim, latent_dim=32):
        super().__init__()
        self.latent_dim = latent_dim
        self.tabnet = TabNetNoEmbeddings(input_dim, latent_dim, n_d=latent_dim, n_a=latent_dim, n_steps=3, gamma=1.5,
                                         n_independent=2, n_shared=2)

    def forward(self, x):
        return self.tabnet(x)


def multimodal_tabnet_model(num_classes=2):
    """multimodal.py:332-460's ECGMultimodalModel: widths 512 / 128 / 32 and the TabNet clinical branch (mask loss dropped)"""
    from . import ref_models as O

    class _Clin(nn.Module):
        def __init__(self):
            super().__init__()
            self.tabnet = TabNetNoEmbeddings(2, 32, n_d=32, n_a=32, n_steps=3, gamma=1.5, n_independent=2, n_shared=2)

        def forward(self, x):
            return self.tabnet(x)[0]

    m = O.ECGMultimodalModel(num_classes, clinical_in=2, image_dim=512, signal_dim=128, clinical_dim=32)
    m.clinical_encoder = _Clin()
    return m
