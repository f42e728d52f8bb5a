"""SpeakerClassifier with the reference's interface (scripts/model.py:8-71).

    SpeakerClassifier(parameters: argparse.Namespace, device)
        .getEmbedding(x [B,T,80]) -> [B, embedding_size]                     model.py:52-59
        .forward(x, label=None, step=0) -> (costh, scaled margin logits)     model.py:61-71

Reads from ``parameters``: front_end, kernel_size, pooling_method, heads_number, mask_prob,
embedding_size, num_spkrs, scalingFactor, marginFactor, annealing; writes feature_size = 80
(model.py:13).  Sub-module names (front_end, poolingLayer, fc1, b1, fc2, b2, preLayer, b3,
predictionLayer) and therefore state_dict keys equal the reference's; b1 and b3 exist but are
unused there too (model.py:46,50).
"""
import argparse

import torch
from torch import nn
from torch.nn import functional as F

from .poolings import *  # noqa: F401,F403  (callers rely on `from model import *`)
from .CNNs import *  # noqa: F401,F403
from .loss import *  # noqa: F401,F403
from .poolings import Attention, DoubleMHA, MultiHeadAttention
from .CNNs import VGG3L, VGG4L, getVGG3LOutputDimension, getVGG4LOutputDimension
from .loss import AMSoftmax

# getEmbeddingExample.py:27 calls torch.load() on a checkpoint holding an argparse.Namespace;
# torch >= 2.6 defaults to weights_only=True, which needs the class allow-listed.
try:
    torch.serialization.add_safe_globals([argparse.Namespace])
except AttributeError:  # older torch: nothing to do
    pass


class SpeakerClassifier(nn.Module):
    def __init__(self, parameters, device):
        super().__init__()
        parameters.feature_size = 80
        self.device = device
        self._init_front_end(parameters)
        self._init_pooling(parameters)
        self._init_fc(parameters)
        self.predictionLayer = AMSoftmax(parameters.embedding_size, parameters.num_spkrs,
                                         s=parameters.scalingFactor, m=parameters.marginFactor,
                                         annealing=parameters.annealing)

    def _init_front_end(self, p):
        if p.front_end == "VGG3L":
            self.vector_size = getVGG3LOutputDimension(p.feature_size, outputChannel=p.kernel_size)
            self.front_end = VGG3L(p.kernel_size)
        if p.front_end == "VGG4L":
            self.vector_size = getVGG4LOutputDimension(p.feature_size, outputChannel=p.kernel_size)
            self.front_end = VGG4L(p.kernel_size)

    def _init_pooling(self, p):
        self.pooling_method = p.pooling_method
        if self.pooling_method == "Attention":
            self.poolingLayer = Attention(self.vector_size)
        elif self.pooling_method == "MHA":
            self.poolingLayer = MultiHeadAttention(self.vector_size, p.heads_number)
        elif self.pooling_method == "DoubleMHA":
            self.poolingLayer = DoubleMHA(self.vector_size, p.heads_number, mask_prob=p.mask_prob)
            self.vector_size = self.vector_size // p.heads_number

    def _init_fc(self, p):
        self.fc1 = nn.Linear(self.vector_size, p.embedding_size)
        self.b1 = nn.BatchNorm1d(p.embedding_size)
        self.fc2 = nn.Linear(p.embedding_size, p.embedding_size)
        self.b2 = nn.BatchNorm1d(p.embedding_size)
        self.preLayer = nn.Linear(p.embedding_size, p.embedding_size)
        self.b3 = nn.BatchNorm1d(p.embedding_size)

    def _pooled(self, x):
        enc = self.front_end(x)
        if isinstance(self.poolingLayer, DoubleMHA):
            return self.poolingLayer.pool(enc)  # same launch, alignments not materialised
        return self.poolingLayer(enc)[0]

    def getEmbedding(self, x):
        e1 = F.relu(self.fc1(self._pooled(x)))
        return self.b2(F.relu(self.fc2(e1)))

    def forward(self, x, label=None, step=0):
        e3 = self.preLayer(self.getEmbedding(x))
        return self.predictionLayer(e3, label, step)

__all__ = ["SpeakerClassifier", "new_parameter", "Attention", "HeadAttention", "innerKeyValueAttention",
           "MultiHeadAttention", "DoubleMHA", "getVGG3LOutputDimension", "getVGG4LOutputDimension",
           "VGG3L", "VGG4L", "AMSoftmax", "FocalSoftmax"]
