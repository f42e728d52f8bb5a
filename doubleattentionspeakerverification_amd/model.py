"""SpeakerClassifier with the reference's interface (scripts/model.py:8-71).

    SpeakerClassifier(parameters: argparse.Namespace, device)
        .getEmbedding(x [B,T,80]) -> [B, embedding_size]                     model.py:52-59
        .forward(x, label=None, step=0) -> (costh, scaled margin logits)     model.py:61-71
    x may also be a LIST of [T_i, 80] utterances of different lengths (a variable-length minibatch, BASELINE.json north_star):
    every utterance is then processed at its own length — what the reference can only do one per forward — and the dense tail
    sees them as one batch (forward, backward and BatchNorm statistics included).

Reads from ``parameters``: front_end, kernel_size, pooling_method, heads_number, mask_prob, embedding_size, num_spkrs,
scalingFactor, marginFactor, annealing; writes feature_size = 80 (model.py:13).  The sub-module names — front_end,
poolingLayer, fc1, b1, fc2, b2, preLayer, b3, predictionLayer — are the drop-in contract (state_dict keys of reference
checkpoints); b1 and b3 are registered but never called, exactly like upstream (model.py:46,50), so they carry no gradient.

Data flow on the GPU: NHWC fp32 front-end on the hand-written conv / pool kernels -> ONE fused DoubleMHA launch (the
[B,T,H] alignments are not materialised on this path) -> two small dense layers and BatchNorm through PyTorch-ROCm.
"""
import argparse

import torch
from torch import nn
from torch.nn import functional as F

from . import CNNs as _cnns
from . import loss as _loss
from . import poolings as _pool
from .CNNs import *  # noqa: F401,F403  (callers rely on `from model import *` re-exporting everything)
from .loss import *  # noqa: F401,F403
from .poolings import *  # noqa: F401,F403

# getEmbeddingExample.py:27 calls torch.load() on a checkpoint holding an argparse.Namespace; torch >= 2.6 defaults to
# weights_only=True, which needs the class allow-listed.
if hasattr(torch.serialization, "add_safe_globals"):
    torch.serialization.add_safe_globals([argparse.Namespace])

MEL_BINS = 80

# front_end name -> (module class, output-dimension helper)
_FRONT_ENDS = {
    "VGG3L": (_cnns.VGG3L, _cnns.getVGG3LOutputDimension),
    "VGG4L": (_cnns.VGG4L, _cnns.getVGG4LOutputDimension),
}


def _make_pooling(method, width, heads, mask_prob):
    """-> (pooling module, width of the pooled vector)."""
    if method == "Attention":
        return _pool.Attention(width), width
    if method == "MHA":
        return _pool.MultiHeadAttention(width, heads), width
    if method == "DoubleMHA":
        return _pool.DoubleMHA(width, heads, mask_prob=mask_prob), width // heads
    raise ValueError("unknown pooling_method %r" % (method,))


class SpeakerClassifier(nn.Module):
    def __init__(self, parameters, device):
        super().__init__()
        parameters.feature_size = MEL_BINS
        self.device = device
        self.pooling_method = parameters.pooling_method
        emb = parameters.embedding_size

        front_cls, out_dim = _FRONT_ENDS[parameters.front_end]
        self.front_end = front_cls(parameters.kernel_size)
        frame_width = out_dim(parameters.feature_size, outputChannel=parameters.kernel_size)
        self.poolingLayer, self.vector_size = _make_pooling(parameters.pooling_method, frame_width,
                                                            parameters.heads_number, parameters.mask_prob)
        # registration order below fixes the state_dict key order of the reference: fc1, b1, fc2, b2, preLayer, b3
        for name, layer in (("fc1", nn.Linear(self.vector_size, emb)), ("b1", nn.BatchNorm1d(emb)),
                            ("fc2", nn.Linear(emb, emb)), ("b2", nn.BatchNorm1d(emb)),
                            ("preLayer", nn.Linear(emb, emb)), ("b3", nn.BatchNorm1d(emb))):
            self.add_module(name, layer)
        self.predictionLayer = _loss.AMSoftmax(emb, parameters.num_spkrs, s=parameters.scalingFactor,
                                               m=parameters.marginFactor, annealing=parameters.annealing)
        # variable-length TRAINING minibatches (a list of utterances): "stacked" = one set of launches with border tables where the
        # front-end's kernels take them (split8, exact channel counts), else / "buckets" = one dense pass per distinct length
        self.ragged_training = "stacked"

    def _utterance_vector(self, x):
        pool = self.poolingLayer
        if isinstance(pool, _pool.DoubleMHA):
            fmap = self.front_end.forward_map(x)
            if _pool.nhwc_frames_supported(fmap, pool.heads_number):
                return pool.pool_map(fmap)     # the fused launch reads the NHWC map in place: no transposing copy
            frames = fmap.permute(0, 2, 1, 3).reshape(fmap.size(0), fmap.size(2), fmap.size(1) * fmap.size(3))
            return pool.pool(frames)           # same launch as forward(), alignments skipped
        return pool(self.front_end(x))[0]

    def _utterance_vectors_ragged(self, utterances):
        """Pooled vectors [B, vector_size] of a minibatch of utterances of DIFFERENT lengths (list of [T_i, 80] tensors), in input
        order, differentiable.  The reference has no padding and no masks anywhere in the front-end or the pooling, so an
        utterance must be processed at its own length (train.py:117-133 does one per forward): utterances of EQUAL length share one
        dense pass (batching.bucket_by_length), and the pooled vectors are put back in order — everything behind the pooling
        (fc1, fc2, BatchNorm b2 on the statistics of the WHOLE minibatch, the AM-softmax head) then sees one [B, .] batch exactly as
        if all B utterances had gone through one forward."""
        from .batching import bucket_by_length
        lengths = [int(u.shape[0]) for u in utterances]
        dev0 = next(self.parameters()).device
        if (self.ragged_training == "stacked" and self.training and torch.is_grad_enabled() and dev0.type == "cuda"
                and len(set(lengths)) > 1 and self.front_end.ragged_training_ok()):
            # ONE set of launches for the whole minibatch, forward and backward: the utterances stacked along time, borders from
            # tables (ragged.utterance_vectors_ragged_train)
            from .ragged import utterance_vectors_ragged_train
            self.last_ragged_buckets = None
            return utterance_vectors_ragged_train(self, utterances)
        buckets = bucket_by_length(lengths, max_batch=len(utterances))
        dev = next(self.parameters()).device
        parts, order = [], []
        for b in buckets:
            x = torch.stack([utterances[i] for i in b]).to(dev, non_blocking=True).float()
            parts.append(self._utterance_vector(x))
            order += b
        self.last_ragged_buckets = buckets          # (tests read which utterances shared a pass)
        inv = torch.empty(len(order), dtype=torch.long)
        inv[torch.tensor(order)] = torch.arange(len(order))
        return torch.cat(parts, dim=0).index_select(0, inv.to(dev))

    def getEmbedding(self, x):
        vec = self._utterance_vectors_ragged(x) if isinstance(x, (list, tuple)) else self._utterance_vector(x)
        hidden = F.relu(self.fc1(vec))
        return self.b2(F.relu(self.fc2(hidden)))

    def forward(self, x, label=None, step=0):
        return self.predictionLayer(self.preLayer(self.getEmbedding(x)), label, step)

    def forward_loss(self, x, label, step=0):
        """(loss, costh): forward() followed by train.py:219's CrossEntropyLoss in one fused head (loss.AMSoftmax.fused_loss)."""
        return self.predictionLayer.fused_loss(self.preLayer(self.getEmbedding(x)).float(), label, step)


__all__ = ["SpeakerClassifier", "new_parameter", "Attention", "HeadAttention", "innerKeyValueAttention",
           "MultiHeadAttention", "DoubleMHA", "getVGG3LOutputDimension", "getVGG4LOutputDimension",
           "VGG3L", "VGG4L", "AMSoftmax", "FocalSoftmax"]
