"""AM-softmax head with the reference's interface (scripts/loss.py:5-52).

    AMSoftmax(in_feats, n_classes, m=0.3, s=15, annealing=False)
        .forward(x, label, step) -> (costh [B,n], s * margin-logits [B,n])
        .getAnnealedFactor(step)

Maths as upstream: cos(theta) between L2-normalised rows of x and L2-normalised columns of W (norms clamped at 1e-12),
margin m subtracted in the label column, optional annealing blend
    logits = s * ((cos - m*onehot) + alpha*cos) / (1 + alpha),   alpha = 1000 / (1 + 1e-4*step)^2  (0 when not annealing)
but everything stays on the device: upstream moves the labels to the CPU, scatters the one-hot there and copies it back
(loss.py:46-48) — a host synchronisation on every step.
"""
import torch
from torch import nn

__all__ = ["AMSoftmax", "FocalSoftmax"]

_EPS = 1e-12


def _annealing_alpha(step):
    return max(0, 1000.0 / pow(1.0 + 0.0001 * float(step), 2.0))


class AMSoftmax(nn.Module):
    def __init__(self, in_feats, n_classes, m=0.3, s=15, annealing=False):
        super().__init__()
        self.in_feats, self.m, self.s, self.annealing = in_feats, m, s, annealing
        weight = torch.empty(in_feats, n_classes)
        nn.init.xavier_normal_(weight, gain=1)
        self.W = nn.Parameter(weight)

    def getAnnealedFactor(self, step):
        return 1 / (1 + (_annealing_alpha(step) if self.annealing else 0.0))

    def forward(self, x, label=None, step=0):
        assert x.size(0) == label.size(0)
        assert x.size(1) == self.in_feats
        rows = x / x.norm(p=2, dim=1, keepdim=True).clamp(min=_EPS)
        cols = self.W / self.W.norm(p=2, dim=0, keepdim=True).clamp(min=_EPS)
        costh = rows @ cols.to(rows.dtype)
        with_margin = costh - self.m * torch.nn.functional.one_hot(label, costh.size(1)).to(costh.dtype)
        alpha = _annealing_alpha(step) if self.annealing else 0.0
        return costh, self.s * (with_margin + alpha * costh) / (1 + alpha)


class FocalSoftmax(nn.Module):
    """Kept only so that `from loss import *` exports the same names (loss.py:54-70 is never used upstream)."""

    def __init__(self, gamma=2):
        super().__init__()
        self.gamma = gamma
        self.ce = nn.CrossEntropyLoss()

    def forward(self, input, target):
        nll = self.ce(input, target)
        return (nll * (1 - torch.exp(-nll)) ** self.gamma).mean()
