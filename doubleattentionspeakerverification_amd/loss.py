"""AM-softmax head with the reference's interface (scripts/loss.py:5-52).

    AMSoftmax(in_feats, n_classes, m=0.3, s=15, annealing=False)
        .forward(x, label, step) -> (costh [B,n], s*margin-logits [B,n])
        .getAnnealedFactor(step)

Same maths (row-normalised x times column-normalised W, margin m subtracted at the label
column, optional annealing blend, scale s) but the one-hot margin is built on the device:
the reference's label -> CPU -> scatter -> GPU round trip (loss.py:46-48), a host
synchronisation every step, is gone.
"""
import torch
from torch import nn

__all__ = ["AMSoftmax", "FocalSoftmax"]


class AMSoftmax(nn.Module):
    def __init__(self, in_feats, n_classes, m=0.3, s=15, annealing=False):
        super().__init__()
        self.in_feats = in_feats
        self.m = m
        self.s = s
        self.annealing = annealing
        self.W = nn.Parameter(torch.randn(in_feats, n_classes), requires_grad=True)
        nn.init.xavier_normal_(self.W, gain=1)

    @staticmethod
    def _alpha(step):
        return max(0, 1000.0 / (pow(1.0 + 0.0001 * float(step), 2.0)))

    def getAnnealedFactor(self, step):
        alpha = self._alpha(step) if self.annealing else 0.0
        return 1 / (1 + alpha)

    def forward(self, x, label=None, step=0):
        assert x.size()[0] == label.size()[0]
        assert x.size()[1] == self.in_feats
        xn = x / torch.norm(x, p=2, dim=1, keepdim=True).clamp(min=1e-12)
        wn = self.W / torch.norm(self.W, p=2, dim=0, keepdim=True).clamp(min=1e-12)
        costh = torch.mm(xn, wn.to(xn.dtype))
        margin = torch.zeros_like(costh).scatter_(1, label.view(-1, 1), self.m)  # on the device
        alpha = self._alpha(step) if self.annealing else 0.0
        combined = ((costh - margin) + alpha * costh) / (1 + alpha)
        return costh, self.s * combined


class FocalSoftmax(nn.Module):
    """Name kept for star-import compatibility (loss.py:54-70); never used by the reference."""

    def __init__(self, gamma=2):
        super().__init__()
        self.gamma = gamma
        self.ce = nn.CrossEntropyLoss()

    def forward(self, input, target):
        logp = self.ce(input, target)
        return ((1 - torch.exp(-logp)) ** self.gamma * logp).mean()
