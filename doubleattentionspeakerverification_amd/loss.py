"""AM-softmax head with the reference's interface (scripts/loss.py:5-52).

    AMSoftmax(in_feats, n_classes, m=0.3, s=15, annealing=False)
        .forward(x, label, step) -> (costh [B,n], s * margin-logits [B,n])
        .getAnnealedFactor(step)

Maths as upstream: cos(theta) between L2-normalised rows of x and L2-normalised columns of W (norms clamped at 1e-12),
margin m subtracted in the label column, optional annealing blend
    logits = s * ((cos - m*onehot) + alpha*cos) / (1 + alpha),   alpha = 1000 / (1 + 1e-4*step)^2  (0 when not annealing)
but everything stays on the device: upstream moves the labels to the CPU, scatters the one-hot there and copies it back
(loss.py:46-48) — a host synchronisation on every step.

On the GPU (fp32) the head runs on hand-written kernels (csrc/amsoftmax.hip): `forward` is one launch (normalisations, product,
margin, scale — no one-hot), and `fused_loss` adds the cross-entropy of train.py:219 to the same pass (row log-sum-exps from
per-tile partials: no [B,N] softmax, no separate log-softmax), with a backward of four small launches.  CPU tensors (the
world_size-2 gloo tests, the CPU oracle's callers) take the torch ops below — the same maths.
"""
import torch
from torch import nn

from . import _lib

__all__ = ["AMSoftmax", "FocalSoftmax"]

_EPS = 1e-12


def _annealing_alpha(step):
    return max(0, 1000.0 / pow(1.0 + 0.0001 * float(step), 2.0))


class _AMSoftmaxFn(torch.autograd.Function):
    """csrc/amsoftmax.hip.  fused=False: (costh, logits) like AMSoftmax.forward; fused=True: (loss, costh)."""

    @staticmethod
    def forward(ctx, x, W, label, m, s, alpha, fused):
        lib = _lib.load()
        B, E = x.shape
        N = W.shape[1]
        x, Wc = x.contiguous(), W.detach().contiguous()
        label = label.to(device=x.device, dtype=torch.int64).contiguous()
        f32 = dict(dtype=torch.float32, device=x.device)
        costh = torch.empty(B, N, **f32)
        logits = None if fused else torch.empty(B, N, **f32)
        loss = torch.empty(1, **f32) if fused else None
        lse = torch.empty(B, **f32) if fused else None
        xinv, winv = torch.empty(B, **f32), torch.empty(N, **f32)
        nws = lib.dasv_amsoftmax_workspace_bytes(B, E, N)
        ws = torch.empty(nws, dtype=torch.uint8, device=x.device)
        with torch.cuda.device(x.device):
            rc = lib.dasv_amsoftmax_fwd(x.data_ptr(), Wc.data_ptr(), label.data_ptr(), costh.data_ptr(),
                                        None if logits is None else logits.data_ptr(), None if loss is None else loss.data_ptr(),
                                        None if lse is None else lse.data_ptr(), xinv.data_ptr(), winv.data_ptr(), ws.data_ptr(), nws,
                                        B, E, N, float(m), float(s), float(alpha), torch.cuda.current_stream(x.device).cuda_stream)
        _lib.check(rc, "dasv_amsoftmax_fwd")
        ctx.save_for_backward(x, Wc, label, costh, lse, xinv, winv)
        ctx.cfg = (float(m), float(s), float(alpha), fused)
        if fused:
            ctx.mark_non_differentiable(costh)      # accuracy only: the loss carries the gradient
            return loss.view(()), costh
        return costh, logits

    @staticmethod
    def backward(ctx, g0, g1):
        x, Wc, label, costh, lse, xinv, winv = ctx.saved_tensors
        m, s, alpha, fused = ctx.cfg
        lib = _lib.load()
        B, E = x.shape
        N = Wc.shape[1]
        gx, gW = torch.empty_like(x), torch.empty_like(Wc)
        nws = lib.dasv_amsoftmax_workspace_bytes(B, E, N)
        ws = torch.empty(nws, dtype=torch.uint8, device=x.device)
        if fused:
            gl = None if g0 is None else g0.reshape(1).float().contiguous()
            args = (lse.data_ptr(), xinv.data_ptr(), winv.data_ptr(), None if gl is None else gl.data_ptr(), None, None)
        else:
            gc = None if g0 is None else g0.float().contiguous()
            gg = None if g1 is None else g1.float().contiguous()
            if gc is None and gg is None:
                return None, None, None, None, None, None, None
            args = (None, xinv.data_ptr(), winv.data_ptr(), None, None if gc is None else gc.data_ptr(), None if gg is None else gg.data_ptr())
        with torch.cuda.device(x.device):
            rc = lib.dasv_amsoftmax_bwd(x.data_ptr(), Wc.data_ptr(), label.data_ptr(), costh.data_ptr(), *args, gx.data_ptr(), gW.data_ptr(),
                                        ws.data_ptr(), nws, B, E, N, m, s, alpha, torch.cuda.current_stream(x.device).cuda_stream)
        _lib.check(rc, "dasv_amsoftmax_bwd")
        return gx, gW, None, None, None, None, None


class AMSoftmax(nn.Module):
    def __init__(self, in_feats, n_classes, m=0.3, s=15, annealing=False):
        super().__init__()
        self.in_feats, self.m, self.s, self.annealing = in_feats, m, s, annealing
        weight = torch.empty(in_feats, n_classes)
        nn.init.xavier_normal_(weight, gain=1)
        self.W = nn.Parameter(weight)

    def getAnnealedFactor(self, step):
        return 1 / (1 + (_annealing_alpha(step) if self.annealing else 0.0))

    def _alpha(self, step):
        return _annealing_alpha(step) if self.annealing else 0.0

    def _hip(self, x):
        return x.is_cuda and x.dtype == torch.float32 and self.W.dtype == torch.float32 and not torch.is_autocast_enabled()

    def fused_loss(self, x, label, step=0):
        """(mean cross-entropy of the scaled margin logits, costh) — loss.py:37-52 + train.py:219 in one pass; the [B,N] logits
        are never materialised.  Same value and gradients as CrossEntropyLoss()(self(x, label, step)[1], label)."""
        assert x.size(0) == label.size(0)
        assert x.size(1) == self.in_feats
        if not self._hip(x):
            costh, logits = self.forward(x, label, step)
            return torch.nn.functional.cross_entropy(logits.float(), label), costh
        return _AMSoftmaxFn.apply(x, self.W, label, self.m, self.s, self._alpha(step), True)

    def forward(self, x, label=None, step=0):
        assert x.size(0) == label.size(0)
        assert x.size(1) == self.in_feats
        if self._hip(x):
            return _AMSoftmaxFn.apply(x, self.W, label, self.m, self.s, self._alpha(step), False)
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
