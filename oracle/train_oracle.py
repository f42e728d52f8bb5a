"""CPU restatement of one training step of the reference (test infrastructure, see oracle/__init__.py), in float64, with
the network's DECISIONS made explicit.

A ReLU network's gradient is a smooth function of its weights only as long as the decisions of its forward pass stay put: which
pre-activations are positive (CNNs.py:38-48 / 72-86, `F.relu`) and which element of every 2x2 window is the maximum
(`F.max_pool2d`, CNNs.py:40,44,48 / 74,78,82,86).  Two forwards that agree to 1e-5 can disagree on a handful of those decisions, and
ONE disagreement moves gradients by percents on small maps.  Gradient parity of a kernel path is therefore checked in two parts:
the decisions its forward took are compared with the float64 forward's own (and the disagreements COUNTED), and its gradients are
compared element by element with the float64 gradients computed UNDER THE SAME DECISIONS — that comparison has no flips by
construction, so the tight bound holds for every element.

Reference lines restated (all under /root/reference/scripts):
  CNNs.py:34-52, 68-91     VGG3L / VGG4L forward                       -> vgg_forward_decided
  model.py:52-71           getEmbedding / forward (train mode)        -> classifier_train_forward
  loss.py:37-52            AMSoftmax                                   -> dmha_oracle.amsoftmax_forward
  train.py:215-226         forward, CrossEntropyLoss, backward, summed accumulation -> train_step_grads

Pinned by fixture G7 (tests/test_oracle_golden.py::test_g7_train_step_gradients): the float64 step under its own decisions
reproduces the reference's losses and every gradient tensor.
"""
import torch
import torch.nn.functional as F

from .dmha_oracle import amsoftmax_forward, dmha_opseq

__all__ = ["vgg_forward_decided", "classifier_train_forward", "train_step_grads", "count_flips", "vgg_layer_names"]


def vgg_layer_names(n_blocks):
    """The decision points of a front-end in forward order: 'conv11', 'conv12', 'pool1', 'conv21', ..."""
    names = []
    for blk in range(1, n_blocks + 1):
        names += ["conv%d1" % blk, "conv%d2" % blk, "pool%d" % blk]
    return names


def vgg_forward_decided(x, sd, n_blocks, decisions=None, record=None, prefix="front_end."):
    """CNNs.py:34-52 / 68-91 from a state_dict: x [B,T,F] -> [B,T',C*F'], differentiable in whatever dtype x / sd carry.

    decisions (optional): {'convXY': bool [B,C,H,W] (the ReLU passes where True), 'poolX': int64 [B,C,Ho,Wo] (flat index h*W + w
    of the element each window hands on, the form F.max_pool2d(..., return_indices=True) gives)} — any subset; missing entries
    are decided by this forward itself.  record (optional dict): receives the decisions that were used, plus 'z:convXY' = the
    pre-activations (detached) so that a caller can see how close to zero a flipped one was."""
    h = x.view(x.size(0), x.size(1), 1, x.size(2)).transpose(1, 2)
    for blk in range(1, n_blocks + 1):
        for j in (1, 2):
            lname = "conv%d%d" % (blk, j)
            name = prefix + lname
            z = F.conv2d(h, sd[name + ".weight"], sd[name + ".bias"], stride=1, padding=1)
            m = decisions[lname] if decisions is not None and lname in decisions else (z.detach() > 0)
            if record is not None:
                record[lname] = m
                record["z:" + lname] = z.detach()
            h = z * m.to(z.dtype)
        pname = "pool%d" % blk
        if decisions is not None and pname in decisions:
            idx = decisions[pname]
        else:
            idx = F.max_pool2d(h.detach(), 2, stride=2, ceil_mode=True, return_indices=True)[1]
        if record is not None:
            record[pname] = idx
            record["in:" + pname] = h.detach()
        h = h.flatten(2).gather(2, idx.flatten(2)).view(idx.shape)
    h = h.transpose(1, 2)
    return h.contiguous().view(h.size(0), h.size(1), h.size(2) * h.size(3))


def classifier_train_forward(x, sd, n_blocks, label, keep, m, s, decisions=None, record=None, eps=1e-5):
    """model.py:61-71 in TRAIN mode (BatchNorm1d b2 on batch statistics, model.py:57; b1 / b3 unused, model.py:46,50) with the
    head mask of poolings.py:39-43 made explicit (`keep` [B,H] bool).  Returns (costh, logits).
    x: [B,T,F] — or a LIST of [T_i,F] utterances of different lengths: the reference can only run those one utterance per
    forward through front-end and pooling (no padding, no masks: train.py:117-133); the dense tail then sees the B pooled vectors
    as ONE batch (`decisions` / `record`: lists, one entry per utterance)."""
    q, a = sd["poolingLayer.utteranceAttention.query"], sd["poolingLayer.headsAttention.att"]
    if isinstance(x, (list, tuple)):
        outs = []
        for i, u in enumerate(x):
            rec = {} if record is not None else None
            enc = vgg_forward_decided(u.unsqueeze(0), sd, n_blocks, None if decisions is None else decisions[i], rec)
            if record is not None:
                record.append(rec)
            outs.append(dmha_opseq(enc, q, a, None if keep is None else keep[i:i + 1])[0])
        out = torch.cat(outs, dim=0)
    else:
        enc = vgg_forward_decided(x, sd, n_blocks, decisions, record)
        out, _, _ = dmha_opseq(enc, q, a, keep)
    e1 = F.relu(F.linear(out, sd["fc1.weight"], sd["fc1.bias"]))
    e2 = F.relu(F.linear(e1, sd["fc2.weight"], sd["fc2.bias"]))
    e2 = F.batch_norm(e2, None, None, sd["b2.weight"], sd["b2.bias"], True, 0.0, eps)
    e3 = F.linear(e2, sd["preLayer.weight"], sd["preLayer.bias"])
    return amsoftmax_forward(e3, sd["predictionLayer.W"], label, m, s)


def train_step_grads(sd, batches, n_blocks, m, s, decisions=None, dtype=torch.float64):
    """train.py:215-226: per micro-batch forward, CrossEntropyLoss (mean over the micro-batch), backward; gradients SUMMED over
    the micro-batches.  sd: state_dict (any dtype; cast to `dtype`); batches: [(x [B,T,F], label [B], keep [B,H] bool)];
    decisions: per micro-batch dicts for vgg_forward_decided (or None).
    Returns (losses [float], {parameter name: gradient}, [decisions used per micro-batch])."""
    p = {k: (v.detach().to(dtype).requires_grad_(True) if v.dtype.is_floating_point else v) for k, v in sd.items()}
    losses, used = [], []
    for i, (x, label, keep) in enumerate(batches):
        ragged = isinstance(x, (list, tuple))
        rec = [] if ragged else {}
        _, logits = classifier_train_forward([u.to(dtype) for u in x] if ragged else x.to(dtype), p, n_blocks, label, keep, m, s,
                                             None if decisions is None else decisions[i], rec)
        loss = F.cross_entropy(logits, label)
        loss.backward()
        losses.append(float(loss.detach()))
        used.append(rec)
    grads = {k: v.grad for k, v in p.items() if v.dtype.is_floating_point and v.grad is not None}
    return losses, grads, used


def count_flips(own, other, n_blocks):
    """Decisions of two forwards of the same network on the same input, compared: {'convXY': number of ReLU signs that differ,
    'poolX': number of windows that hand on a different element although the maximum is positive in either forward}.
    `own` must come from vgg_forward_decided(record=...) (it carries the pool inputs)."""
    flips = {}
    for name in vgg_layer_names(n_blocks):
        if name.startswith("conv") and name.endswith("2"):
            # convX2 feeds the pool: only the sign of the element a window hands on has any effect (forward and backward), and a
            # one-launch conv + ReLU + pool only knows that one — compared where `other`'s pool hands on
            idx = other["pool" + name[4]].flatten(2)
            sel = torch.zeros(own[name].flatten(2).shape, dtype=torch.bool).scatter_(2, idx, True).view(own[name].shape)
            flips[name] = int(((own[name] != other[name]) & sel).sum())
        elif name.startswith("conv"):
            flips[name] = int((own[name] != other[name]).sum())
        else:
            h = own["in:" + name].flatten(2)
            live = (h.gather(2, own[name].flatten(2)) > 0) | (h.gather(2, other[name].flatten(2)) > 0)
            flips[name] = int(((own[name].flatten(2) != other[name].flatten(2)) & live).sum())
    return flips
