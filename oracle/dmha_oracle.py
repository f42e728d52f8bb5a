"""CPU restatement of the reference hot path (test infrastructure, see oracle/__init__.py).

Two flavours of every pooling function:

* ``*_opseq``  — plain torch CPU ops issued in the same order as the reference
  (batched matmul -> diagonal -> softmax -> broadcast multiply -> sum), so its
  cost profile is that of the reference's CPU path.  This is what
  ``bench.py``'s ``cpu_baseline`` leg times (the reference files themselves
  never travel to the GPU box).
* ``*_closed`` — numpy closed forms (float64 by default) of the same maths,
  used as the high-precision checker for the HIP kernels.

Reference lines restated (all under /root/reference/scripts):
  poolings.py:14-27    Attention                     -> attention_pool_*
  poolings.py:73-80    innerKeyValueAttention        -> mha_*
  poolings.py:93-109   MultiHeadAttention            -> mha_*
  poolings.py:39-51,61-71 HeadAttention (narrow)     -> head_attention_*
  poolings.py:120-129  DoubleMHA                     -> dmha_*
  CNNs.py:7-20         output-dimension helpers      -> vgg_output_dim
  CNNs.py:34-52,68-91  VGG3L / VGG4L forward         -> vgg_forward
  model.py:52-71       getEmbedding / forward        -> embedding_forward / classifier_forward
  loss.py:24-52        AMSoftmax                     -> amsoftmax_forward / annealed_factor
  utils.py:5-21        Score / scoreCosineDistance   -> far_frr / cosine_score
  train.py:135-150     EER search                    -> eer_from_scores
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

__all__ = [
    "mha_opseq", "head_attention_opseq", "dmha_opseq", "attention_pool_opseq",
    "mha_closed", "dmha_closed", "dmha_backward_closed", "attention_pool_closed",
    "head_keep_from_torch_rng", "vgg_output_dim", "vgg_out_frames", "vgg_forward",
    "embedding_forward", "classifier_forward", "amsoftmax_forward", "annealed_factor",
    "cosine_score", "far_frr", "eer_from_scores", "dmha_c",
]


# --------------------------------------------------------------------------
# op-sequence-faithful torch versions (CPU baseline + fp32 checker)
# --------------------------------------------------------------------------
def mha_opseq(x, query):
    """poolings.py:73-80 + 100-109.  x:[B,T,D] contiguous, query:[dh,H].

    Returns (ctx [B,D], alignment [B,T,H]).  Note the score scale is
    1/sqrt(query.size(-1)) = 1/sqrt(H) (poolings.py:75)."""
    B, T, D = x.shape
    dh, H = query.shape
    assert D == dh * H
    keys = x.view(B * T, H, dh)
    vals = x.view(B, T, H, dh)
    full = torch.matmul(keys, query) / math.sqrt(query.size(-1))   # [B*T,H,H]
    scores = torch.diagonal(full, dim1=-2, dim2=-1).view(B, T, H)
    p = F.softmax(scores, dim=-2)
    ctx = torch.sum(vals * p.unsqueeze(-1), dim=1)                 # [B,H,dh]
    return ctx.view(B, -1), p


def head_attention_opseq(c, att, head_keep=None):
    """poolings.py:45-51, 61-71.  c:[B,H,dh], att:[dh,1].

    ``head_keep`` ([B,H] bool, True = keep) restates the train-time mask of
    poolings.py:39-43 with the random draw made explicit."""
    B, H, _ = c.shape
    u = torch.matmul(c, att).squeeze()
    if head_keep is not None:
        u = u.clone().view(B, H)
        u[~head_keep] = -float("inf")
    w = F.softmax(u, dim=-1).view(B, H, 1)
    return torch.sum(c * w, dim=1), w


def dmha_opseq(x, query, att, head_keep=None):
    """poolings.py:126-129 (+120-124).  Returns (out [B,dh], alignment [B,T,H], head_w [B,H,1])."""
    dh, H = query.shape
    ctx, p = mha_opseq(x, query)
    out, w = head_attention_opseq(ctx.view(ctx.size(0), H, dh), att, head_keep)
    return out, p, w


def attention_pool_opseq(x, att):
    """poolings.py:22-27.  x:[B,T,D], att:[D,1] -> ([B,D], [B,T,1])."""
    B, T, _ = x.shape
    s = torch.matmul(x, att).squeeze()
    a = F.softmax(s, dim=-1).view(B, T, 1)
    return torch.sum(x * a, dim=1), a


def head_keep_from_torch_rng(B, H, mask_prob, seed):
    """The draw of poolings.py:41 made on the CPU generator: keep = randint(k) > 0, k = int(1/mask_prob)."""
    k = int(1 / mask_prob)
    g = torch.Generator().manual_seed(seed)
    return torch.empty(B, H).random_(k, generator=g) > 0


# --------------------------------------------------------------------------
# closed forms (numpy)
# --------------------------------------------------------------------------
def _softmax(a, axis):
    a = a - np.max(a, axis=axis, keepdims=True)
    e = np.exp(a)
    return e / np.sum(e, axis=axis, keepdims=True)


def mha_closed(x, query, dtype=np.float64):
    x = np.asarray(x, dtype=dtype)
    q = np.asarray(query, dtype=dtype)
    B, T, D = x.shape
    dh, H = q.shape
    xv = x.reshape(B, T, H, dh)
    s = np.einsum("bthd,dh->bth", xv, q) / math.sqrt(H)
    p = _softmax(s, axis=1)
    c = np.einsum("bth,bthd->bhd", p, xv)
    return c, p, s


def dmha_closed(x, query, att, head_keep=None, dtype=np.float64):
    """Returns dict(out [B,dh], align [B,T,H], head_w [B,H], ctx [B,H,dh], m [B,H], l [B,H])."""
    c, p, s = mha_closed(x, query, dtype)
    a = np.asarray(att, dtype=dtype).reshape(-1)
    u = c @ a
    if head_keep is not None:
        u = np.where(np.asarray(head_keep, dtype=bool), u, -np.inf)
    with np.errstate(invalid="ignore"):
        w = _softmax(u, axis=1)
    out = np.einsum("bh,bhd->bd", w, c)
    m = s.max(axis=1)
    l = np.exp(s - m[:, None, :]).sum(axis=1)
    return dict(out=out, align=p, head_w=w, ctx=c, m=m, l=l)


def dmha_backward_closed(x, query, att, grad_out, head_keep=None, dtype=np.float64):
    """Gradients of sum(out * grad_out) w.r.t. x, query, att."""
    x = np.asarray(x, dtype=dtype)
    q = np.asarray(query, dtype=dtype)
    a = np.asarray(att, dtype=dtype).reshape(-1)
    go = np.asarray(grad_out, dtype=dtype)
    B, T, D = x.shape
    dh, H = q.shape
    f = dmha_closed(x, q, a, head_keep, dtype)
    c, p, w = f["ctx"], f["align"], f["head_w"]
    xv = x.reshape(B, T, H, dh)
    gw = np.einsum("bd,bhd->bh", go, c)
    gu = w * (gw - np.sum(w * gw, axis=1, keepdims=True))
    gc = w[:, :, None] * go[:, None, :] + gu[:, :, None] * a[None, None, :]
    ga = np.einsum("bh,bhd->d", gu, c)
    gp = np.einsum("bhd,bthd->bth", gc, xv)
    gs = p * (gp - np.einsum("bhd,bhd->bh", gc, c)[:, None, :])
    rs = 1.0 / math.sqrt(H)
    gx = p[..., None] * gc[:, None, :, :] + (gs[..., None] * q.T[None, None, :, :]) * rs
    gq = np.einsum("bth,bthd->dh", gs, xv) * rs
    return dict(grad_x=gx.reshape(B, T, D), grad_query=gq, grad_att=ga.reshape(dh, 1))


def attention_pool_closed(x, att, dtype=np.float64):
    x = np.asarray(x, dtype=dtype)
    a = np.asarray(att, dtype=dtype).reshape(-1)
    p = _softmax(x @ a, axis=1)
    return np.einsum("bt,btd->bd", p, x), p[..., None]


# --------------------------------------------------------------------------
# front-end, model, loss, scoring
# --------------------------------------------------------------------------
def vgg_output_dim(feature_size, kernel_size, n_blocks):
    """CNNs.py:7-20: ceil-halve the mel axis once per block, times the channel count."""
    f = feature_size
    for _ in range(n_blocks):
        f = int(math.ceil(f / 2))
    return f * kernel_size


def vgg_out_frames(T, n_blocks):
    for _ in range(n_blocks):
        T = (T + 1) // 2
    return T


def vgg_forward(x, sd, n_blocks, prefix="front_end."):
    """CNNs.py:34-52 / 68-91 from a state_dict: x [B,T,F] -> [B,T',C*F']."""
    h = x.view(x.size(0), x.size(1), 1, x.size(2)).transpose(1, 2)
    for blk in range(1, n_blocks + 1):
        for j in (1, 2):
            name = "%sconv%d%d" % (prefix, blk, j)
            h = F.relu(F.conv2d(h, sd[name + ".weight"], sd[name + ".bias"], stride=1, padding=1))
        h = F.max_pool2d(h, 2, stride=2, ceil_mode=True)
    h = h.transpose(1, 2)
    return h.contiguous().view(h.size(0), h.size(1), h.size(2) * h.size(3))


def _bn_eval(x, sd, name, eps=1e-5):
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"],
                        sd[name + ".weight"], sd[name + ".bias"], False, 0.0, eps)


def embedding_forward(x, sd, n_blocks, heads):
    """model.py:52-59 in eval mode (BatchNorm on running statistics)."""
    enc = vgg_forward(x, sd, n_blocks)
    out, _, _ = dmha_opseq(enc, sd["poolingLayer.utteranceAttention.query"],
                           sd["poolingLayer.headsAttention.att"])
    e1 = F.relu(F.linear(out, sd["fc1.weight"], sd["fc1.bias"]))
    e2 = F.relu(F.linear(e1, sd["fc2.weight"], sd["fc2.bias"]))
    return _bn_eval(e2, sd, "b2")


def annealed_factor(step, annealing):
    """loss.py:24-29."""
    alpha = max(0.0, 1000.0 / pow(1.0 + 0.0001 * float(step), 2.0)) if annealing else 0.0
    return 1.0 / (1.0 + alpha)


def amsoftmax_forward(x, W, label, m, s, annealing=False, step=0):
    """loss.py:37-52.  Returns (costh, scaled margin logits)."""
    xn = x / torch.norm(x, p=2, dim=1, keepdim=True).clamp(min=1e-12)
    wn = W / torch.norm(W, p=2, dim=0, keepdim=True).clamp(min=1e-12)
    costh = torch.mm(xn, wn)
    delta = torch.zeros_like(costh).scatter_(1, label.view(-1, 1), m)
    alpha = max(0.0, 1000.0 / pow(1.0 + 0.0001 * float(step), 2.0)) if annealing else 0.0
    comb = ((costh - delta) + alpha * costh) / (1 + alpha)
    return costh, s * comb


def classifier_forward(x, sd, n_blocks, heads, label, m, s, annealing=False, step=0):
    """model.py:61-71 in eval mode."""
    emb = embedding_forward(x, sd, n_blocks, heads)
    e3 = F.linear(emb, sd["preLayer.weight"], sd["preLayer.bias"])
    return amsoftmax_forward(e3, sd["predictionLayer.W"], label, m, s, annealing, step)


def cosine_score(e1, e2):
    """utils.py:18-21."""
    return F.cosine_similarity(e1, e2, dim=-1, eps=1e-8)


def far_frr(scores, th, rate):
    """utils.py:5-15 vectorised: percentage (rounded to 4 dp) of scores >= th (FAR) or < th (FRR)."""
    sc = np.asarray(scores, dtype=np.float64)
    n = np.count_nonzero(sc >= float(th)) if rate == "FAR" else np.count_nonzero(sc < float(th))
    return round(n * 100 / float(len(sc)), 4)


def eer_from_scores(clients, impostors):
    """train.py:135-150: thresholds -1..1 step 0.01, first sign change of FAR-FRR, else 50."""
    th = np.arange(-1, 1, 0.01)
    frr = np.array([far_frr(clients, t, "FRR") for t in th])
    far = np.array([far_frr(impostors, t, "FAR") for t in th])
    idx = np.argwhere(np.diff(np.sign(far - frr)) != 0).reshape(-1)
    if len(idx) == 0:
        return 50.0
    i = int(idx[0])
    return round((far[i] + frr[i]) / 2, 4)


# --------------------------------------------------------------------------
# plain-C restatement (oracle/dmha_oracle.c), loaded through ctypes
# --------------------------------------------------------------------------
def dmha_c(x, query, att, head_keep=None):
    """Run oracle/dmha_oracle.c (build with `make -C oracle`).  Returns the same dict keys as dmha_closed."""
    import ctypes
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    lib = ctypes.CDLL(os.path.join(here, "libdmha_oracle_c.so"))
    x = np.ascontiguousarray(x, dtype=np.float32)
    q = np.ascontiguousarray(query, dtype=np.float32)
    a = np.ascontiguousarray(att, dtype=np.float32).reshape(-1)
    B, T, D = x.shape
    dh, H = q.shape
    out = np.empty((B, dh), np.float32)
    align = np.empty((B, T, H), np.float32)
    head_w = np.empty((B, H), np.float32)
    ctx = np.empty((B, H, dh), np.float32)
    keep = None if head_keep is None else np.ascontiguousarray(head_keep, dtype=np.uint8)
    fp = ctypes.POINTER(ctypes.c_float)
    rc = lib.dmha_oracle_forward(x.ctypes.data_as(fp), q.ctypes.data_as(fp), a.ctypes.data_as(fp),
                                 None if keep is None else keep.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)),
                                 out.ctypes.data_as(fp), align.ctypes.data_as(fp), head_w.ctypes.data_as(fp),
                                 ctx.ctypes.data_as(fp), B, T, H, dh)
    if rc != 0:
        raise MemoryError("dmha_oracle_forward")
    return dict(out=out, align=align, head_w=head_w, ctx=ctx)
