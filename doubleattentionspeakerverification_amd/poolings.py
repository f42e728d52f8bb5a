"""Pooling layers with the reference's public interface, computed by libdasv_hip.so.

Mirrors /root/reference/scripts/poolings.py (names, constructor arguments, forward
return values, parameter names and shapes — hence state_dict keys):

    new_parameter                      poolings.py:9-12
    Attention(embedding_size)          poolings.py:14-27
    HeadAttention(encoder_size, heads_number, mask_prob=0.25, attentionSmoothing=False)   :29-71
    innerKeyValueAttention(query, key, value)                                            :73-80
    MultiHeadAttention(encoder_size, heads_number)                                        :83-109
    DoubleMHA(encoder_size, heads_number, mask_prob=0.2)                                  :112-129

The whole forward of DoubleMHA / MultiHeadAttention / Attention is ONE HIP launch
(csrc/dmha_fwd.hip) and the backward one streaming launch plus a tiny reduction
(csrc/dmha_bwd.hip).  Inputs must live on the GPU: there is deliberately no CPU
path here (the CPU restatement lives in oracle/ and is test infrastructure only).

Differences from the reference, all deliberate:
  * the train-time head mask (poolings.py:41) is drawn with torch's device generator
    (``torch.empty(..., device=x.device).random_(k) > 0``) instead of the legacy
    ``torch.cuda.FloatTensor`` constructor; same distribution, and it can be supplied
    explicitly (``head_keep=``) for reproducible tests;
  * the returned alignments carry no gradient (the model never differentiates them);
  * ``lengths=`` (optional) lets a zero-padded batch of unequal utterances be pooled as
    if each utterance had been run alone (frames >= length are ignored).
"""
import math
import threading

import torch
from torch import nn

from . import _lib

__all__ = ["new_parameter", "Attention", "HeadAttention", "innerKeyValueAttention",
           "MultiHeadAttention", "DoubleMHA", "double_mha", "multi_head_attention", "DoubleMHAPlan", "reset_workspaces",
           "nhwc_frames_supported"]


def new_parameter(*size):
    out = nn.Parameter(torch.empty(*size, dtype=torch.float32))
    nn.init.xavier_normal_(out)
    return out


# --------------------------------------------------------------------------- plumbing
_ws_lock = threading.Lock()
_fwd_ws = {}


def _stream_ptr(device):
    return torch.cuda.current_stream(device).cuda_stream


def reset_workspaces():
    """Re-zero every cached forward workspace (call after a HIP error was seen on a stream or a graph capture was
    abandoned: the in-launch head exchange relies on its arrival counters being zero between launches)."""
    with _ws_lock:
        for buf in _fwd_ws.values():
            buf.zero_()


def _fwd_workspace(device, nbytes):
    """Arrival counters for the in-launch head exchange: zeroed once, left zeroed by every launch.
    One buffer per (device, stream) because launches sharing it must not overlap in time."""
    key = (device.index, _stream_ptr(device))
    with _ws_lock:
        buf = _fwd_ws.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.zeros(max(nbytes, 4096), dtype=torch.uint8, device=device)
            _fwd_ws[key] = buf
    return buf


def _require_gpu(x, what):
    if not x.is_cuda:
        raise RuntimeError("%s: input is on %s; this module only runs on an AMD GPU through libdasv_hip.so "
                           "(no CPU fallback — see oracle/ for the CPU restatement used by the tests)" % (what, x.device))
    if x.dtype not in (torch.float32, torch.bfloat16, torch.float16):
        raise TypeError("%s: unsupported dtype %s" % (what, x.dtype))


def _rows(x):
    """x [B,T,D] with contiguous features -> (tensor, stride_b, stride_t) usable by the kernels."""
    if x.dim() != 3:
        raise RuntimeError("expected a [batch, frames, features] tensor, got shape %s" % (tuple(x.shape),))
    if x.stride(2) != 1 or x.stride(1) < x.size(2) or x.stride(0) < 0:
        x = x.contiguous()
    return x, x.stride(0), x.stride(1)


def _ptr(t):
    return None if t is None else t.data_ptr()


def _launch_fwd(x, query, att, head_keep, lengths, want_align):
    lib = _lib.load()
    x, sb, st = _rows(x)
    B, T, D = x.shape
    dh, H = query.shape
    if D != dh * H:
        raise RuntimeError("feature size %d does not match query %s" % (D, tuple(query.shape)))
    dev = x.device
    f32 = dict(dtype=torch.float32, device=dev)
    q = query.detach().to(**f32).contiguous()
    a = None if att is None else att.detach().to(**f32).contiguous().view(-1)
    keep = None
    if head_keep is not None:
        keep = head_keep.to(device=dev, dtype=torch.uint8).contiguous()
        if tuple(keep.shape) != (B, H):
            raise RuntimeError("head_keep must be [B,H]")
    lens = None
    if lengths is not None:
        lens = torch.as_tensor(lengths).to(device=dev, dtype=torch.int32).contiguous()
        if tuple(lens.shape) != (B,):
            raise RuntimeError("lengths must be [B]")
    ctxv = torch.empty(B, H, dh, **f32)
    ml = torch.empty(B, H, 2, **f32)
    align = torch.empty(B, T, H, **f32) if want_align else None
    out = torch.empty(B, dh, **f32) if a is not None else None
    head_w = torch.empty(B, H, **f32) if a is not None else None
    nbytes = lib.dasv_dmha_workspace_bytes(B, T, H, dh)
    ws = _fwd_workspace(dev, nbytes)
    with torch.cuda.device(dev):
        rc = lib.dasv_dmha_fwd(_lib.DTYPE_CODES[str(x.dtype).split(".")[-1]], x.data_ptr(), q.data_ptr(), _ptr(a),
                               _ptr(keep), _ptr(lens), _ptr(out), _ptr(align), _ptr(head_w), ctxv.data_ptr(),
                               ml.data_ptr(), ws.data_ptr(), ws.numel(), B, T, H, dh, sb, st, _stream_ptr(dev))
        if rc != 0:   # a launch that did not complete may leave arrival counters behind: later launches would never elect
            lib.dasv_dmha_workspace_init(ws.data_ptr(), ws.numel(), _stream_ptr(dev))   # a last arriver
    _lib.check(rc, "dasv_dmha_fwd")
    return x, q, a, keep, lens, out, align, head_w, ctxv, ml


def _launch_bwd(x, q, a, lens, head_w, ctxv, ml, out, go):
    lib = _lib.load()
    B, T, D = x.shape
    dh, H = q.shape
    dev = x.device
    sb, st = x.stride(0), x.stride(1)
    gx = torch.empty_strided(x.shape, x.stride(), dtype=x.dtype, device=dev)
    if x.stride(1) != D or x.stride(0) != T * D:
        gx.zero_()  # gaps between rows are never written by the kernel
    gq = torch.empty(dh, H, dtype=torch.float32, device=dev)
    ga = torch.empty(dh, 1, dtype=torch.float32, device=dev) if a is not None else None
    nbytes = lib.dasv_dmha_bwd_workspace_bytes(B, T, H, dh)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    go = go.to(dtype=torch.float32).contiguous()
    with torch.cuda.device(dev):
        rc = lib.dasv_dmha_bwd(_lib.DTYPE_CODES[str(x.dtype).split(".")[-1]], x.data_ptr(), q.data_ptr(), _ptr(a),
                               None, _ptr(lens), _ptr(head_w), ctxv.data_ptr(), ml.data_ptr(), _ptr(out),
                               go.data_ptr(), gx.data_ptr(), gq.data_ptr(), _ptr(ga), ws.data_ptr(), ws.numel(),
                               B, T, H, dh, sb, st, _stream_ptr(dev))
    _lib.check(rc, "dasv_dmha_bwd")
    return gx, gq, ga


def nhwc_frames_supported(h4, heads):
    """True when `h4` — a front-end map [B, C, T', F'] fp32 in channels_last memory ([B][T'][F'][C]) — can be pooled in place:
    the reference's frame vector is `h4.transpose(1, 2).contiguous().view(B, T', C*F')` (CNNs.py:88-89, feature c*F' + f)."""
    if h4.dim() != 4 or not h4.is_cuda or h4.dtype != torch.float32:
        return False
    B, C, T, F = h4.shape
    if h4.stride() != (T * F * C, 1, F * C, C):
        return False
    return bool(_lib.load().dasv_dmha_nhwc_supported(F, C, heads))


def _launch_fwd_nhwc(h4, query, att, head_keep, lengths, want_align, want_ctx=True):
    lib = _lib.load()
    B, C, T, F = h4.shape
    dh, H = query.shape
    if C * F != dh * H:
        raise RuntimeError("feature size %d does not match query %s" % (C * F, tuple(query.shape)))
    dev = h4.device
    f32 = dict(dtype=torch.float32, device=dev)
    q = query.detach().to(**f32).contiguous()
    a = None if att is None else att.detach().to(**f32).contiguous().view(-1)
    keep = None
    if head_keep is not None:
        keep = head_keep.to(device=dev, dtype=torch.uint8).contiguous()
        if tuple(keep.shape) != (B, H):
            raise RuntimeError("head_keep must be [B,H]")
    lens = None
    if lengths is not None:
        lens = torch.as_tensor(lengths).to(device=dev, dtype=torch.int32).contiguous()
        if tuple(lens.shape) != (B,):
            raise RuntimeError("lengths must be [B]")
    # the per-head context vectors and softmax statistics are what the backward pass needs; inference with the head stage in
    # the launch does not write them at all (1.3 MB per [64,19,5120] map)
    want_ctx = want_ctx or a is None
    ctxv = torch.empty(B, H, dh, **f32) if want_ctx else None
    ml = torch.empty(B, H, 2, **f32) if want_ctx else None
    align = torch.empty(B, T, H, **f32) if want_align else None
    out = torch.empty(B, dh, **f32) if a is not None else None
    head_w = torch.empty(B, H, **f32) if a is not None else None
    ws = _fwd_workspace(dev, lib.dasv_dmha_workspace_bytes(B, T, H, dh))
    with torch.cuda.device(dev):
        rc = lib.dasv_dmha_fwd_nhwc_f32(h4.data_ptr(), q.data_ptr(), _ptr(a), _ptr(keep), _ptr(lens), _ptr(out), _ptr(align),
                                        _ptr(head_w), _ptr(ctxv), _ptr(ml), ws.data_ptr(), ws.numel(), B, T, F, C, H,
                                        _stream_ptr(dev))
        if rc != 0:
            lib.dasv_dmha_workspace_init(ws.data_ptr(), ws.numel(), _stream_ptr(dev))
    _lib.check(rc, "dasv_dmha_fwd_nhwc_f32")
    return q, a, lens, out, align, head_w, ctxv, ml


class _PoolNHWCFn(torch.autograd.Function):
    """The fused pooling on the front-end's own NHWC map (no transposing copy, forward or backward): h4 [B,C,T',F']
    channels_last, query, att(optional) -> (out | ctx, align).  The gradient comes back as a channels_last map."""

    @staticmethod
    def forward(ctx, h4, query, att, head_keep, lengths, want_align):
        need_bwd = any(ctx.needs_input_grad[:3])
        q, a, lens, out, align, head_w, ctxv, ml = _launch_fwd_nhwc(h4, query, att, head_keep, lengths, want_align, want_ctx=need_bwd)
        if need_bwd:
            ctx.save_for_backward(h4, q, a, lens, head_w, ctxv, ml, out)
        ctx.q_dtype = query.dtype
        ctx.a_meta = None if att is None else (att.dtype, att.shape)
        dh, H = q.shape
        B = h4.shape[0]
        if align is None:
            align = torch.empty(0, device=h4.device)
        primary = out if a is not None else ctxv.view(B, H * dh)
        ctx.mark_non_differentiable(align)
        return primary, align

    @staticmethod
    def backward(ctx, g_primary, _g_align):
        h4, q, a, lens, head_w, ctxv, ml, out = ctx.saved_tensors
        lib = _lib.load()
        B, C, T, F = h4.shape
        dh, H = q.shape
        dev = h4.device
        gx = torch.empty_like(h4, memory_format=torch.channels_last)
        if gx.stride() != h4.stride():
            gx = torch.empty_strided(h4.shape, h4.stride(), dtype=h4.dtype, device=dev)
        gq = torch.empty(dh, H, dtype=torch.float32, device=dev)
        ga = torch.empty(dh, 1, dtype=torch.float32, device=dev) if a is not None else None
        nbytes = lib.dasv_dmha_bwd_workspace_bytes(B, T, H, dh)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        go = g_primary.to(dtype=torch.float32).contiguous()
        with torch.cuda.device(dev):
            rc = lib.dasv_dmha_bwd_nhwc_f32(h4.data_ptr(), q.data_ptr(), _ptr(a), _ptr(lens), _ptr(head_w), ctxv.data_ptr(),
                                            ml.data_ptr(), _ptr(out), go.data_ptr(), gx.data_ptr(), gq.data_ptr(), _ptr(ga),
                                            ws.data_ptr(), ws.numel(), B, T, F, C, H, _stream_ptr(dev))
        _lib.check(rc, "dasv_dmha_bwd_nhwc_f32")
        gq = gq.to(ctx.q_dtype)
        if ga is not None:
            ga = ga.to(ctx.a_meta[0]).view(ctx.a_meta[1])
        return gx, gq, ga, None, None, None


class _PoolFn(torch.autograd.Function):
    """x, query, att(optional) -> (out | ctx, align, head_w, ctx).  One HIP launch each way."""

    @staticmethod
    def forward(ctx, x, query, att, head_keep, lengths, want_align):
        xk, q, a, keep, lens, out, align, head_w, ctxv, ml = _launch_fwd(x, query, att, head_keep, lengths, want_align)
        ctx.save_for_backward(xk, q, a, lens, head_w, ctxv, ml, out)
        ctx.in_shape = x.shape
        ctx.q_dtype = query.dtype
        ctx.a_meta = None if att is None else (att.dtype, att.shape)
        B, H, dh = ctxv.shape
        if align is None:
            align = torch.empty(0, device=x.device)
        if a is not None:
            primary, hw, side = out, head_w, ctxv
        else:  # per-head stage only: the context vectors ARE the differentiable output
            primary, hw, side = ctxv.view(B, H * dh), torch.empty(0, device=x.device), torch.empty(0, device=x.device)
        ctx.mark_non_differentiable(align, hw, side)
        return primary, align, hw, side

    @staticmethod
    def backward(ctx, g_primary, _g_align, _g_hw, g_ctxv):
        xk, q, a, lens, head_w, ctxv, ml, out = ctx.saved_tensors
        if g_primary is None:
            g_primary = torch.zeros_like(out if a is not None else ctxv.view(ctxv.size(0), -1))
        gx, gq, ga = _launch_bwd(xk, q, a, lens, head_w, ctxv, ml, out, g_primary)
        gx = gx.view(ctx.in_shape) if gx.shape == ctx.in_shape else gx
        gq = gq.to(ctx.q_dtype)
        if ga is not None:
            ga = ga.to(ctx.a_meta[0]).view(ctx.a_meta[1])
        return gx, gq, ga, None, None, None


def double_mha(x, query, att, head_keep=None, lengths=None, want_align=True):
    """Functional form: returns (out [B,dh], align [B,T,H] or None, head_w [B,H,1], ctx [B,H,dh])."""
    _require_gpu(x, "double_mha")
    out, align, hw, ctxv = _PoolFn.apply(x, query, att, head_keep, lengths, want_align)
    return out, (align if want_align else None), hw.unsqueeze(-1), ctxv


def multi_head_attention(x, query, lengths=None, want_align=True):
    """Functional form of the per-head stage only: returns (ctx [B,D], align [B,T,H] or None)."""
    _require_gpu(x, "multi_head_attention")
    ctx, align, _, _ = _PoolFn.apply(x, query, None, None, lengths, want_align)
    return ctx, (align if want_align else None)


class DoubleMHAPlan:
    """Allocation-free inference launcher for a fixed [B,T,D] shape (serving loops, hipGraph capture).

    Outputs, saved statistics and the workspace are allocated once; every call is exactly one
    dasv_dmha_fwd launch on the current stream and returns the SAME output tensors (consume or copy them
    before the next call).  No autograd."""

    def __init__(self, module, batch, frames, dtype=torch.float32, device=None, want_align=True):
        lib = _lib.load()
        q = module.utteranceAttention.query
        device = torch.device(device) if device is not None else q.device
        dh, H = q.shape
        f32 = dict(dtype=torch.float32, device=device)
        self.B, self.T, self.H, self.dh, self.D = batch, frames, H, dh, dh * H
        self.dtype, self.device = dtype, device
        self.query = q.detach().to(**f32).contiguous()
        self.att = module.headsAttention.att.detach().to(**f32).contiguous().view(-1)
        self.out = torch.empty(batch, dh, **f32)
        self.align = torch.empty(batch, frames, H, **f32) if want_align else None
        self.head_w = torch.empty(batch, H, **f32)
        self.ctx = torch.empty(batch, H, dh, **f32)
        self.ml = torch.empty(batch, H, 2, **f32)
        self.ws = torch.zeros(max(4096, lib.dasv_dmha_workspace_bytes(batch, frames, H, dh)), dtype=torch.uint8, device=device)
        self._fn = lib.dasv_dmha_fwd
        self._code = _lib.DTYPE_CODES[str(dtype).split(".")[-1]]
        self._tail = (self.query.data_ptr(), self.att.data_ptr(), None, None, self.out.data_ptr(), _ptr(self.align),
                      self.head_w.data_ptr(), self.ctx.data_ptr(), self.ml.data_ptr(), self.ws.data_ptr(),
                      self.ws.numel(), batch, frames, H, dh)

    def __call__(self, x):
        if x.shape != (self.B, self.T, self.D) or x.dtype != self.dtype or x.device != self.device or x.stride(2) != 1:
            raise RuntimeError("DoubleMHAPlan was built for a %s %s tensor on %s" % ((self.B, self.T, self.D), self.dtype, self.device))
        rc = self._fn(self._code, x.data_ptr(), *self._tail, x.stride(0), x.stride(1), _stream_ptr(self.device))
        if rc != 0:
            _lib.load().dasv_dmha_workspace_init(self.ws.data_ptr(), self.ws.numel(), _stream_ptr(self.device))
            _lib.check(rc, "dasv_dmha_fwd")
        return self.out, self.align

    def reset(self):
        """Re-zero the arrival counters (after a HIP error on the stream or an abandoned graph capture: a launch that did
        not run to completion leaves them non-zero and every later launch would return uninitialised outputs)."""
        _lib.check(_lib.load().dasv_dmha_workspace_init(self.ws.data_ptr(), self.ws.numel(), _stream_ptr(self.device)),
                   "dasv_dmha_workspace_init")


def _like_input(t, x):
    return t if t.dtype == x.dtype else t.to(x.dtype)


# --------------------------------------------------------------------------- modules
class Attention(nn.Module):
    """Single learned-query attention pooling over time (poolings.py:14-27)."""

    def __init__(self, embedding_size):
        super().__init__()
        self.embedding_size = embedding_size
        self.att = new_parameter(self.embedding_size, 1)

    def forward(self, ht, lengths=None):
        # one head spanning all features; score scale 1/sqrt(1) = 1, as the reference's plain matmul
        ct, score = multi_head_attention(ht, self.att, lengths=lengths)
        return _like_input(ct, ht), _like_input(score, ht)


class HeadAttention(nn.Module):
    """Attention over the H per-head context vectors (poolings.py:29-71, narrow path).

    Inside DoubleMHA this stage runs in the tail of the fused launch; the standalone forward below
    (never on the model's hot path) uses torch's device ops on the small [B,H,dh] tensor."""

    def __init__(self, encoder_size, heads_number, mask_prob=0.25, attentionSmoothing=False):
        super().__init__()
        self.embedding_size = encoder_size // heads_number
        self.att = new_parameter(self.embedding_size, 1)
        self.mask_prob = int(1 / mask_prob)
        if attentionSmoothing:
            raise NotImplementedError("attentionSmoothing=True is unreachable dead code in the reference (poolings.py:53-59)")
        self.attentionSmoothing = False

    def draw_head_keep(self, batch, heads, device):
        """poolings.py:41: keep = randint{0..k-1} > 0 with k = int(1/mask_prob)."""
        return torch.empty(batch, heads, device=device).random_(self.mask_prob) > 0

    def forward(self, ht, head_keep=None):
        _require_gpu(ht, "HeadAttention")
        B, H, _ = ht.shape
        score = torch.matmul(ht, self.att.to(ht.dtype)).view(B, H)
        if self.training and head_keep is None:
            head_keep = self.draw_head_keep(B, H, ht.device)
        if head_keep is not None:
            score = score.masked_fill(~head_keep.to(torch.bool), -float("inf"))
        w = torch.softmax(score, dim=-1).view(B, H, 1)
        return torch.sum(ht * w, dim=1), w


def innerKeyValueAttention(query, key, value):
    """poolings.py:73-80 on GPU tensors: key [B*T,H,dh], value [B,T,H,dh] views of one [B,T,D] tensor."""
    B, T, H, dh = value.shape
    x = value.reshape(B, T, H * dh)
    ctx, align = multi_head_attention(x, query)
    return _like_input(ctx.view(B, H, dh), value), _like_input(align, value)


class MultiHeadAttention(nn.Module):
    def __init__(self, encoder_size, heads_number):
        super().__init__()
        self.encoder_size = encoder_size
        assert self.encoder_size % heads_number == 0  # d_model
        self.head_size = self.encoder_size // heads_number
        self.heads_number = heads_number
        self.query = new_parameter(self.head_size, self.heads_number)
        self.alignment = None

    def getAlignments(self, ht):
        _, self.alignment = multi_head_attention(ht, self.query)
        return self.alignment

    def getHeadsContextVectors(self, ht):
        ctx, self.alignment = multi_head_attention(ht, self.query)
        return _like_input(ctx.view(ht.size(0), self.heads_number, self.head_size), ht)

    def forward(self, ht, lengths=None):
        ctx, self.alignment = multi_head_attention(ht, self.query, lengths=lengths)
        return _like_input(ctx, ht), _like_input(self.alignment, ht)


class DoubleMHA(nn.Module):
    def __init__(self, encoder_size, heads_number, mask_prob=0.2):
        super().__init__()
        self.heads_number = heads_number
        self.utteranceAttention = MultiHeadAttention(encoder_size, heads_number)
        self.heads_size = encoder_size // heads_number
        self.headsAttention = HeadAttention(encoder_size, heads_number, mask_prob=mask_prob, attentionSmoothing=False)

    def _run(self, x, head_keep, lengths, want_align):
        _require_gpu(x, "DoubleMHA")
        if self.training and head_keep is None:
            head_keep = self.headsAttention.draw_head_keep(x.size(0), self.heads_number, x.device)
        return double_mha(x, self.utteranceAttention.query, self.headsAttention.att,
                          head_keep=head_keep, lengths=lengths, want_align=want_align)

    def getAlignments(self, x, head_keep=None, lengths=None):
        _, alignment, head_w, _ = self._run(x, head_keep, lengths, True)
        return _like_input(alignment, x), _like_input(head_w, x)

    def forward(self, x, head_keep=None, lengths=None):
        out, alignment, _, _ = self._run(x, head_keep, lengths, True)
        self.utteranceAttention.alignment = alignment
        return _like_input(out, x), _like_input(alignment, x)

    def pool(self, x, head_keep=None, lengths=None):
        """forward() without materialising the [B,T,H] alignments (what the model needs)."""
        out, _, _, _ = self._run(x, head_keep, lengths, False)
        return _like_input(out, x)

    def pool_map(self, h4, head_keep=None, lengths=None, want_align=False):
        """pool() / forward() straight from the front-end's [B,C,T',F'] channels_last map (see nhwc_frames_supported):
        the frame vectors [B,T',C*F'] of CNNs.py:88-89 are never materialised.  Returns out, or (out, alignment)."""
        if self.training and head_keep is None:
            head_keep = self.headsAttention.draw_head_keep(h4.size(0), self.heads_number, h4.device)
        try:
            out, align = _PoolNHWCFn.apply(h4, self.utteranceAttention.query, self.headsAttention.att, head_keep, lengths, want_align)
        except _lib.DasvError as e:
            if e.code != -2:   # DASV_E_SHAPE: maps too long for the in-place kernel's LDS score rows (no shape-agnostic NHWC form)
                raise
            frames = h4.permute(0, 2, 1, 3).reshape(h4.size(0), h4.size(2), h4.size(1) * h4.size(3))
            out, align, _, _ = self._run(frames, head_keep, lengths, want_align)   # any length pools, like the reference
        if want_align:
            self.utteranceAttention.alignment = align
            return out, align
        return out
