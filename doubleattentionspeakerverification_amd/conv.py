"""conv3x3 + bias + ReLU through libdasv_hip.so (csrc/conv3x3.hip): the hand-written MFMA implicit-GEMM that
replaces `F.relu(self.convXY(x))` of the reference's CNNs.py:38-48 / 72-86 for the wide layers.

Activations stay fp32 in NHWC memory (torch `channels_last` tensors are passed zero-copy).  Operand precision (`mode`):
  "split8"  fp32-grade forward on the fewest matrix-pipe cycles: v = fp16 hi + residual, hi*hi on the fp16 MFMA, both cross terms
            on one block-scaled e4m3 MFMA (~1e-5 of the output scale vs a float64 convolution); activations only — gradients
            of a layer run in "split";
  "split"   fp32-grade on the bf16 matrix cores, forward and backward (operands split hi + lo, three MFMAs per product, ~5e-6);
  "bf16"    operands rounded to bf16;  "fp16"  operands rounded to fp16 (11 mantissa bits, forward only: its gradients run with
            bf16 operands, whose exponent range covers small gradients).  Backward: data gradient = the same kernel with transposed, flipped weights; weight gradient =
csrc/conv3x3_wgrad.hip.
"""
import threading
import weakref

import torch

from . import _lib

__all__ = ["conv3x3_bias_relu", "PackedConvWeights", "invalidate_packed_weights", "supported", "conv3x3_c1_bias_relu", "maxpool2x2_ceil",
           "conv3x3_wgrad", "supported_wgrad", "conv3x3_c1_conv3x3_bias_relu", "conv3x3_x8",
           "maxpool2x2_ceil_autograd", "conv3x3_c1_bias_relu_autograd", "new_ranges", "range_of", "conv3x3_f16", "pool_fusable"]

MODES = {"split": 0, "bf16": 1, "fp16": 2, "split8": 3}
# "split8": fp32-grade on fewer matrix-pipe cycles (fp16 products + block-scaled e4m3 cross terms, csrc/conv3x3.hip X8).  Its
# operand planes have fp16's / e4m3's exponent range, which is why round 2 kept gradients on the split-bf16 kernel; with every
# tensor scaled by its range record the DATA gradient runs in split8 too (DGRAD_MODE), and since round 3 the WEIGHT gradient as well
# (WGRAD_MODE; its own kernel, csrc/conv3x3_wgrad.hip X8).
BACKWARD_MODE = {"split": "split", "bf16": "bf16", "fp16": "bf16", "split8": "split"}
DGRAD_MODE = {"split": "split", "bf16": "bf16", "fp16": "bf16", "split8": "split8"}
WGRAD_MODE = {"split": "split", "bf16": "bf16", "fp16": "bf16", "split8": "split8"}
_PACK_KIND = {"split": 0, "bf16": 0, "fp16": 1, "split8": 2}
USE_HIP_WGRAD = True   # weight gradients on the hand-written kernel (False: ATen / MIOpen)


def supported(cin, cout):
    """Every layer runs on the hand-written kernels: the MFMA implicit GEMM works on 64-channel slabs and 128-channel output
    tiles, and a layer whose channel counts are not such multiples (the reference takes any `kernel_size`, CNNs.py:27-32 /
    59-66) is run as the zero-padded layer of the next such size — padded weight planes, padded activations between the layers
    (`padded_channels`): the pad channels are identically zero through bias, ReLU and max-pool, so nothing else changes."""
    return cin > 0 and cout > 0


def padded_channels(cout):
    """Channels of the activation tensor a layer with `cout` output channels writes (a multiple of the kernel's 128-channel tile)."""
    return (cout + 127) // 128 * 128


def exact(cin, cout):
    """True when a layer needs no padding at all (what the packed-activation chains and the ragged path are built for)."""
    return cin % 64 == 0 and cout % 128 == 0


_REC_POOLS = {}      # (device, stream) -> [zeroed int32 [1024, 2] block, rows handed out]
_REC_LOCK = threading.Lock()      # (two host threads driving the same stream must never be handed the same row)


def new_ranges(device, n=1):
    """n zeroed RANGE RECORDS (include/dasv_hip.h): row i = (bits of max |v|, exponent of a packed tensor) of one activation
    tensor.  The split8 convolution scales its operand planes by its input's record — on the device, no host round trip — and
    every forward kernel of the front-end fills the record of what it writes.
    Rows of a zeroed block per (device, stream), each handed out once (a training step asks for ~20 records one at a time: one fill
    launch per 1,024 of them instead of one each); under stream capture a fresh zeros() — a replayed graph must zero its records
    itself."""
    device = torch.device(device)
    if device.type != "cuda" or n > 256 or torch.cuda.is_current_stream_capturing():
        return torch.zeros((n, 2), dtype=torch.int32, device=device)
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream)
    with _REC_LOCK:
        pool = _REC_POOLS.get(key)
        if pool is None or pool[1] + n > pool[0].shape[0]:
            if len(_REC_POOLS) > 64:         # streams come and go: drop the blocks of old ones (rows in use keep their block alive)
                _REC_POOLS.clear()
            pool = _REC_POOLS[key] = [torch.zeros((1024, 2), dtype=torch.int32, device=device), 0]
        r = pool[0][pool[1]:pool[1] + n]
        pool[1] += n
    return r


def _sum_rows(part):
    """[rows, C] per-workgroup column sums -> [C] (fixed order; dasv_colsum_reduce)."""
    out = torch.empty(part.shape[1], dtype=torch.float32, device=part.device)
    with torch.cuda.device(part.device):
        _lib.check(_lib.load().dasv_colsum_reduce(part.data_ptr(), part.shape[0], part.shape[1], out.data_ptr(),
                                                  torch.cuda.current_stream(part.device).cuda_stream), "dasv_colsum_reduce")
    return out


def range_of(x):
    """Range record of a tensor no kernel of this library produced (one pass over it: the input features, a caller's tensor)."""
    lib = _lib.load()
    x = x if x.is_contiguous() or x.is_contiguous(memory_format=torch.channels_last) else x.contiguous()
    rec = new_ranges(x.device)[0]
    with torch.cuda.device(x.device):
        rc = lib.dasv_range_absmax_f32(x.data_ptr(), x.numel(), rec.data_ptr(), torch.cuda.current_stream(x.device).cuda_stream)
    _lib.check(rc, "dasv_range_absmax_f32")
    return rec


class PackedConvWeights:
    """bf16 hi/lo planes [Cout][9][Cin] of a Conv2d weight, re-packed when the parameter changes.

    The packed planes are reused only for the SAME leaf `nn.Parameter` object (held by weak reference) at the same
    address and version counter, and only in passes that cannot be followed by an optimizer step (no gradient is being
    recorded for the weight).  Anything else is packed on every call: nn.DataParallel replicas (train.py:68-70
    re-broadcasts the weights on every forward; the copies are plain tensors whose version is always 0 and whose
    allocator addresses repeat), functional weights, transposed temporaries — and every TRAINING forward: torch's fused
    optimizers (`Adam(..., fused=True)`) update the parameters without moving their version counters, so a cached copy
    would silently go stale after the first step (seven small launches per step, ~1 % of it).  Writes through `.data`
    do not move the version counter either: code that updates weights that way outside training (EMA swaps, raw
    collectives on `.data`) calls `invalidate()` / `invalidate_packed_weights(module)` afterwards."""

    def __init__(self):
        self.key = None
        self.owner = None
        self.hi = self.lo = None
        self.bias = None      # the bias zero-padded to the planes' channel count (None: no bias or no padding)

    def invalidate(self):
        self.key = None
        self.owner = None

    def get(self, weight, mode="split", bias=None, pad=None, fresh=False):
        """bias: the layer's bias (split8: its maximum enters the bound a packed output is scaled by).
        pad = (cin_p, cout_p): planes of the zero-padded layer [cout_p][9][cin_p] (and `self.bias` padded to cout_p).
        fresh: a TRAINING pass (the weight's gradient is wanted — an autograd Function's forward knows that from
        ctx.needs_input_grad; grad mode is off inside it): pack now and keep nothing."""
        kind = _PACK_KIND[mode]
        leaf = isinstance(weight, torch.nn.Parameter) and weight.is_leaf
        cacheable = leaf and not fresh and not (torch.is_grad_enabled() and weight.requires_grad)
        cout, cin = weight.shape[0], weight.shape[1]
        if pad is not None and tuple(pad) == (cin, cout):
            pad = None
        key = (weight.data_ptr(), weight._version, weight.device, kind, pad,
               None if (bias is None or (kind == 0 and pad is None)) else (bias.data_ptr(), bias._version))
        if not (cacheable and key == self.key and self.owner is not None and self.owner() is weight):
            lib = _lib.load()
            w = weight.detach().contiguous(memory_format=torch.contiguous_format).float()
            self.bias = None
            if pad is not None:
                cin_p, cout_p = pad
                if cin_p < cin or cout_p < cout:
                    raise RuntimeError("padded layer %s is smaller than the layer %s" % ((cin_p, cout_p), (cin, cout)))
                w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, cin_p - cin, 0, cout_p - cout))
                if bias is not None:
                    self.bias = bias = torch.nn.functional.pad(bias.detach().float(), (0, cout_p - cout))
                cout, cin = cout_p, cin_p
            n = max(lib.dasv_conv3x3_packed_weight_bytes(cout, cin), lib.dasv_conv3x3_packed_weight_bytes_f16(cout, cin))
            nlo = max(n, lib.dasv_conv3x3_packed_weight_bytes_x8(cout, cin))   # the e4m3 plane carries its scale bytes behind it
            if not leaf or self.hi is None or self.hi.numel() != n or self.hi.device != weight.device:
                # (weights that are not a leaf Parameter — DataParallel replicas, functional weights — get fresh planes per call:
                # re-packing into shared buffers would race with a convolution still reading them on another stream)
                self.hi = torch.empty(n, dtype=torch.uint8, device=weight.device)
                self.lo = torch.empty(nlo, dtype=torch.uint8, device=weight.device)
            with torch.cuda.device(weight.device):
                stream = torch.cuda.current_stream(weight.device).cuda_stream
                if kind == 2:
                    b = None if bias is None else bias.detach().float().contiguous()
                    rc = lib.dasv_conv3x3_pack_weights_x8(w.data_ptr(), None if b is None else b.data_ptr(), self.hi.data_ptr(),
                                                          self.lo.data_ptr(), cout, cin, stream)
                elif kind == 1:
                    b = None if bias is None else bias.detach().float().contiguous()
                    rc = lib.dasv_conv3x3_pack_weights_f16(w.data_ptr(), None if b is None else b.data_ptr(), self.hi.data_ptr(), cout, cin, stream)
                else:
                    rc = lib.dasv_conv3x3_pack_weights(w.data_ptr(), self.hi.data_ptr(), self.lo.data_ptr(), cout, cin, stream)
            _lib.check(rc, "dasv_conv3x3_pack_weights")
            self.key = key if cacheable else None
            self.owner = weakref.ref(weight) if cacheable else None
        return self.hi, self.lo


# The pack kernel stages one plane row (9 * cout floats of the layer's weight, gathered) in LDS when it fits a launch's 64 KB; rows that
# do not fit go through a transposed scratch copy first when the layer is large (cin * cout above this)
_DGRAD_SCRATCH_ABOVE = 256 * 256
_DGRAD_ROW_FITS = (65536 - 64) // 36


def _pack_dgrad_x8(weight):
    """split8 planes of a layer's DATA-GRADIENT convolution (weights transposed over (out, in), flipped over (ky, kx)) packed from
    the layer's own weight [Cout,Cin,3,3] in one launch."""
    lib = _lib.load()
    cout, cin = weight.shape[0], weight.shape[1]
    n = max(lib.dasv_conv3x3_packed_weight_bytes(cin, cout), lib.dasv_conv3x3_packed_weight_bytes_f16(cin, cout))
    hi = torch.empty(n, dtype=torch.uint8, device=weight.device)
    lo = torch.empty(max(n, lib.dasv_conv3x3_packed_weight_bytes_x8(cin, cout)), dtype=torch.uint8, device=weight.device)
    # index arithmetic on the forward weight, one launch, the gathered row staged in LDS (conv42: 13 us against 26 us through a
    # transposed scratch copy; before the staging the direct form cost 215 us there and the scratch route 65 us); the scratch route
    # remains for rows too long for LDS
    scratch = torch.empty(weight.numel(), dtype=torch.float32, device=weight.device) if cout > _DGRAD_ROW_FITS and cin * cout > _DGRAD_SCRATCH_ABOVE else None
    with torch.cuda.device(weight.device):
        rc = lib.dasv_conv3x3_pack_weights_x8_dgrad_ws(weight.data_ptr(), None if scratch is None else scratch.data_ptr(), hi.data_ptr(),
                                                       lo.data_ptr(), cout, cin, torch.cuda.current_stream(weight.device).cuda_stream)
    _lib.check(rc, "dasv_conv3x3_pack_weights_x8_dgrad")
    return hi, lo


def invalidate_packed_weights(module):
    """Drop every packed copy of convolution weights held under `module` (call after writing weights through `.data`)."""
    for m in module.modules():
        packed = getattr(m, "_packed", None)
        if isinstance(packed, dict):
            for pk in packed.values():
                if isinstance(pk, PackedConvWeights):
                    pk.invalidate()


def _attach_rec(t, rec):
    """Let the range record of a gradient tensor ride on it to the kernel that consumes it (the layer below's data / weight
    gradient), together with the tensor's version counter: autograd accumulates a second consumer's gradient IN PLACE into the
    same tensor object, which keeps the attribute but bumps the version — the record is then stale and `_rec_of`
    re-measures."""
    t._dasv_rec = (rec, t._version)


def _rec_of(t):
    """The record attached by `_attach_rec` if the tensor has not been written since, else None (the caller measures)."""
    tagged = getattr(t, "_dasv_rec", None)
    if tagged is None or tagged[1] != t._version:
        return None
    return tagged[0]


def _launch(x, hi, lo, bias, cout, mode, relu=1, mask=None, colsum=None, x_rec=None, y_rec=None, span=None, x_packed=False, y_packed=False,
            mask_packed=False):
    """colsum (with mask): a dict that receives 'gb' = per-channel sums of the masked output (fixed-order partial sums
    from the kernel's epilogue, then one small sum) when the map is narrow enough for the window kernel.
    x_rec / y_rec (split8): range records of x (made here by a pass over x when the caller has none) and of y (optional).
    span (split8): x is a RAGGED batch — utterances of different lengths stacked along H in one image [1, C, rows, W] — and span
    its border table (ragged.ragged_tables): taps never cross from one utterance into the next.
    x_packed / y_packed / mask_packed (split8, dense batches): that tensor is a PACKED activation tensor (conv3x3_x8)."""
    lib = _lib.load()
    B, cin, H, W = x.shape
    y = torch.empty((B, cout, H, W), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    if mode == "split8" and x_rec is None:
        x_rec = range_of(x)
    if span is not None:
        if mode != "split8" or B != 1 or W > 80:
            raise RuntimeError("ragged batches in training run in the split8 mode (one stacked image, maps up to 80 columns)")
        with torch.cuda.device(x.device):
            stream = torch.cuda.current_stream(x.device).cuda_stream
            if mask is not None and mask_packed:
                rows = lib.dasv_conv3x3_colsum_rows(3, 1, H, W, 1) if colsum is not None else 0
                part = torch.empty((rows, cout), dtype=torch.float32, device=x.device) if rows > 0 else None
                rc = lib.dasv_conv3x3_masked_ragged_x8p_nhwc(x.data_ptr(), int(x_packed), x_rec.data_ptr(), hi.data_ptr(), lo.data_ptr(),
                                                             mask.data_ptr(), y.data_ptr(), None if y_rec is None else y_rec.data_ptr(),
                                                             None if part is None else part.data_ptr(), span.data_ptr(), H, W, cin, cout, stream)
                if part is not None and rc == 0:
                    colsum["gb"] = _sum_rows(part)
            elif mask is not None:
                if x_packed:
                    raise RuntimeError("a packed gradient operand belongs to the packed training chain (packed mask)")
                rows = lib.dasv_conv3x3_colsum_rows(3, 1, H, W, 1) if colsum is not None else 0
                part = torch.empty((rows, cout), dtype=torch.float32, device=x.device) if rows > 0 else None
                rc = lib.dasv_conv3x3_masked_ragged_x8_nhwc(x.data_ptr(), x_rec.data_ptr(), hi.data_ptr(), lo.data_ptr(), mask.data_ptr(),
                                                            y.data_ptr(), None if y_rec is None else y_rec.data_ptr(),
                                                            None if part is None else part.data_ptr(), span.data_ptr(), H, W, cin, cout, stream)
                if part is not None and rc == 0:
                    colsum["gb"] = _sum_rows(part)
            else:
                nws = 0 if y_packed else lib.dasv_conv3x3_workspace_bytes(1, H, W, cin, cout)
                ws = torch.empty(nws, dtype=torch.uint8, device=x.device) if nws else None
                rc = lib.dasv_conv3x3_bias_relu_ragged_x8_nhwc(x.data_ptr(), int(x_packed), x_rec.data_ptr(), hi.data_ptr(), lo.data_ptr(),
                                                               None if bias is None else bias.data_ptr(), y.data_ptr(), int(y_packed),
                                                               None if y_rec is None else y_rec.data_ptr(), span.data_ptr(),
                                                               None if ws is None else ws.data_ptr(), nws, H, W, cin, cout, relu, stream)
        _lib.check(rc, "dasv_conv3x3_*_ragged_x8_nhwc")
        return y
    if (x_packed or y_packed or mask_packed) and mode != "split8":
        raise RuntimeError("packed activation tensors belong to the split8 mode")
    with torch.cuda.device(x.device):
        stream = torch.cuda.current_stream(x.device).cuda_stream
        if x_packed and mask is not None and not mask_packed:
            raise RuntimeError("a packed gradient operand belongs to the packed training chain (packed mask)")
        if mode == "split8" and (x_packed or y_packed) and mask is None:
            rc = lib.dasv_conv3x3_bias_relu_x8_nhwc(x.data_ptr(), int(x_packed), x_rec.data_ptr(), hi.data_ptr(), lo.data_ptr(),
                                                    None if bias is None else bias.data_ptr(), y.data_ptr(), int(y_packed),
                                                    None if y_rec is None else y_rec.data_ptr(), None, 0, B, H, W, cin, cout, relu, stream)
        elif mode == "split8" and mask is not None and mask_packed:
            rows = lib.dasv_conv3x3_colsum_rows(3, B, H, W, 0) if colsum is not None else 0
            part = torch.empty((rows, cout), dtype=torch.float32, device=x.device) if rows > 0 else None
            nws = lib.dasv_conv3x3_workspace_bytes(B, H, W, cin, cout) if part is None else 0      # (column sums: one pass)
            ws = torch.empty(nws, dtype=torch.uint8, device=x.device) if nws else None
            rc = lib.dasv_conv3x3_masked_x8p_nhwc(x.data_ptr(), int(x_packed), x_rec.data_ptr(), hi.data_ptr(), lo.data_ptr(), mask.data_ptr(), y.data_ptr(),
                                                  None if y_rec is None else y_rec.data_ptr(), None if part is None else part.data_ptr(),
                                                  None if ws is None else ws.data_ptr(), nws, B, H, W, cin, cout, stream)
            if part is not None and rc == 0:
                colsum["gb"] = _sum_rows(part)
        elif mode == "split8" and mask is not None:
            rows = lib.dasv_conv3x3_colsum_rows(3, B, H, W, 0) if colsum is not None else 0
            part = torch.empty((rows, cout), dtype=torch.float32, device=x.device) if rows > 0 else None
            rc = lib.dasv_conv3x3_masked_x8_nhwc(x.data_ptr(), x_rec.data_ptr(), hi.data_ptr(), lo.data_ptr(), mask.data_ptr(), y.data_ptr(),
                                                 None if y_rec is None else y_rec.data_ptr(), None if part is None else part.data_ptr(),
                                                 B, H, W, cin, cout, stream)
            if part is not None and rc == 0:
                colsum["gb"] = _sum_rows(part)
        elif mode == "split8":
            nws = lib.dasv_conv3x3_workspace_bytes(B, H, W, cin, cout)
            ws = torch.empty(nws, dtype=torch.uint8, device=x.device) if nws else None
            rc = lib.dasv_conv3x3_bias_relu_x8_nhwc(x.data_ptr(), 0, x_rec.data_ptr(), hi.data_ptr(), lo.data_ptr(),
                                                    None if bias is None else bias.data_ptr(), y.data_ptr(), 0,
                                                    None if y_rec is None else y_rec.data_ptr(),
                                                    None if ws is None else ws.data_ptr(), nws, B, H, W, cin, cout, relu, stream)
        elif mask is not None:   # y = conv(x) where mask > 0 else 0 (no bias, no ReLU): the fused data-gradient form
            rows = lib.dasv_conv3x3_colsum_rows(MODES[mode], B, H, W, 0) if colsum is not None else 0
            part = torch.empty((rows, cout), dtype=torch.float32, device=x.device) if rows > 0 else None
            rc = lib.dasv_conv3x3_masked_nhwc(MODES[mode], x.data_ptr(), hi.data_ptr(), lo.data_ptr(), mask.data_ptr(),
                                              y.data_ptr(), None if part is None else part.data_ptr(), B, H, W, cin, cout, stream)
            if part is not None and rc == 0:
                colsum["gb"] = _sum_rows(part)
        else:
            # small maps (single utterances): split-K over the channel slabs, partial tiles in a workspace
            nws = lib.dasv_conv3x3_workspace_bytes(B, H, W, cin, cout)
            ws = torch.empty(nws, dtype=torch.uint8, device=x.device) if nws else None
            rc = lib.dasv_conv3x3_bias_relu_nhwc_ws(MODES[mode], x.data_ptr(), hi.data_ptr(), lo.data_ptr(),
                                                    None if bias is None else bias.data_ptr(), y.data_ptr(),
                                                    None if ws is None else ws.data_ptr(), nws, B, H, W, cin, cout, relu, stream)
    _lib.check(rc, "dasv_conv3x3_bias_relu_nhwc")
    return y


def supported_wgrad(cin, cout, width=1, pixels=0):
    """Shapes the weight-gradient kernel takes: 64-channel tiles, maps up to 80 columns, operands below 4 GB
    (`pixels` = B*H*W; the kernel addresses them through 32-bit buffer offsets)."""
    return cin % 64 == 0 and cout % 64 == 0 and width <= 80 and pixels * max(cin, cout) * 4 < 0xffffffff


def conv3x3_wgrad(x, g, mode="split", x_rec=None, g_rec=None, borders=None, x_packed=False, g_packed=False):
    """Weight gradient [Cout,Cin,3,3] of conv3x3 (stride 1, pad 1) from x [B,Cin,H,W] and g [B,Cout,H,W]
    (both channels_last fp32) on the hand-written MFMA kernel (csrc/conv3x3_wgrad.hip).
    mode "split8": fp16 products + block-scaled e4m3 cross terms; x_rec / g_rec are the operands' range records (an upper bound
    of max |v| is enough; measured here with one pass over the tensor when None).
    borders (ragged batch: x / g are utterances stacked along H in one image, B == 1): int64 tensor of the first rows of the
    utterances 1 ... n-1.  The launch on the stacked image also multiplies across every border — g of an utterance's first row with x of
    the previous utterance's last row (taps ky = 0) and g of a last row with x of the next first row (ky = 2); exactly those products
    are computed a second time from one 2-row image {last row | first row} per border and subtracted.
    x_packed (split8): x is a PACKED activation tensor and x_rec the record its producer filled — the kernel's producer waves
    then copy instead of converting (dasv_conv3x3_wgrad_x8p_nhwc); g_packed (with x_packed): g is packed too (the pool's backward
    wrote it that way), g_rec its record."""
    if x_packed and (mode != "split8" or x_rec is None):
        raise RuntimeError("a packed x operand: split8 mode, with the record its producer filled")
    if g_packed and not (x_packed and g_rec is not None):
        raise RuntimeError("a packed g operand: with a packed x, and the record its producer filled")
    if borders is not None and borders.numel() > 0:
        if x.shape[0] != 1:
            raise RuntimeError("a ragged batch is one stacked image")
        tall = conv3x3_wgrad(x, g, mode, x_rec, g_rec, x_packed=x_packed, g_packed=g_packed)
        # (packed operands: whole pixel rows are gathered, so their slabs stay intact, and zero bits are a packed zero)
        x3, g3 = x.permute(0, 2, 3, 1)[0], g.permute(0, 2, 3, 1)[0]              # [rows, W, C] views of the channels_last memory
        # ONE 2-row image per border, {last row of utterance i | first row of utterance i + 1} for x and for g alike: its ky = 0 taps are
        # exactly g_first x_last (row 1 looking up at row 0), its ky = 2 taps g_last x_first (row 0 looking down at row 1) — the two
        # cross-border products of the stacked launch — and its ky = 1 taps (each row with itself: genuine products) are dropped
        idx = torch.stack([borders - 1, borders], dim=1).reshape(-1)
        xb = x3[idx].view(borders.numel(), 2, x3.shape[1], x3.shape[2]).permute(0, 3, 1, 2)                   # [nb, C, 2, W] channels_last
        gb = g3[idx].view(borders.numel(), 2, g3.shape[1], g3.shape[2]).permute(0, 3, 1, 2)
        if mode == "split8":           # the same scales as the stacked launch (the records bound these rows too)
            x_rec = range_of(x) if x_rec is None else x_rec
            g_rec = range_of(g) if g_rec is None else g_rec
        cross = conv3x3_wgrad(xb, gb, mode, x_rec, g_rec, x_packed=x_packed, g_packed=g_packed)
        cross[:, :, 1, :] = 0
        return tall - cross
    lib = _lib.load()
    B, cin, H, W = x.shape
    cout = g.shape[1]
    if not x.is_contiguous(memory_format=torch.channels_last):
        x = x.contiguous(memory_format=torch.channels_last)
    if not g.is_contiguous(memory_format=torch.channels_last):
        if g_packed:
            raise RuntimeError("a packed tensor must keep its channels_last layout")
        g = g.contiguous(memory_format=torch.channels_last)
    gw = torch.empty(cout, cin, 3, 3, dtype=torch.float32, device=x.device)
    nbytes = lib.dasv_conv3x3_wgrad_workspace_bytes(B, H, W, cin, cout)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    if mode == "split8":
        x_rec = range_of(x) if x_rec is None else x_rec
        g_rec = range_of(g) if g_rec is None else g_rec
        with torch.cuda.device(x.device):
            st = torch.cuda.current_stream(x.device).cuda_stream
            if x_packed:
                rc = lib.dasv_conv3x3_wgrad_x8p_nhwc(x.data_ptr(), x_rec.data_ptr(), g.data_ptr(), int(g_packed), g_rec.data_ptr(), gw.data_ptr(),
                                                     ws.data_ptr(), nbytes, B, H, W, cin, cout, st)
            else:
                rc = lib.dasv_conv3x3_wgrad_x8_nhwc(x.data_ptr(), x_rec.data_ptr(), g.data_ptr(), g_rec.data_ptr(), gw.data_ptr(), ws.data_ptr(),
                                                    nbytes, B, H, W, cin, cout, st)
        _lib.check(rc, "dasv_conv3x3_wgrad_x8_nhwc")
        return gw
    with torch.cuda.device(x.device):
        rc = lib.dasv_conv3x3_wgrad_nhwc(MODES[mode], x.data_ptr(), g.data_ptr(), gw.data_ptr(), ws.data_ptr(), nbytes,
                                         B, H, W, cin, cout, torch.cuda.current_stream(x.device).cuda_stream)
    _lib.check(rc, "dasv_conv3x3_wgrad_nhwc")
    return gw


class _ConvReLUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, packed, mode, grad_premasked, mask_input_grad, link_in, link_out, x_rec, y_rec, span=None, borders=None,
                fresh=False, x_packed=False, y_packed=False, g_packed=False):
        # x carries cin_x >= Cin channels (a multiple of 64; the surplus is zero) and y gets padded_channels(Cout): the layer runs
        # as its zero-padded form (see `supported`)
        cout, cin = weight.shape[0], weight.shape[1]
        cin_x, cout_p = x.shape[1], padded_channels(cout)
        hi, lo = packed.get(weight, mode, bias, (cin_x, cout_p), fresh=fresh)
        bias_k = packed.bias if (cin_x, cout_p) != (cin, cout) and bias is not None else bias
        if mode == "split8" and x_rec is None:
            x_rec = range_of(x)
        if y_packed and not grad_premasked:
            raise RuntimeError("a packed output's ReLU mask is applied by its consumer (grad_premasked)")
        y = _launch(x, hi, lo, bias_k, cout_p, mode, x_rec=x_rec, y_rec=y_rec, span=span, x_packed=x_packed, y_packed=y_packed)
        ctx.span, ctx.borders = span, borders           # ragged batch (one stacked image): border tables of this level
        if g_packed and not grad_premasked:
            raise RuntimeError("a packed incoming gradient comes pre-masked from the pool's backward")
        ctx.x_packed, ctx.g_packed = x_packed, g_packed
        ctx.save_for_backward(x, weight, y)
        ctx.x_rec = x_rec           # (split8) the weight gradient scales x by the same record
        ctx.has_bias = bias is not None
        ctx.mode = mode
        ctx.grad_premasked = grad_premasked
        ctx.mask_input_grad = mask_input_grad
        ctx.link_in, ctx.link_out = link_in, link_out
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        gy = gy.contiguous(memory_format=torch.channels_last)
        # ReLU backward in one pass (g = gy where y > 0 else 0) — unless the consumer of y already applied it
        g = gy if ctx.grad_premasked else torch.ops.aten.threshold_backward(gy, y, 0.0)
        return _conv_backward(ctx, x, weight, g, _rec_of(gy), y.shape[1], g_packed=ctx.g_packed) + (None,) * 14


def _conv_backward(ctx, x, weight, g, g_rec_in, cout_p, g_packed=False):
    """(gx, gw, gb) of a conv3x3 + bias layer from g = the gradient at its pre-activation [B, cout_p, H, W] (channels_last fp32; g_rec_in
    its range record or None) — shared by the layer's own autograd node and by the conv + ReLU + max-pool node.
    g_packed (packed training chain): g is a packed tensor (the pool's backward wrote it so), g_rec_in its record."""
    cout, cin = weight.shape[0], weight.shape[1]
    cin_x = x.shape[1]        # (with cout_p: the padded layer that ran forward)
    padded = (cin_x, cout_p) != (cin, cout)
    gx = None
    bmode = BACKWARD_MODE[ctx.mode]
    dmode = DGRAD_MODE[ctx.mode] if x.shape[3] <= 80 else bmode       # (the split8 form exists in the window kernel only)
    # split8 data gradient: the incoming gradient's range record rides on the tensor when a kernel of this library produced it
    # (the layer above's data gradient, the pool's backward); otherwise one pass over it measures it
    g_rec = g_rec_in if dmode == "split8" else None
    gx_rec = new_ranges(x.device)[0] if dmode == "split8" else None
    want_x, want_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
    want_b = ctx.has_bias and ctx.needs_input_grad[2]
    if want_x and g.dtype == torch.float32:
        # data gradient of a 3x3 / stride 1 / pad 1 convolution = the same convolution of g with the weights
        # transposed over (Cout, Cin) and flipped over (ky, kx): reuse the forward implicit-GEMM kernel
        # (the kernel writes 128-channel tiles: an input with a 64-multiple of channels gets its gradient through a padded
        # transposed layer and a slice — narrow kernel sizes only)
        n_out = (cin_x + 127) // 128 * 128
        if dmode == "split8" and not padded and n_out == cin_x and weight.dtype == torch.float32 and weight.is_contiguous():
            # straight from the layer's weight (conv._pack_dgrad_x8): no torch flip / copy / temporaries
            hi, lo = _pack_dgrad_x8(weight.detach())
        else:
            wt = weight.detach()
            if padded or n_out != cin_x:
                wt = torch.nn.functional.pad(wt, (0, 0, 0, 0, 0, n_out - cin, 0, cout_p - cout))
            wt = wt.transpose(0, 1).flip(2, 3).contiguous()
            pk = PackedConvWeights()
            hi, lo = pk.get(wt, dmode)
        if n_out == cin_x:
            # mask_input_grad: x is the ReLU output of the layer below (its only consumer is this layer), so that ReLU's
            # backward is applied in this kernel's epilogue
            gx = _launch(g, hi, lo, None, cin_x, dmode, relu=0, mask=x if ctx.mask_input_grad else None,
                         colsum=ctx.link_out if ctx.mask_input_grad else None, x_rec=g_rec, y_rec=gx_rec, span=ctx.span,
                         mask_packed=ctx.x_packed and ctx.mask_input_grad, x_packed=g_packed)
            if gx_rec is not None:
                _attach_rec(gx, gx_rec)
        else:
            if ctx.x_packed:
                raise RuntimeError("packed training chain: input channels must be a multiple of 128")
            gx = _launch(g, hi, lo, None, n_out, dmode, relu=0, x_rec=g_rec, span=ctx.span)[:, :cin_x].contiguous(memory_format=torch.channels_last)
            if ctx.mask_input_grad:
                gx = torch.ops.aten.threshold_backward(gx, x, 0.0)
        want_x = False
    gw = gb = None
    if want_w and USE_HIP_WGRAD and supported_wgrad(cin_x, cout_p, x.shape[3], x.shape[0] * x.shape[2] * x.shape[3]) and g.dtype == torch.float32:
        wmode = WGRAD_MODE[ctx.mode]
        # (the record of the gradient BEFORE this layer's ReLU mask bounds the masked one)
        gw = conv3x3_wgrad(x, g, wmode, x_rec=ctx.x_rec if wmode == "split8" else None,
                           g_rec=g_rec_in if wmode == "split8" else None, borders=ctx.borders if ctx.span is not None else None,
                           x_packed=ctx.x_packed, g_packed=g_packed)
        if padded:
            gw = gw[:cout, :cin].contiguous()
        want_w = False
        if want_b:
            # the kernel that produced this (pre-masked) gradient may have left its per-channel sums behind
            gb = ctx.link_in.pop("gb", None) if (ctx.grad_premasked and ctx.link_in is not None) else None
            if gb is None:
                if g_packed:
                    raise RuntimeError("packed training chain: the bias gradient comes from the pool's backward")
                gb = g.sum(dim=(0, 2, 3))
            gb = gb[:cout] if padded else gb
            want_b = False
    if (want_x or want_w or want_b) and (ctx.span is not None or ctx.x_packed):
        raise RuntimeError("ragged / packed training batch: a gradient fell outside the hand-written kernels' limits (operands beyond 4 GB?)")
    if want_x or want_w or want_b:
        # (maps wider than 80 columns / operands beyond 4 GB only: the weight-gradient kernel's limits)
        wfull = torch.nn.functional.pad(weight, (0, 0, 0, 0, 0, cin_x - cin, 0, cout_p - cout)) if padded else weight
        gx2, gw2, gb2 = torch.ops.aten.convolution_backward(
            g, x, wfull, [cout_p] if ctx.has_bias else None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
            [want_x, want_w, want_b])
        if want_x:
            gx = torch.ops.aten.threshold_backward(gx2, x, 0.0) if ctx.mask_input_grad else gx2
        gw = (gw2[:cout, :cin].contiguous() if padded else gw2) if want_w else gw
        gb = (gb2[:cout] if padded else gb2) if want_b else gb
    return gx, gw, gb


def _launch_pool_idx(x, hi, lo, bias, cout, x_rec, y_rec, x_packed, y_packed):
    """conv3x3 + bias + ReLU + 2x2 ceil max-pool in one launch, training form: (pooled map, arg-max bytes)."""
    lib = _lib.load()
    B, cin, H, W = x.shape
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    y = torch.empty((B, cout, Ho, Wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    idx = torch.empty((B, Ho, Wo, cout), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.dasv_conv3x3_bias_relu_pool_idx_x8_nhwc(x.data_ptr(), int(x_packed), x_rec.data_ptr(), hi.data_ptr(), lo.data_ptr(),
                                                         None if bias is None else bias.data_ptr(), y.data_ptr(), int(y_packed),
                                                         None if y_rec is None else y_rec.data_ptr(), idx.data_ptr(), B, H, W, cin, cout,
                                                         torch.cuda.current_stream(x.device).cuda_stream)
    _lib.check(rc, "dasv_conv3x3_bias_relu_pool_idx_x8_nhwc")
    return y, idx


def _launch_dgrad_c1(g, hi, lo, cout, g_rec, mask, mel, span=None):
    """conv12's data-gradient launch (g packed, mask = conv11's packed output) with conv11's backward in its write-out pass:
    the per-tile partial rows [tiles][cout][10].  span: a ragged batch (one stacked image, border table)."""
    lib = _lib.load()
    B, cin, H, W = g.shape
    rows = lib.dasv_conv3x3_colsum_rows(3, B, H, W, 0 if span is None else 1)
    cpart = torch.empty((rows, cout, 10), dtype=torch.float32, device=g.device)
    with torch.cuda.device(g.device):
        st = torch.cuda.current_stream(g.device).cuda_stream
        if span is not None:
            rc = lib.dasv_conv3x3_masked_c1_ragged_x8p_nhwc(g.data_ptr(), 1, g_rec.data_ptr(), hi.data_ptr(), lo.data_ptr(), mask.data_ptr(), 1,
                                                            mel.data_ptr(), cpart.data_ptr(), span.data_ptr(), H, W, cin, cout, st)
        else:
            rc = lib.dasv_conv3x3_masked_c1_x8p_nhwc(g.data_ptr(), 1, g_rec.data_ptr(), hi.data_ptr(), lo.data_ptr(), mask.data_ptr(), 1,
                                                 mel.data_ptr(), cpart.data_ptr(), B, H, W, cin, cout, st)
    _lib.check(rc, "dasv_conv3x3_masked_c1_x8p_nhwc")
    return cpart


def _pool_scatter(idx, gy, B, H, W, C, want_sum, packed):
    """The pool's backward from the forward's arg-max bytes (which carry the ReLU's mask too): (g [B,C,H,W], its range record or None,
    per-workgroup column sums or None).  packed: g is written as a PACKED tensor scaled by the exponent of gy's maximum (the
    scattered tensor's own maximum) — its consumers, the layer's data- and weight-gradient launches, then copy its pieces."""
    lib = _lib.load()
    dev = gy.device
    g = torch.empty((B, C, H, W), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
    g_rec = _rec_of(gy)
    with torch.cuda.device(dev):
        st = torch.cuda.current_stream(dev).cuda_stream
        rows = lib.dasv_maxpool2x2_bwd_colsum_rows(B, H, W, C) if want_sum else 0
        part = torch.empty((rows, C), dtype=torch.float32, device=dev) if rows > 0 else None
        if packed:
            if g_rec is None:
                g_rec = range_of(gy)           # (a gradient no kernel of this library produced: one pass over the POOLED tensor)
            rec = new_ranges(dev)[0]
            rc = lib.dasv_maxpool2x2_ceil_bwd_idx_x8_nhwc(idx.data_ptr(), gy.data_ptr(), g_rec.data_ptr(), g.data_ptr(), rec.data_ptr(),
                                                          None if part is None else part.data_ptr(), 1, B, H, W, C, st)
            g_rec = rec
        else:
            # (the scattered gradient's range record is its source's: max |g| <= max |gy|)
            rc = lib.dasv_maxpool2x2_ceil_bwd_idx_nhwc(idx.data_ptr(), gy.data_ptr(), g.data_ptr(), None if part is None else part.data_ptr(),
                                                       1, B, H, W, C, st)
    _lib.check(rc, "dasv_maxpool2x2_ceil_bwd_idx_nhwc")
    return g, g_rec, part


class _ConvReLUPoolFn(torch.autograd.Function):
    """`F.max_pool2d(F.relu(convX2(x)), 2, stride=2, ceil_mode=True)` (CNNs.py:73-74 ... 85-86) as ONE forward launch in training
    (dasv_conv3x3_bias_relu_pool_idx_x8_nhwc): the pooled map plus one byte per pooled element {arg-max | 4 if the maximum is > 0}.
    The backward scatters the pooled gradient through those bytes (which apply the ReLU's mask too and leave the bias gradient
    behind), then runs the layer's weight- and data-gradient launches; the full-resolution activation map never exists."""

    @staticmethod
    def forward(ctx, x, weight, bias, packed, mask_input_grad, link_out, x_rec, y_rec, fresh=True, x_packed=False, y_packed=False):
        hi, lo = packed.get(weight, "split8", bias, fresh=fresh)
        if x_rec is None:
            x_rec = range_of(x)
        y, idx = _launch_pool_idx(x, hi, lo, bias, weight.shape[0], x_rec, y_rec, x_packed, y_packed)
        ctx.save_for_backward(x, weight, idx)
        ctx.x_rec, ctx.has_bias, ctx.mode = x_rec, bias is not None, "split8"
        ctx.grad_premasked, ctx.mask_input_grad = True, mask_input_grad
        ctx.link_in, ctx.link_out = {}, link_out
        ctx.span = ctx.borders = None
        ctx.x_packed = x_packed
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, idx = ctx.saved_tensors
        lib = _lib.load()
        B, _, H, W = x.shape
        C = weight.shape[0]
        gy = gy.contiguous(memory_format=torch.channels_last)
        g, g_rec, part = _pool_scatter(idx, gy, B, H, W, C, ctx.has_bias, packed=ctx.x_packed)
        if part is not None:
            ctx.link_in["gb"] = _sum_rows(part)
        return _conv_backward(ctx, x, weight, g, g_rec, C, g_packed=ctx.x_packed) + (None,) * 8


class _FirstBlockFn(torch.autograd.Function):
    """The first VGG block of the packed training chain as ONE autograd node — `pool(relu(conv12(relu(conv11(mel)))))`, CNNs.py:38-40 /
    72-74: forward = the one-channel kernel (packed output) + the conv + ReLU + pool launch with arg-max bytes; backward = the pool's
    scatter, conv12's weight gradient from the packed activations, and conv12's data-gradient launch with conv11's WHOLE backward in its
    write-out pass (dasv_conv3x3_masked_c1_x8p_nhwc): the gradient at conv11's output — the largest tensor of the step — is never
    written."""

    @staticmethod
    def forward(ctx, mel, w11, b11, w12, b12, packed12, x_rec, h_rec, y_rec, y_packed, fresh):
        h = conv3x3_c1_bias_relu(mel, w11, b11, pack=True, x_rec=x_rec, y_rec=h_rec)
        hi, lo = packed12.get(w12, "split8", b12, fresh=fresh)
        y, idx = _launch_pool_idx(h, hi, lo, b12, w12.shape[0], h_rec, y_rec, True, y_packed)
        ctx.save_for_backward(mel.contiguous(), h, w12, idx)
        ctx.h_rec, ctx.bias = h_rec, (b11 is not None, b12 is not None)
        ctx.w11_shape = w11.shape
        return y

    @staticmethod
    def backward(ctx, gy):
        mel, h, w12, idx = ctx.saved_tensors
        lib = _lib.load()
        B, C1, H, W = h.shape
        C2 = w12.shape[0]
        dev = gy.device
        gy = gy.contiguous(memory_format=torch.channels_last)
        g, g_rec, part = _pool_scatter(idx, gy, B, H, W, C2, ctx.bias[1], packed=True)
        with torch.cuda.device(dev):
            st = torch.cuda.current_stream(dev).cuda_stream
            gw12 = conv3x3_wgrad(h, g, "split8", x_rec=ctx.h_rec, g_rec=g_rec, x_packed=True, g_packed=True) if ctx.needs_input_grad[3] else None
            gb12 = _sum_rows(part) if part is not None and ctx.needs_input_grad[4] else None
            gw11 = gb11 = None
            if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
                hi, lo = _pack_dgrad_x8(w12.detach())
                cpart = _launch_dgrad_c1(g, hi, lo, C1, g_rec, h, mel)
                prow = cpart.shape[0]
                gw11 = torch.empty(ctx.w11_shape, dtype=torch.float32, device=dev)
                gb11 = torch.empty(C1, dtype=torch.float32, device=dev) if ctx.bias[0] else None
                _lib.check(lib.dasv_conv3x3_c1_bwd_reduce_parts(cpart.data_ptr(), prow, gw11.data_ptr(), None if gb11 is None else gb11.data_ptr(),
                                                                C1, st), "dasv_conv3x3_c1_bwd_reduce_parts")
        return None, gw11, gb11, gw12, gb12, None, None, None, None, None, None


class _FirstBlockRaggedFn(torch.autograd.Function):
    """conv11 + conv12 of the packed training chain over a RAGGED batch (one stacked image, border tables) as one autograd node: forward =
    the one-channel kernel (packed output) + conv12 (packed in, fp32 out: the pool that follows is its own launch under border tables);
    backward = conv12's weight gradient (stacked image minus the cross-border products) and its data-gradient launch with conv11's whole
    backward in the write-out pass (dasv_conv3x3_masked_c1_ragged_x8p_nhwc) — the gradient at conv11's output is never written.  The
    incoming gradient is the packed, pre-masked tensor the pool's backward writes (maxpool2x2_ceil_autograd(..., pack_grad=True)), which
    also leaves conv12's bias gradient in `link_in`."""

    @staticmethod
    def forward(ctx, mel, w11, b11, w12, b12, packed12, x_rec, h_rec, y_rec, span, borders, link_in, fresh):
        lib = _lib.load()
        mel = mel.contiguous()
        _, H, W = mel.shape
        C1, C2 = w11.shape[0], w12.shape[0]
        h = torch.empty((1, C1, H, W), dtype=torch.float32, device=mel.device, memory_format=torch.channels_last)
        with torch.cuda.device(mel.device):
            rc = lib.dasv_conv3x3_c1_bias_relu_ragged_x8_nhwc(mel.data_ptr(), x_rec.data_ptr(), w11.detach().contiguous().data_ptr(),
                                                              None if b11 is None else b11.detach().data_ptr(), h.data_ptr(), h_rec.data_ptr(),
                                                              span.data_ptr(), H, W, C1, 1, torch.cuda.current_stream(mel.device).cuda_stream)
        _lib.check(rc, "dasv_conv3x3_c1_bias_relu_ragged_x8_nhwc")
        hi, lo = packed12.get(w12, "split8", b12, fresh=fresh)
        y = _launch(h, hi, lo, b12, C2, "split8", x_rec=h_rec, y_rec=y_rec, span=span, x_packed=True)
        ctx.save_for_backward(mel, h, w12)
        ctx.h_rec, ctx.span, ctx.borders, ctx.link_in = h_rec, span, borders, link_in
        ctx.bias = (b11 is not None, b12 is not None)
        ctx.w11_shape = w11.shape
        return y

    @staticmethod
    def backward(ctx, gy):
        mel, h, w12 = ctx.saved_tensors
        lib = _lib.load()
        C1, dev = h.shape[1], gy.device
        g_rec = _rec_of(gy)
        if g_rec is None or not gy.is_contiguous(memory_format=torch.channels_last):
            raise RuntimeError("the ragged first block expects the packed gradient of maxpool2x2_ceil_autograd(..., pack_grad=True)")
        gw12 = conv3x3_wgrad(h, gy, "split8", x_rec=ctx.h_rec, g_rec=g_rec, borders=ctx.borders, x_packed=True, g_packed=True) \
            if ctx.needs_input_grad[3] else None
        gb12 = ctx.link_in.pop("gb", None) if (ctx.bias[1] and ctx.needs_input_grad[4]) else None
        gw11 = gb11 = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            hi, lo = _pack_dgrad_x8(w12.detach())
            cpart = _launch_dgrad_c1(gy, hi, lo, C1, g_rec, h, mel, span=ctx.span)
            gw11 = torch.empty(ctx.w11_shape, dtype=torch.float32, device=dev)
            gb11 = torch.empty(C1, dtype=torch.float32, device=dev) if ctx.bias[0] else None
            with torch.cuda.device(dev):
                _lib.check(lib.dasv_conv3x3_c1_bwd_reduce_parts(cpart.data_ptr(), cpart.shape[0], gw11.data_ptr(),
                                                                None if gb11 is None else gb11.data_ptr(), C1,
                                                                torch.cuda.current_stream(dev).cuda_stream), "dasv_conv3x3_c1_bwd_reduce_parts")
        return (None, gw11, gb11, gw12, gb12) + (None,) * 8


def first_block_ragged_train(mel, w11, b11, w12, b12, packed12, x_rec, h_rec, y_rec, span, borders, link_in):
    """Differentiable (w.r.t. the four parameters) conv11 + conv12 on packed activations over a ragged batch: mel [1, rows, W] (utterances
    stacked along the rows) -> conv12's fp32 map [1, C, rows, W]; its one consumer is maxpool2x2_ceil_autograd(..., relu_input=True,
    link_out=link_in, pack_grad=True)."""
    fresh = torch.is_grad_enabled() and w12.requires_grad
    return _FirstBlockRaggedFn.apply(mel, w11, b11, w12, b12, packed12, x_rec, h_rec, y_rec, span, borders, link_in, fresh)


def first_block_train(mel, w11, b11, w12, b12, packed12, x_rec, h_rec, y_rec, y_packed=True):
    """Differentiable (w.r.t. the four parameters) first block on packed activations: mel [B,H,W] fp32 -> the pooled map
    [B, C, ceil(H/2), ceil(W/2)] (packed when y_packed); x_rec = the record of mel (filled), h_rec / y_rec = fresh records.  Channel counts:
    conv11's output a power of two in [128, 1024], conv12 exact (conv.exact), the shape pool_fusable."""
    fresh = torch.is_grad_enabled() and w12.requires_grad
    return _FirstBlockFn.apply(mel, w11, b11, w12, b12, packed12, x_rec, h_rec, y_rec, y_packed, fresh)


def conv3x3_bias_relu_pool(x, weight, bias, packed=None, mask_input_grad=False, link_out=None, x_rec=None, y_rec=None, x_packed=False,
                           y_packed=False):
    """Differentiable `max_pool2d(relu(conv2d(x, weight, bias, padding 1)), 2, stride=2, ceil_mode=True)` in the split8 mode for shapes
    `pool_fusable("split8", ...)` accepts (x [B,Cin,H,W] channels_last fp32): one forward launch, no full-resolution map.
    mask_input_grad / link_out / x_rec / y_rec / x_packed / y_packed as in conv3x3_bias_relu (y: the pooled map)."""
    if not x.is_cuda or x.dtype != torch.float32:
        raise RuntimeError("conv3x3_bias_relu_pool runs on fp32 GPU tensors only (libdasv_hip.so); no CPU fallback")
    if (x_packed and (x_rec is None or weight.shape[1] % 128 != 0)) or (y_packed and y_rec is None):
        raise RuntimeError("packed activations: Cin % 128 == 0 and the range records of the packed tensors")
    if not x.is_contiguous(memory_format=torch.channels_last):
        if x_packed:
            raise RuntimeError("a packed activation tensor must keep its channels_last layout")
        x = x.contiguous(memory_format=torch.channels_last)
    fresh = torch.is_grad_enabled() and weight.requires_grad
    return _ConvReLUPoolFn.apply(x, weight, bias, packed or PackedConvWeights(), mask_input_grad, link_out, x_rec, y_rec, fresh, x_packed, y_packed)


def conv3x3_bias_relu(x, weight, bias, packed=None, mode="split", grad_premasked=False, mask_input_grad=False,
                      link_in=None, link_out=None, x_rec=None, y_rec=None, span=None, borders=None, x_packed=False, y_packed=False,
                      grad_packed=False):
    """relu(conv2d(x, weight, bias, stride 1, padding 1)) for x [B,Cin,H,W] fp32 on the GPU; returns channels_last.

    Two backward fusions a caller that knows the graph may switch on (the VGG front-end does; results are identical):
      grad_premasked   the one consumer of the returned tensor multiplies the gradient it sends back by (y > 0) itself
                       (a following conv3x3_bias_relu(..., mask_input_grad=True) or maxpool2x2_ceil_autograd(...,
                       relu_input=True)), so this layer skips its own ReLU-mask pass;
      mask_input_grad  x is the ReLU output of a layer that was called with grad_premasked=True: the gradient w.r.t. x
                       is multiplied by (x > 0) in the data-gradient kernel's epilogue.
    link_in / link_out are plain dicts shared between neighbouring calls: the kernel that produces a pre-masked gradient
    also leaves its per-channel sums (= the producer layer's bias gradient) in `link_out["gb"]`; the layer that receives
    that gradient passes the same dict as `link_in` and skips its own reduction pass.
    x_rec / y_rec (mode "split8"): range records (new_ranges) of x — measured here with one pass over x when None — and of y.
    span / borders (training on a RAGGED batch, split8): x is [1, C, rows, W] — utterances of different lengths stacked along H — with
    its border table (ragged.ragged_tables) and the first rows of utterances 1 ... n-1 (int64): forward, data gradient and weight
    gradient then treat every utterance as its own image.
    x_packed / y_packed (training on PACKED activations, split8, dense batches, exact channel counts with Cin % 128 == 0): x and / or
    the result are packed activation tensors (conv3x3_x8) with their records in x_rec / y_rec — the weight gradient then reads x
    without converting it and the data-gradient kernel takes the ReLU mask from x's fp16 plane; a packed result requires
    grad_premasked (its consumer applies the mask).  grad_packed: the gradient that will arrive at the result is a packed tensor (the
    consumer is maxpool2x2_ceil_autograd(..., pack_grad=True)).
    Channel counts that are not multiples of (64, 128): x may carry more channels than the layer has inputs (zeros), and the
    result carries padded_channels(Cout) channels, the surplus identically zero (see `supported`); slice when needed."""
    if not x.is_cuda:
        raise RuntimeError("conv3x3_bias_relu runs on the GPU only (libdasv_hip.so); no CPU fallback")
    if x.dtype != torch.float32:
        raise RuntimeError("unsupported conv3x3 dtype for the HIP kernel: %s %s" % (tuple(weight.shape), x.dtype))
    cin = weight.shape[1]
    if x.shape[1] < cin or x.shape[1] % 64 != 0:
        # the caller's tensor has the layer's own (non-multiple-of-64) channel count: zero-pad it (one pass over x; the VGG
        # front-end keeps its activations padded between the layers instead, CNNs.py)
        x = torch.nn.functional.pad(x, (0, 0, 0, 0, 0, (cin + 63) // 64 * 64 - x.shape[1]))
    if x_packed or y_packed:
        if (mode != "split8" or x.shape[3] > 80 or not exact(cin, weight.shape[0]) or cin % 128 != 0 or x.shape[1] != cin
                or (x_packed and x_rec is None) or (y_packed and y_rec is None) or not x.is_contiguous(memory_format=torch.channels_last)):
            raise RuntimeError("packed activations: split8 mode, maps up to 80 columns, exact channel counts (Cin % 128 == 0), range records")
    if not x.is_contiguous(memory_format=torch.channels_last):
        x = x.contiguous(memory_format=torch.channels_last)
    if mode == "split8" and x.shape[3] > 80:
        mode = "split"   # the fp16 + e4m3 form exists in the window kernel (maps up to 80 columns) only
    packed = packed or PackedConvWeights()
    b = None if bias is None else bias.detach().float().contiguous() if not bias.requires_grad else bias
    # (a training pass — the weight's gradient is being recorded — packs the planes afresh: PackedConvWeights; decided here because
    # grad mode is off inside an autograd Function's forward and ctx.needs_input_grad ignores torch.no_grad())
    fresh = torch.is_grad_enabled() and weight.requires_grad
    return _ConvReLUFn.apply(x, weight, b, packed, mode, grad_premasked, mask_input_grad, link_in, link_out, x_rec, y_rec, span, borders, fresh,
                             x_packed, y_packed, grad_packed)


def conv3x3_c1_bias_relu(x, weight, bias, pack=False, x_rec=None, y_rec=None):
    """Inference-only relu(conv2d(x[:,None], weight, bias, padding 1)) for x [B,H,W] fp32; returns [B,Cout,H,W]
    channels_last (NHWC memory).  One input channel: a pure HBM write stream (csrc/frontend_misc.hip).
    y_rec: range record of the result (optional unless pack).
    pack=True: the result is a PACKED activation tensor for conv3x3_x8(..., x_packed=True) (see there); Cout % 64 == 0;
    x_rec = the range record of x (measured here when None)."""
    lib = _lib.load()
    B, H, W = x.shape
    cout = weight.shape[0]
    x = x.contiguous()
    y = torch.empty((B, cout, H, W), dtype=torch.float16 if pack == 2 else torch.float32, device=x.device, memory_format=torch.channels_last)
    wp, bp = weight.detach().contiguous().data_ptr(), None if bias is None else bias.detach().data_ptr()
    with torch.cuda.device(x.device):
        stream = torch.cuda.current_stream(x.device).cuda_stream
        if pack == 2:      # a half-precision tensor for the fp16 activation chain (conv3x3_f16)
            if y_rec is None:
                raise RuntimeError("a half-precision output needs a range record (conv.new_ranges)")
            x_rec = range_of(x) if x_rec is None else x_rec
            rc = lib.dasv_conv3x3_c1_bias_relu_f16_nhwc(x.data_ptr(), x_rec.data_ptr(), wp, bp, y.data_ptr(), y_rec.data_ptr(), B, H, W,
                                                        cout, 1, stream)
        elif pack:
            if y_rec is None:
                raise RuntimeError("a packed output needs a range record (conv.new_ranges)")
            x_rec = range_of(x) if x_rec is None else x_rec
            rc = lib.dasv_conv3x3_c1_bias_relu_x8_nhwc(x.data_ptr(), x_rec.data_ptr(), wp, bp, y.data_ptr(), y_rec.data_ptr(), B, H, W,
                                                       cout, 1, stream)
        else:
            rc = lib.dasv_conv3x3_c1_bias_relu_nhwc(x.data_ptr(), wp, bp, y.data_ptr(), None if y_rec is None else y_rec.data_ptr(),
                                                    B, H, W, cout, 1, stream)
    _lib.check(rc, "dasv_conv3x3_c1_bias_relu_nhwc")
    return y


def pool_fusable(mode, B, H, W, cin, cout):
    """True when conv3x3 + bias + ReLU + 2x2 ceil max-pool runs as ONE launch for this layer (conv3x3_x8 / conv3x3_f16 with
    pool=True): maps up to 80 columns with enough tiles that the launch does not want split-K."""
    return exact(cin, cout) and bool(_lib.load().dasv_conv3x3_pool_supported(MODES[mode], B, H, W, cin, cout))


def conv3x3_x8(x, weight, bias, packed=None, x_packed=False, y_packed=False, x_rec=None, y_rec=None, pool=False, pool_tile=0):
    """Inference-only relu(conv2d(x, weight, bias, padding 1)) in the split8 mode on PACKED activations.

    A packed tensor looks like the fp32 channels_last tensor it replaces (same shape, dtype, strides; C % 64 == 0) but every
    256-byte slab of 64 channels holds the three operand planes the split8 kernel would make of the fp32 values while staging
    its window: fp16(c) | e4m3(c) | e4m3((c - hi) * 2^12) with c = v * 2^E, E in the tensor's range record (csrc/dasv_common.hpp,
    store_x8).  Made once by the producer — this function with y_packed=True, conv3x3_c1_bias_relu(pack=True),
    maxpool2x2_ceil(pack=True) — it turns the consumer's staging into a copy.  Only these three functions understand packed
    tensors: never hand one to a torch op.  x_rec: the record of x (required for a packed x; measured when None otherwise);
    y_rec: the record of y (required for a packed y).
    pool=True (see pool_fusable): F.max_pool2d(relu(conv(x)), 2, stride=2, ceil_mode=True) in the same launch — y is the pooled
    map [B, Cout, ceil(H/2), ceil(W/2)] and the full-resolution map never reaches memory (CNNs.py:73-74 ... 85-86)."""
    lib = _lib.load()
    B, cin, H, W = x.shape
    cout = weight.shape[0]
    if not x.is_cuda or x.dtype != torch.float32 or not exact(cin, cout) or W > 80:
        raise RuntimeError("conv3x3_x8: unsupported shape/dtype %s %s" % (tuple(x.shape), x.dtype))
    if not x.is_contiguous(memory_format=torch.channels_last):
        if x_packed:
            raise RuntimeError("a packed activation tensor must keep its channels_last layout")
        x = x.contiguous(memory_format=torch.channels_last)
    packed = packed or PackedConvWeights()
    hi, lo = packed.get(weight, "split8", bias)
    if x_rec is None:
        if x_packed:
            raise RuntimeError("a packed input needs the range record its producer filled")
        x_rec = range_of(x)
    if y_packed and y_rec is None:
        raise RuntimeError("a packed output needs a range record (conv.new_ranges)")
    if pool:
        y = torch.empty((B, cout, (H + 1) // 2, (W + 1) // 2), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        with torch.cuda.device(x.device):
            rc = lib.dasv_conv3x3_bias_relu_pool_x8_nhwc(x.data_ptr(), int(x_packed), x_rec.data_ptr(), hi.data_ptr(), lo.data_ptr(),
                                                         None if bias is None else bias.detach().data_ptr(), y.data_ptr(), int(y_packed),
                                                         None if y_rec is None else y_rec.data_ptr(), B, H, W, cin, cout, int(pool_tile),
                                                         torch.cuda.current_stream(x.device).cuda_stream)
        _lib.check(rc, "dasv_conv3x3_bias_relu_pool_x8_nhwc")
        return y
    y = torch.empty((B, cout, H, W), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    with torch.cuda.device(x.device):
        nws = 0 if y_packed else lib.dasv_conv3x3_workspace_bytes(B, H, W, cin, cout)
        ws = torch.empty(nws, dtype=torch.uint8, device=x.device) if nws else None
        rc = lib.dasv_conv3x3_bias_relu_x8_nhwc(x.data_ptr(), int(x_packed), x_rec.data_ptr(), hi.data_ptr(), lo.data_ptr(),
                                                None if bias is None else bias.detach().data_ptr(), y.data_ptr(), int(y_packed),
                                                None if y_rec is None else y_rec.data_ptr(),
                                                None if ws is None else ws.data_ptr(), nws, B, H, W, cin, cout, 1,
                                                torch.cuda.current_stream(x.device).cuda_stream)
    _lib.check(rc, "dasv_conv3x3_bias_relu_x8_nhwc")
    return y


def conv3x3_f16(x, weight, bias, packed=None, x_f16=False, y_f16=False, x_rec=None, y_rec=None, pool=False, pool_tile=0):
    """Inference-only relu(conv2d(x, weight, bias, padding 1)) with fp16 operands on HALF-PRECISION activation tensors.

    A half-precision tensor is what the fp32 channels_last tensor would be with 2-byte elements: [B,H,W,C] fp16 in NHWC memory
    (here a torch.float16 tensor of logical shape [B,C,H,W], channels_last), holding v * 2^E with E in the tensor's range record
    (conv.new_ranges) — half the HBM traffic of every layer, and the convolution's window staging becomes a copy.  Producers:
    this function (y_f16), conv3x3_c1_bias_relu(pack=2), maxpool2x2_ceil on a half-precision input."""
    lib = _lib.load()
    B, cin, H, W = x.shape
    cout = weight.shape[0]
    want = torch.float16 if x_f16 else torch.float32
    if not x.is_cuda or x.dtype != want or not exact(cin, cout) or W > 80 or not x.is_contiguous(memory_format=torch.channels_last):
        raise RuntimeError("conv3x3_f16: unsupported shape/dtype/layout %s %s" % (tuple(x.shape), x.dtype))
    packed = packed or PackedConvWeights()
    hi, _ = packed.get(weight, "fp16", bias)
    if x_rec is None:
        if x_f16:
            raise RuntimeError("a half-precision input needs the range record its producer filled")
        x_rec = range_of(x)
    if y_f16 and y_rec is None:
        raise RuntimeError("a half-precision output needs a range record (conv.new_ranges)")
    if pool:      # the 2x2 ceil max-pool in the same launch (conv3x3_x8's docstring): y is the pooled map
        y = torch.empty((B, cout, (H + 1) // 2, (W + 1) // 2), dtype=torch.float16 if y_f16 else torch.float32, device=x.device,
                        memory_format=torch.channels_last)
        with torch.cuda.device(x.device):
            rc = lib.dasv_conv3x3_bias_relu_pool_f16_nhwc(x.data_ptr(), int(x_f16), x_rec.data_ptr(), hi.data_ptr(),
                                                          None if bias is None else bias.detach().data_ptr(), y.data_ptr(), int(y_f16),
                                                          None if y_rec is None else y_rec.data_ptr(), B, H, W, cin, cout, int(pool_tile),
                                                          torch.cuda.current_stream(x.device).cuda_stream)
        _lib.check(rc, "dasv_conv3x3_bias_relu_pool_f16_nhwc")
        return y
    y = torch.empty((B, cout, H, W), dtype=torch.float16 if y_f16 else torch.float32, device=x.device, memory_format=torch.channels_last)
    with torch.cuda.device(x.device):
        nws = 0 if y_f16 else lib.dasv_conv3x3_workspace_bytes(B, H, W, cin, cout)
        ws = torch.empty(nws, dtype=torch.uint8, device=x.device) if nws else None
        rc = lib.dasv_conv3x3_bias_relu_f16_nhwc(x.data_ptr(), int(x_f16), x_rec.data_ptr(), hi.data_ptr(),
                                                 None if bias is None else bias.detach().data_ptr(), y.data_ptr(), int(y_f16),
                                                 None if y_rec is None else y_rec.data_ptr(), None if ws is None else ws.data_ptr(), nws,
                                                 B, H, W, cin, cout, 1, torch.cuda.current_stream(x.device).cuda_stream)
    _lib.check(rc, "dasv_conv3x3_bias_relu_f16_nhwc")
    return y


def conv3x3_bf16(x, weight, bias, packed=None, x_bf16=False, y_bf16=False):
    """Inference-only relu(conv2d(x, weight, bias, padding 1)) with bf16 operands on bf16 activation tensors (the bf16 STORAGE chain:
    torch.bfloat16 tensors of logical shape [B,C,H,W], channels_last — plain round-to-nearest-even values, no scale, no record).
    Producers: this function (y_bf16), conv3x3_c1_bias_relu_bf16, maxpool2x2_ceil_bf16."""
    lib = _lib.load()
    B, cin, H, W = x.shape
    cout = weight.shape[0]
    want = torch.bfloat16 if x_bf16 else torch.float32
    if not x.is_cuda or x.dtype != want or not exact(cin, cout) or W > 80 or not x.is_contiguous(memory_format=torch.channels_last):
        raise RuntimeError("conv3x3_bf16: unsupported shape/dtype/layout %s %s" % (tuple(x.shape), x.dtype))
    packed = packed or PackedConvWeights()
    hi, _ = packed.get(weight, "bf16")
    y = torch.empty((B, cout, H, W), dtype=torch.bfloat16 if y_bf16 else torch.float32, device=x.device, memory_format=torch.channels_last)
    with torch.cuda.device(x.device):
        nws = 0 if y_bf16 else lib.dasv_conv3x3_workspace_bytes(B, H, W, cin, cout)
        ws = torch.empty(nws, dtype=torch.uint8, device=x.device) if nws else None
        rc = lib.dasv_conv3x3_bias_relu_bf16_nhwc(x.data_ptr(), int(x_bf16), hi.data_ptr(), None if bias is None else bias.detach().data_ptr(),
                                                  y.data_ptr(), int(y_bf16), None if ws is None else ws.data_ptr(), nws, B, H, W, cin, cout, 1,
                                                  torch.cuda.current_stream(x.device).cuda_stream)
    _lib.check(rc, "dasv_conv3x3_bias_relu_bf16_nhwc")
    return y


def conv3x3_c1_bias_relu_bf16(x, weight, bias):
    """relu(conv2d(x[:,None], weight, bias, padding 1)) for x [B,H,W] fp32 -> a bf16 tensor [B,Cout,H,W] channels_last."""
    lib = _lib.load()
    B, H, W = x.shape
    cout = weight.shape[0]
    x = x.contiguous()
    y = torch.empty((B, cout, H, W), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    with torch.cuda.device(x.device):
        rc = lib.dasv_conv3x3_c1_bias_relu_bf16_nhwc(x.data_ptr(), weight.detach().contiguous().data_ptr(),
                                                     None if bias is None else bias.detach().data_ptr(), y.data_ptr(), B, H, W, cout, 1,
                                                     torch.cuda.current_stream(x.device).cuda_stream)
    _lib.check(rc, "dasv_conv3x3_c1_bias_relu_bf16_nhwc")
    return y


def maxpool2x2_ceil_bf16(x, y_bf16=True):
    """F.max_pool2d(x, 2, stride=2, ceil_mode=True) of a bf16 channels_last tensor; y bf16 (exact: a maximum of bf16 values) or fp32."""
    lib = _lib.load()
    B, C, H, W = x.shape
    if x.dtype != torch.bfloat16 or not x.is_contiguous(memory_format=torch.channels_last):
        raise RuntimeError("maxpool2x2_ceil_bf16: a bf16 channels_last tensor")
    y = torch.empty((B, C, (H + 1) // 2, (W + 1) // 2), dtype=torch.bfloat16 if y_bf16 else torch.float32, device=x.device,
                    memory_format=torch.channels_last)
    with torch.cuda.device(x.device):
        rc = lib.dasv_maxpool2x2_ceil_bf16_nhwc(x.data_ptr(), y.data_ptr(), int(y_bf16), B, H, W, C, torch.cuda.current_stream(x.device).cuda_stream)
    _lib.check(rc, "dasv_maxpool2x2_ceil_bf16_nhwc")
    return y


def maxpool2x2_ceil(x, pack=False, x_rec=None, y_rec=None):
    """Inference-only F.max_pool2d(x, 2, stride=2, ceil_mode=True) for a channels_last fp32 tensor [B,C,H,W].
    x_rec / y_rec (together): the pooled map's range record is its input's.
    pack=True: the pooled map is written as a PACKED activation tensor (conv3x3_x8), scaled by x_rec; C % 64 == 0."""
    lib = _lib.load()
    B, C, H, W = x.shape
    if not x.is_contiguous(memory_format=torch.channels_last):
        x = x.contiguous(memory_format=torch.channels_last)
    if x.dtype == torch.float16:
        # a half-precision tensor of the fp16 activation chain (conv3x3_f16): pack=2 keeps it one (same exponent), pack=False
        # gives the fp32 map (unscaled)
        if x_rec is None or pack not in (False, 0, 2) or (pack == 2 and y_rec is None):
            raise RuntimeError("a half-precision input needs its range record (and a record for a half-precision output)")
        y = torch.empty((B, C, (H + 1) // 2, (W + 1) // 2), dtype=torch.float16 if pack == 2 else torch.float32, device=x.device,
                        memory_format=torch.channels_last)
        with torch.cuda.device(x.device):
            rc = lib.dasv_maxpool2x2_ceil_f16_nhwc(x.data_ptr(), x_rec.data_ptr(), y.data_ptr(), int(pack == 2),
                                                   None if y_rec is None else y_rec.data_ptr(), B, H, W, C,
                                                   torch.cuda.current_stream(x.device).cuda_stream)
        _lib.check(rc, "dasv_maxpool2x2_ceil_f16_nhwc")
        return y
    if pack and y_rec is None:
        raise RuntimeError("a packed output needs a range record (conv.new_ranges)")
    if pack and x_rec is None:
        x_rec = range_of(x)
    if x_rec is None or y_rec is None:
        x_rec = y_rec = None
    y = torch.empty((B, C, (H + 1) // 2, (W + 1) // 2), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    fn = lib.dasv_maxpool2x2_ceil_x8_nhwc if pack else lib.dasv_maxpool2x2_ceil_nhwc
    with torch.cuda.device(x.device):
        rc = fn(x.data_ptr(), None if x_rec is None else x_rec.data_ptr(), y.data_ptr(), None if y_rec is None else y_rec.data_ptr(),
                B, H, W, C, torch.cuda.current_stream(x.device).cuda_stream)
    _lib.check(rc, "dasv_maxpool2x2_ceil_nhwc")
    return y


def conv3x3_c1_conv3x3_bias_relu(mel, w1, b1, weight2, bias2, packed=None, mode="split"):
    """Inference-only first VGG block in one launch: relu(conv2(relu(conv1(mel[:,None])))) for mel [B,H,W] fp32, conv1 with
    one input channel (its activations are produced while conv2's input window is staged and never reach HBM)."""
    lib = _lib.load()
    B, H, W = mel.shape
    cmid, cout = weight2.shape[1], weight2.shape[0]
    packed = packed or PackedConvWeights()
    hi, lo = packed.get(weight2, mode)
    y = torch.empty((B, cout, H, W), dtype=torch.float32, device=mel.device, memory_format=torch.channels_last)
    with torch.cuda.device(mel.device):
        rc = lib.dasv_conv3x3_c1_conv3x3_bias_relu_nhwc(
            MODES[mode], mel.contiguous().data_ptr(), w1.detach().contiguous().data_ptr(),
            None if b1 is None else b1.detach().data_ptr(), hi.data_ptr(), lo.data_ptr(),
            None if bias2 is None else bias2.detach().data_ptr(), y.data_ptr(), B, H, W, cmid, cout,
            torch.cuda.current_stream(mel.device).cuda_stream)
    _lib.check(rc, "dasv_conv3x3_c1_conv3x3_bias_relu_nhwc")
    return y


class _MaxPoolFn(torch.autograd.Function):
    """2x2 stride-2 ceil-mode max-pool on channels_last fp32 with a hand-written backward.  The forward launch also writes one
    byte per pooled element {arg-max position | 4 if the maximum is > 0} (dasv_maxpool2x2_ceil_idx_nhwc); the backward reads those
    bytes and the incoming gradient only — the full-resolution map is not saved and not read again."""

    @staticmethod
    def forward(ctx, x, relu_input, link_out, x_rec, y_rec, pool_src=None, pack_out=False, pack_grad=False):
        lib = _lib.load()
        if (pack_out or pack_grad) and (pool_src is None or x_rec is None or y_rec is None or not relu_input):
            raise RuntimeError("packed pool output / gradient: the ragged packed training chain (records, ReLU input)")
        if not x.is_contiguous(memory_format=torch.channels_last):
            x = x.contiguous(memory_format=torch.channels_last)
        B, C, H, W = x.shape
        # pool_src (ragged batch: one stacked image): [rows_out, 2] = {first input row, has-second-row} per output row
        Ho, Wo = ((H + 1) // 2 if pool_src is None else int(pool_src.shape[0])), (W + 1) // 2
        y = torch.empty((B, C, Ho, Wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        idx = torch.empty((B, Ho, Wo, C), dtype=torch.uint8, device=x.device)
        xr = None if x_rec is None else x_rec.data_ptr()
        yr = None if (x_rec is None or y_rec is None) else y_rec.data_ptr()
        with torch.cuda.device(x.device):
            st = torch.cuda.current_stream(x.device).cuda_stream
            if pool_src is None:
                rc = lib.dasv_maxpool2x2_ceil_idx_nhwc(x.data_ptr(), xr, y.data_ptr(), yr, idx.data_ptr(), B, H, W, C, st)
            else:
                if B != 1:
                    raise RuntimeError("a ragged batch is one stacked image")
                fn = lib.dasv_maxpool2x2_ceil_idx_ragged_x8_nhwc if pack_out else lib.dasv_maxpool2x2_ceil_idx_ragged_nhwc
                rc = fn(x.data_ptr(), xr, y.data_ptr(), yr, idx.data_ptr(), pool_src.data_ptr(), H, Ho, W, C, st)
        _lib.check(rc, "dasv_maxpool2x2_ceil_idx_nhwc")
        ctx.save_for_backward(idx)
        ctx.pack_grad = pack_grad
        ctx.pool_src = pool_src
        ctx.xshape = (B, C, H, W)
        ctx.relu_input = relu_input
        ctx.link_out = link_out
        return y

    @staticmethod
    def backward(ctx, gy):
        (idx,) = ctx.saved_tensors
        lib = _lib.load()
        B, C, H, W = ctx.xshape
        gy = gy.contiguous(memory_format=torch.channels_last)
        gx = torch.empty((B, C, H, W), dtype=torch.float32, device=gy.device, memory_format=torch.channels_last)
        with torch.cuda.device(gy.device):
            stream = torch.cuda.current_stream(gy.device).cuda_stream
            want_sum = ctx.relu_input and ctx.link_out is not None
            if ctx.pool_src is None:
                rows = lib.dasv_maxpool2x2_bwd_colsum_rows(B, H, W, C) if want_sum else 0
            else:
                rows = lib.dasv_maxpool2x2_bwd_colsum_rows_ragged(int(ctx.pool_src.shape[0]), W, C) if want_sum else 0
            part = torch.empty((rows, C), dtype=torch.float32, device=gy.device) if rows > 0 else None
            rec = _rec_of(gy)
            if ctx.pack_grad:         # the scattered gradient PACKED, scaled by the exponent of gy's maximum (its own maximum)
                if rec is None:
                    rec = range_of(gy)
                grec = new_ranges(gy.device)[0]
                rc = lib.dasv_maxpool2x2_ceil_bwd_idx_ragged_x8_nhwc(idx.data_ptr(), gy.data_ptr(), rec.data_ptr(), gx.data_ptr(), grec.data_ptr(),
                                                                     None if part is None else part.data_ptr(), int(bool(ctx.relu_input)),
                                                                     ctx.pool_src.data_ptr(), H, int(ctx.pool_src.shape[0]), W, C, stream)
                rec = grec
            elif ctx.pool_src is None:
                rc = lib.dasv_maxpool2x2_ceil_bwd_idx_nhwc(idx.data_ptr(), gy.data_ptr(), gx.data_ptr(), None if part is None else part.data_ptr(),
                                                           int(bool(ctx.relu_input)), B, H, W, C, stream)
            else:
                rc = lib.dasv_maxpool2x2_ceil_bwd_idx_ragged_nhwc(idx.data_ptr(), gy.data_ptr(), gx.data_ptr(),
                                                                  None if part is None else part.data_ptr(), int(bool(ctx.relu_input)),
                                                                  ctx.pool_src.data_ptr(), H, int(ctx.pool_src.shape[0]), W, C, stream)
            if part is not None and rc == 0:
                ctx.link_out["gb"] = _sum_rows(part)
        _lib.check(rc, "dasv_maxpool2x2_ceil_bwd_idx_nhwc")
        if rec is not None:      # the scattered gradient's range record is its source's (a bound: max |gx| <= max |gy|)
            _attach_rec(gx, rec)
        return gx, None, None, None, None, None, None, None


def maxpool2x2_ceil_autograd(x, relu_input=False, link_out=None, x_rec=None, y_rec=None, pool_src=None, pack_out=False, pack_grad=False):
    """Differentiable F.max_pool2d(x, 2, stride=2, ceil_mode=True) for channels_last fp32 GPU tensors.

    relu_input=True: x is the output of a conv3x3_bias_relu(..., grad_premasked=True) call and nothing else consumes it —
    the backward then also passes the gradient through that ReLU (one fused pass instead of two), and leaves that conv's
    bias gradient in `link_out["gb"]` (see conv3x3_bias_relu).
    pack_out / pack_grad (ragged packed training chain, pool_src given): the pooled map / the gradient sent back to x are PACKED
    tensors (conv3x3_x8) — x itself stays fp32."""
    return _MaxPoolFn.apply(x, relu_input, link_out, x_rec, y_rec, pool_src, pack_out, pack_grad)


class _Conv11ReLUFn(torch.autograd.Function):
    """relu(conv2d(x[:,None], w, b, padding 1)) for x [B,H,W] (no gradient w.r.t. x: x is the input feature map)."""

    @staticmethod
    def forward(ctx, x, weight, bias, y_rec, grad_premasked=False, span=None, pack=False, x_rec=None):
        if pack and not grad_premasked:
            raise RuntimeError("a packed first-layer output: the ReLU mask is applied by the consumer")
        if span is None:
            y = conv3x3_c1_bias_relu(x, weight, bias, pack=pack, x_rec=x_rec, y_rec=y_rec)
        else:      # ragged batch: x [1, rows, W], utterances stacked along the rows, borders from `span`
            lib = _lib.load()
            x = x.contiguous()
            B, H, W = x.shape
            if B != 1:
                raise RuntimeError("a ragged batch is one stacked image")
            y = torch.empty((1, weight.shape[0], H, W), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
            with torch.cuda.device(x.device):
                if pack:
                    if x_rec is None:
                        x_rec = range_of(x)
                    rc = lib.dasv_conv3x3_c1_bias_relu_ragged_x8_nhwc(x.data_ptr(), x_rec.data_ptr(), weight.detach().contiguous().data_ptr(),
                                                                      None if bias is None else bias.detach().data_ptr(), y.data_ptr(),
                                                                      y_rec.data_ptr(), span.data_ptr(), H, W, weight.shape[0], 1,
                                                                      torch.cuda.current_stream(x.device).cuda_stream)
                else:
                    rc = lib.dasv_conv3x3_c1_bias_relu_ragged_nhwc(x.data_ptr(), weight.detach().contiguous().data_ptr(),
                                                                   None if bias is None else bias.detach().data_ptr(), y.data_ptr(),
                                                                   None if y_rec is None else y_rec.data_ptr(), span.data_ptr(), H, W,
                                                                   weight.shape[0], 1, torch.cuda.current_stream(x.device).cuda_stream)
            _lib.check(rc, "dasv_conv3x3_c1_bias_relu_ragged_nhwc")
        ctx.span = span
        ctx.save_for_backward(x.contiguous(), y)
        ctx.grad_premasked = grad_premasked
        ctx.has_bias = bias is not None
        ctx.wshape = weight.shape
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y = ctx.saved_tensors
        lib = _lib.load()
        B, H, W = x.shape
        cout = y.shape[1]
        gy = gy.contiguous(memory_format=torch.channels_last)
        gw = torch.empty(ctx.wshape, dtype=torch.float32, device=x.device)
        gb = torch.empty(cout, dtype=torch.float32, device=x.device) if ctx.has_bias else None
        nbytes = lib.dasv_conv3x3_c1_bwd_workspace_bytes(B, H, W, cout)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        with torch.cuda.device(x.device):
            # (grad_premasked: the consumer's data-gradient kernel multiplied gy by (y > 0) already — y is not read again)
            if ctx.span is None:
                rc = lib.dasv_conv3x3_c1_bwd_nhwc(x.data_ptr(), None if ctx.grad_premasked else y.data_ptr(), gy.data_ptr(), gw.data_ptr(),
                                                  None if gb is None else gb.data_ptr(), ws.data_ptr(), nbytes, B, H, W, cout,
                                                  torch.cuda.current_stream(x.device).cuda_stream)
            else:
                rc = lib.dasv_conv3x3_c1_bwd_ragged_nhwc(x.data_ptr(), None if ctx.grad_premasked else y.data_ptr(), gy.data_ptr(),
                                                         gw.data_ptr(), None if gb is None else gb.data_ptr(), ws.data_ptr(), nbytes,
                                                         ctx.span.data_ptr(), H, W, cout, torch.cuda.current_stream(x.device).cuda_stream)
        _lib.check(rc, "dasv_conv3x3_c1_bwd_nhwc")
        return None, gw, gb, None, None, None, None, None


def conv3x3_c1_bias_relu_autograd(x, weight, bias, y_rec=None, grad_premasked=False, span=None, pack=False, x_rec=None):
    """Differentiable (w.r.t. weight and bias) first layer on the HIP kernels; x [B,H,W] must not require grad.
    grad_premasked: the one consumer of the result applies this layer's ReLU mask to the gradient it sends back (a following
    conv3x3_bias_relu(..., mask_input_grad=True)), so the backward reads that gradient only."""
    return _Conv11ReLUFn.apply(x, weight, bias, y_rec, grad_premasked, span, pack, x_rec)
