"""VGG frame front-ends with the reference's interface (scripts/CNNs.py).

    getVGG3LOutputDimension / getVGG4LOutputDimension      CNNs.py:7-20
    VGG3L(kernel_size), VGG4L(kernel_size)                 CNNs.py:22-91

Each block is [conv3x3 pad1 + ReLU] x2 followed by a 2x2 stride-2 ceil-mode max-pool on a
[B,1,T,F] map; the output is flattened to [B,T',C*F'] with feature index c*F'+f
(CNNs.py:49-50 / 88-89).  Parameter names (conv11 ... conv42) match the reference so its
checkpoints load with strict=True.

Convolutions: every layer but conv11 runs on the hand-written MFMA implicit-GEMM `conv3x3 + bias + ReLU` of csrc/conv3x3.hip on
fp32 NHWC activations.  The kernel works on 64-channel slabs and 128-channel tiles; at a `kernel_size` whose channel counts are
not such multiples (the reference takes any, CNNs.py:27-32 / 59-66) the front-end runs as the zero-padded network of the next
such width — padded weight planes, activations padded between the layers (the pad channels stay zero through bias, ReLU and
max-pool), the last map sliced back: no second convolution backend on the path.  `conv_mode`:
  "split8" (default) fp32-grade forward: fp16 products + block-scaled e4m3 cross terms (~1e-5 of the output scale per layer) at
           any magnitude of activations and weights — every tensor is scaled into the planes' range by an exact power of two
           taken on the device from its range record (conv.new_ranges), Inf / NaN propagate; the backward pass of these layers
           runs in "split"
  "split"  fp32-grade accuracy on the bf16 matrix cores (operands split hi + lo, 3 MFMAs per product, ~5e-6)
  "bf16"   operands rounded to bf16 (one MFMA per product)
  "fp16"   operands rounded to fp16 (one MFMA per product, 11 mantissa bits; gradients use bf16 operands); in inference the
           activations between the layers are then STORED as fp16 too (`half_activations`: 2 bytes per element, scaled per tensor
           by its range record so that fp16's range is never the limit — BASELINE config 3 "bf16" / config 5 "fp16 MFMA path")
  "torch"  PyTorch-ROCm / MIOpen for every layer (comparison runs only)
Under `torch.autocast` the HIP kernels run with bf16 / fp16 operands (the autocast dtype), fp32 accumulation and fp32
activations.
conv11 (Cin = 1) and the 2x2 ceil-mode max-pools run on hand-written HIP streaming kernels too (csrc/frontend_misc.hip),
forward and backward (arg-max recomputed; one-channel weight/bias gradient as a fixed-order reduction).
"""
import math

import torch
from torch import nn
from torch.nn import functional as F

from . import conv as _hipconv

__all__ = ["getVGG3LOutputDimension", "getVGG4LOutputDimension", "VGG3L", "VGG4L"]


def _pooled(n, times):
    for _ in range(times):
        n = int(math.ceil(n / 2))
    return n


def getVGG3LOutputDimension(inputDimension, outputChannel=128):
    return _pooled(int(inputDimension), 3) * outputChannel


def getVGG4LOutputDimension(inputDimension, outputChannel=128):
    return _pooled(int(inputDimension), 4) * outputChannel


class _VGG(nn.Module):
    """n_blocks VGG blocks; block i (1-based) has kernel_size / 2**(n_blocks-i) channels."""

    n_blocks = 0

    def __init__(self, kernel_size):
        super().__init__()
        cin = 1
        for blk in range(1, self.n_blocks + 1):
            cout = int(kernel_size / (2 ** (self.n_blocks - blk)))
            setattr(self, "conv%d1" % blk, nn.Conv2d(cin, cout, 3, stride=1, padding=1))
            setattr(self, "conv%d2" % blk, nn.Conv2d(cout, cout, 3, stride=1, padding=1))
            cin = cout
        self.channels_last = True
        self.conv_mode = "split8"
        # conv11 fused into conv12's window staging (one launch, conv11's 786 MB output never reaches HBM): correct and
        # tested, but measured SLOWER end to end (6.74 vs 6.63 ms per [64,300,80] batch): conv12's loop is issue-bound and
        # the nine extra FMAs per staged value cost more than the saved traffic.  Off by default.
        self.fuse_first_block = False
        # inference: activations between the split8 layers in the consumer's operand form (conv.conv3x3_x8); same results
        self.packed_activations = True
        # inference in the fp16- / bf16-operand modes: activations between the layers as fp16 (scaled by range records, conv.conv3x3_f16)
        # or bf16 (plain values, conv.conv3x3_bf16) tensors
        self.half_activations = True
        # inference chains: convX2 + ReLU + the block's 2x2 ceil max-pool as ONE launch (conv.conv3x3_x8 / conv3x3_f16, pool=True).
        # `fuse_pool_layers`: the (block, mode) pairs it is switched on for — where the same-process A/B on [64,300,80] says it pays
        # (tools/_diag/pool_fuse_ab.py, profiles/r04_pool_fusion_ab.txt, r04_patch_tiles_ab.txt).  split8: block 1 runs on 16 x 16
        # pixel PATCH tiles (csrc/conv3x3.hip GEOM = 1), which hold whole pooling windows and waste nothing: -5.6 % of the
        # front-end's time; block 2 (row-pair tiles, 240 of 256 pixel rows used) another -0.7 %; blocks 3 / 4 lose 0.4 / 2.4 % — their
        # row-pair tiles push 760 / 1,500 workgroups into one more round of the 256 CUs, which costs more than their small pools
        # save.  With fp16 operands the convolutions are twice as fast and only block 2 breaks even.  Every pair runs and is tested.
        self.fuse_pool = True
        self.pool_tile = 0          # tile shape of the fused launches (0: automatic; measurements force 160 / 192 / 256)
        self.fuse_pool_layers = {(1, "split8"), (2, "split8"), (2, "fp16")}
        # training (split8): the same one-launch form of convX2 + ReLU + pool (conv.conv3x3_bias_relu_pool) for these blocks — the
        # launch also writes the pool's arg-max bytes, which is all the backward needs of the full-resolution map
        self.fuse_pool_training = {1, 2, 3, 4}
        # training (split8, dense batches, exact channel counts): EVERY activation of the forward pass in the packed operand form (the
        # inference chain plus the pools' arg-max bytes; only the last pooled map is fp32) — the next layer's staging, the weight
        # gradient's producers and the data gradient's mask read all take the packed form as it is (_forward_map_packed_train)
        self.packed_training = True
        # ... with the first block as one autograd node whose backward folds conv11's weight / bias gradient into conv12's data-gradient
        # launch (conv.first_block_train): the gradient at conv11's output is never written
        self.fuse_first_backward = True
        self._packed = {}

    def _mode(self):
        """Operand precision of the HIP convolutions for this call: `conv_mode`, except under torch.autocast, where the
        autocast dtype picks the single-MFMA mode (bf16 / fp16 operands, fp32 accumulation and fp32 activations — at least
        the precision autocast asks for); None = not expressible (other autocast dtypes): PyTorch-ROCm ops run."""
        if not torch.is_autocast_enabled():
            return self.conv_mode
        return {torch.bfloat16: "bf16", torch.float16: "fp16"}.get(torch.get_autocast_dtype("cuda"))

    def _use_hip_conv(self, name, h):
        conv = getattr(self, name)
        return (self.conv_mode != "torch" and h.is_cuda and h.dtype == torch.float32
                and self._mode() is not None and conv.weight.dtype == torch.float32
                and _hipconv.supported(conv.in_channels, conv.out_channels))

    def _conv_relu(self, name, h, grad_premasked=False, mask_input_grad=False, link_in=None, link_out=None, x_rec=None, y_rec=None,
                   span=None, borders=None):
        """x_rec / y_rec: range records of h and of the result (split8 mode; None: h is measured by a pass over it).  Returns
        (result, record of the result or None).  span / borders: the border tables of a ragged training batch (forward_map)."""
        conv = getattr(self, name)
        if not self._use_hip_conv(name, h):
            if span is not None:
                raise RuntimeError("ragged training batches run on the hand-written kernels only")
            return F.relu(conv(h)), None
        key = (name, h.device.index)   # nn.DataParallel replicas share this dict: one packed copy per device
        packed = self._packed.get(key)
        if packed is None:
            packed = self._packed[key] = _hipconv.PackedConvWeights()
        return _hipconv.conv3x3_bias_relu(h, conv.weight, conv.bias, packed, self._mode(), grad_premasked, mask_input_grad,
                                          link_in, link_out, x_rec, y_rec, span, borders), y_rec

    def _c1_padded(self):
        """conv11's weight / bias zero-padded to the channel count the hand-written kernels want from the first layer: a multiple
        of 64 that the one-channel kernel takes (a power of two between 64 and 1024).  (weight, bias) themselves when no padding
        is needed; F.pad is differentiable, so training sees the gradients of the real rows."""
        c = self.conv11.out_channels
        cp = 64
        while cp < c:
            cp *= 2
        if cp == c or cp > 1024:
            return self.conv11.weight, self.conv11.bias
        return F.pad(self.conv11.weight, (0, 0, 0, 0, 0, 0, 0, cp - c)), F.pad(self.conv11.bias, (0, cp - c))

    def _packed_chain_ok(self, fdim, batch=None, frames=None):
        """Inference in the split8 mode with every wide layer on the HIP kernel and channel counts the packed format takes:
        activations between the layers are then written once in the consumer's operand form (conv.conv3x3_x8).  Not for small
        batches: a packed output is written by the convolution's own epilogue, so those launches could not divide their channel
        slabs over workgroups (split-K) — the fp32 chain is faster there (single utterances: train.py:117-133)."""
        mode = self._mode()
        if (mode != "split8" and not (mode in ("fp16", "bf16") and self.half_activations)) or self.fuse_first_block or fdim > 80 \
                or self.conv11.out_channels % 64 != 0:
            return False
        if batch is not None:
            from . import _lib
            h, w = _pooled(frames, self.n_blocks - 1), _pooled(fdim, self.n_blocks - 1)     # the last block's (smallest) maps
            last = getattr(self, "conv%d1" % self.n_blocks)
            if _lib.load().dasv_conv3x3_workspace_bytes(batch, h, w, last.in_channels, last.out_channels) != 0:
                return False
        for blk in range(1, self.n_blocks + 1):
            for j in (1, 2):
                c = getattr(self, "conv%d%d" % (blk, j))
                if (blk, j) != (1, 1) and not (_hipconv.exact(c.in_channels, c.out_channels) and c.weight.dtype == torch.float32):
                    return False
        return True

    def _pk(self, name, device):
        key = (name, device.index)
        packed = self._packed.get(key)
        if packed is None:
            packed = self._packed[key] = _hipconv.PackedConvWeights()
        return packed

    def _forward_map_packed(self, x):
        """conv11 -> PACKED -> conv12 (+ pool) -> PACKED -> convX1 -> PACKED -> convX2 (+ pool) -> PACKED ... : the fp32 chain's values to
        2e-5 (same conversion arithmetic, done by the producer instead of by every consumer workgroup; a packed OUTPUT is scaled by
        the layer's bound where the consumer of an fp32 tensor uses its measured maximum).  A packed
        input saves a layer 5-7 %.  The pools write packed for free (0.185 vs 0.183 ms), conv11 for +0.02 ms; a convolution's
        epilogue only because it goes through LDS (coalescing epilogue): with 8- and 4-byte stores straight to memory the packed
        output cost the producing layer 18 % (tools/_diag/packed_pieces.py)."""
        dev = x.device
        x = x.contiguous()      # (a time crop `feats[:, s:s+T]` is a view: the record below must be taken over what conv11 reads)
        # one range record per tensor of the chain (conv.new_ranges): the input features' is measured (6 MB), every kernel fills
        # its output's; a packed tensor is scaled by the exponent its producer left in its record
        recs = _hipconv.new_ranges(dev, 1 + 3 * self.n_blocks)
        r = iter(recs)
        rx = next(r)
        _lib = _hipconv._lib
        with torch.cuda.device(dev):
            _lib.check(_lib.load().dasv_range_absmax_f32(x.data_ptr(), x.numel(), rx.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
                       "dasv_range_absmax_f32")
        rh = next(r)
        h = _hipconv.conv3x3_c1_bias_relu(x, self.conv11.weight, self.conv11.bias, pack=True, x_rec=rx, y_rec=rh)
        for blk in range(1, self.n_blocks + 1):
            c1, c2 = getattr(self, "conv%d1" % blk), getattr(self, "conv%d2" % blk)
            if blk > 1:
                ry = next(r)
                h = _hipconv.conv3x3_x8(h, c1.weight, c1.bias, self._pk("conv%d1" % blk, dev), x_packed=True, y_packed=True, x_rec=rh, y_rec=ry)
                rh = ry
            ry = next(r)
            Bn, _, Hn, Wn = h.shape
            if self.fuse_pool and (blk, "split8") in self.fuse_pool_layers and \
                    _hipconv.pool_fusable("split8", Bn, Hn, Wn, c2.in_channels, c2.out_channels):
                # conv + ReLU + max-pool in ONE launch: convX2's full-resolution map never reaches memory
                h = _hipconv.conv3x3_x8(h, c2.weight, c2.bias, self._pk("conv%d2" % blk, dev), x_packed=True,
                                        y_packed=blk < self.n_blocks, x_rec=rh, y_rec=ry, pool=True, pool_tile=self.pool_tile)
                rh = ry
                continue
            h = _hipconv.conv3x3_x8(h, c2.weight, c2.bias, self._pk("conv%d2" % blk, dev), x_packed=True, y_packed=False, x_rec=rh, y_rec=ry)
            rh, ry = ry, next(r)
            h = _hipconv.maxpool2x2_ceil(h, pack=blk < self.n_blocks, x_rec=rh, y_rec=ry)
            rh = ry
        return h

    def _packed_training_ok(self, fdim, batch, frames):
        """The packed training chain exists when the packed inference chain does (_packed_chain_ok), every convolution's input
        channels are a multiple of 128 (its data gradient then runs un-padded) and every block's convX2 + pool has its one-launch
        form."""
        if not self._packed_chain_ok(fdim, batch, frames):
            return False
        H, W = frames, fdim
        for blk in range(1, self.n_blocks + 1):
            c1, c2 = getattr(self, "conv%d1" % blk), getattr(self, "conv%d2" % blk)
            for c in (c1, c2):
                if c is self.conv11:
                    continue
                if c.in_channels % 128 != 0:
                    return False
            if not _hipconv.pool_fusable("split8", batch, H, W, c2.in_channels, c2.out_channels):
                return False
            H, W = (H + 1) // 2, (W + 1) // 2
        cp = self.conv11.out_channels
        return cp % 128 == 0 and (cp & (cp - 1)) == 0 and cp <= 1024       # conv11 writes the packed form for these widths

    def _forward_map_packed_train(self, x):
        """The training forward on packed activations: conv11 -> PACKED -> conv12 + pool (+ arg-max bytes) -> PACKED -> convX1 ->
        PACKED -> convX2 + pool -> ... -> fp32 map; every node's backward reads the packed tensors directly (conv.conv3x3_bias_relu,
        conv.conv3x3_bias_relu_pool with x_packed / y_packed).  Same values as _forward_map_packed."""
        dev = x.device
        x = x.contiguous()
        r = iter(_hipconv.new_ranges(dev, 1 + 2 * self.n_blocks))
        rx = next(r)
        _lib = _hipconv._lib
        with torch.cuda.device(dev):
            _lib.check(_lib.load().dasv_range_absmax_f32(x.data_ptr(), x.numel(), rx.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
                       "dasv_range_absmax_f32")
        rh = next(r)
        first = 1
        if self.fuse_first_backward and self.conv11.bias is not None:
            ry = next(r)
            h = _hipconv.first_block_train(x, self.conv11.weight, self.conv11.bias, self.conv12.weight, self.conv12.bias,
                                           self._pk("conv12", dev), rx, rh, ry, y_packed=self.n_blocks > 1)
            rh, first = ry, 2
        else:
            h = _hipconv.conv3x3_c1_bias_relu_autograd(x, self.conv11.weight, self.conv11.bias, rh, grad_premasked=True, pack=True, x_rec=rx)
        for blk in range(first, self.n_blocks + 1):
            n1, n2 = "conv%d1" % blk, "conv%d2" % blk
            c1, c2 = getattr(self, n1), getattr(self, n2)
            l1 = {}
            if blk > 1:
                ry = next(r)
                h = _hipconv.conv3x3_bias_relu(h, c1.weight, c1.bias, self._pk(n1, dev), "split8", grad_premasked=True, link_in=l1,
                                               x_rec=rh, y_rec=ry, x_packed=True, y_packed=True)
                rh = ry
            ry = next(r)
            h = _hipconv.conv3x3_bias_relu_pool(h, c2.weight, c2.bias, self._pk(n2, dev), mask_input_grad=True,
                                                link_out=None if blk == 1 else l1,       # conv11's kernel sums its own bias gradient
                                                x_rec=rh, y_rec=ry, x_packed=True, y_packed=blk < self.n_blocks)
            rh = ry
        return h

    def _packed_ragged_training_ok(self, fdim):
        """The packed training chain over a RAGGED batch (utterances stacked along time, border tables): exact channel counts, every
        convolution's input a multiple of 128 channels, conv11 writing the packed form."""
        if not (self.packed_training and self.ragged_training_ok(fdim)):
            return False
        cp = self.conv11.out_channels
        if not (cp % 128 == 0 and (cp & (cp - 1)) == 0 and cp <= 1024):
            return False
        return all(getattr(self, "conv%d%d" % (blk, j)).in_channels % 128 == 0
                   for blk in range(1, self.n_blocks + 1) for j in (1, 2) if (blk, j) != (1, 1))

    def _forward_map_packed_ragged_train(self, x, tables):
        """forward_map(x, tables) on packed activations: conv11 -> PACKED -> conv12 -> fp32 -> pool (+ arg-max bytes) -> PACKED -> convX1 ->
        PACKED -> convX2 -> fp32 -> pool -> ... -> fp32 map.  Border tables in every launch; the pools stay their own launches (a tile of
        whole row pairs does not exist across utterance borders), so convX2's output is the one fp32 activation per block.  Backward: the
        pool writes the scattered gradient packed, convX2's data gradient reads it and the packed mask, every weight gradient copies
        its packed x (convX2's: g too) — on the stacked image minus the cross-border products, as in the fp32 wiring."""
        spans, pools, borders = tables
        dev = x.device
        x = x.contiguous()
        r = iter(_hipconv.new_ranges(dev, 2 + 3 * self.n_blocks))
        rx = next(r)
        _lib = _hipconv._lib
        with torch.cuda.device(dev):
            _lib.check(_lib.load().dasv_range_absmax_f32(x.data_ptr(), x.numel(), rx.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
                       "dasv_range_absmax_f32")
        rh = next(r)
        fused_first = self.fuse_first_backward and self.conv11.bias is not None
        if not fused_first:
            h = _hipconv.conv3x3_c1_bias_relu_autograd(x, self.conv11.weight, self.conv11.bias, rh, grad_premasked=True, span=spans[0], pack=True,
                                                       x_rec=rx)
        for blk in range(1, self.n_blocks + 1):
            n1, n2 = "conv%d1" % blk, "conv%d2" % blk
            c1, c2 = getattr(self, n1), getattr(self, n2)
            sp, bd = spans[blk - 1], borders[blk - 1]
            l1, l2 = {}, {}
            if blk == 1 and fused_first:
                # conv11 + conv12 as one node: conv11's backward rides in conv12's data-gradient launch (conv.first_block_ragged_train)
                ry = next(r)
                h = _hipconv.first_block_ragged_train(x, self.conv11.weight, self.conv11.bias, c2.weight, c2.bias, self._pk(n2, dev),
                                                      rx, rh, ry, sp, bd, l2)
                rh, ry = ry, next(r)
                h = _hipconv.maxpool2x2_ceil_autograd(h, relu_input=True, link_out=l2, x_rec=rh, y_rec=ry, pool_src=pools[0],
                                                      pack_out=self.n_blocks > 1, pack_grad=True)
                rh = ry
                continue
            if blk > 1:
                ry = next(r)
                h = _hipconv.conv3x3_bias_relu(h, c1.weight, c1.bias, self._pk(n1, dev), "split8", grad_premasked=True, link_in=l1,
                                               x_rec=rh, y_rec=ry, span=sp, borders=bd, x_packed=True, y_packed=True)
                rh = ry
            ry = next(r)
            h = _hipconv.conv3x3_bias_relu(h, c2.weight, c2.bias, self._pk(n2, dev), "split8", grad_premasked=True, mask_input_grad=True,
                                           link_in=l2, link_out=None if blk == 1 else l1, x_rec=rh, y_rec=ry, span=sp, borders=bd,
                                           x_packed=True, y_packed=False, grad_packed=True)
            rh, ry = ry, next(r)
            h = _hipconv.maxpool2x2_ceil_autograd(h, relu_input=True, link_out=l2, x_rec=rh, y_rec=ry, pool_src=pools[blk - 1],
                                                  pack_out=blk < self.n_blocks, pack_grad=True)
            rh = ry
        return h

    def _forward_map_f16(self, x):
        """The same chain with HALF-PRECISION tensors between the layers (fp16-operand mode, inference): conv11 -> f16 -> conv12 ->
        f16 -> pool -> f16 -> ... -> convN2 -> f16 -> last pool -> fp32 map.  Every tensor holds v * 2^E, E from its range record
        (conv.conv3x3_f16); values are rounded to fp16 once per layer — the rounding the fp16 operands apply anyway."""
        dev = x.device
        x = x.contiguous()      # (see _forward_map_packed)
        recs = _hipconv.new_ranges(dev, 1 + 3 * self.n_blocks)
        r = iter(recs)
        rx = next(r)
        _lib = _hipconv._lib
        with torch.cuda.device(dev):
            _lib.check(_lib.load().dasv_range_absmax_f32(x.data_ptr(), x.numel(), rx.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
                       "dasv_range_absmax_f32")
        rh = next(r)
        h = _hipconv.conv3x3_c1_bias_relu(x, self.conv11.weight, self.conv11.bias, pack=2, x_rec=rx, y_rec=rh)
        for blk in range(1, self.n_blocks + 1):
            c1, c2 = getattr(self, "conv%d1" % blk), getattr(self, "conv%d2" % blk)
            if blk > 1:
                ry = next(r)
                h = _hipconv.conv3x3_f16(h, c1.weight, c1.bias, self._pk("conv%d1" % blk, dev), x_f16=True, y_f16=True, x_rec=rh, y_rec=ry)
                rh = ry
            ry = next(r)
            Bn, _, Hn, Wn = h.shape
            if self.fuse_pool and (blk, "fp16") in self.fuse_pool_layers and \
                    _hipconv.pool_fusable("fp16", Bn, Hn, Wn, c2.in_channels, c2.out_channels):
                h = _hipconv.conv3x3_f16(h, c2.weight, c2.bias, self._pk("conv%d2" % blk, dev), x_f16=True, y_f16=blk < self.n_blocks,
                                         x_rec=rh, y_rec=ry, pool=True, pool_tile=self.pool_tile)
                rh = ry
                continue
            h = _hipconv.conv3x3_f16(h, c2.weight, c2.bias, self._pk("conv%d2" % blk, dev), x_f16=True, y_f16=True, x_rec=rh, y_rec=ry)
            rh, ry = ry, next(r)
            h = _hipconv.maxpool2x2_ceil(h, pack=2 if blk < self.n_blocks else False, x_rec=rh, y_rec=ry)
            rh = ry
        return h

    def _forward_map_bf16(self, x):
        """The bf16 STORAGE chain of the bf16-operand mode (inference; BASELINE config 3 names bf16): conv11 -> bf16 -> conv12 -> bf16 ->
        pool -> bf16 -> ... -> convN2 -> bf16 -> last pool -> fp32 map.  Plain bf16 values between the layers (no scale, no records): each
        layer's output is rounded ONCE, to the 8 bits its consumer's operands would keep anyway — the values the fp32-activation bf16
        mode multiplies, at half the activation traffic."""
        dev = x.device
        h = _hipconv.conv3x3_c1_bias_relu_bf16(x.contiguous(), self.conv11.weight, self.conv11.bias)
        for blk in range(1, self.n_blocks + 1):
            c1, c2 = getattr(self, "conv%d1" % blk), getattr(self, "conv%d2" % blk)
            if blk > 1:
                h = _hipconv.conv3x3_bf16(h, c1.weight, c1.bias, self._pk("conv%d1" % blk, dev), x_bf16=True, y_bf16=True)
            h = _hipconv.conv3x3_bf16(h, c2.weight, c2.bias, self._pk("conv%d2" % blk, dev), x_bf16=True, y_bf16=True)
            h = _hipconv.maxpool2x2_ceil_bf16(h, y_bf16=blk < self.n_blocks)
        return h

    def _hip_front(self, x):
        """conv11 and the max-pools on the HIP kernels: fp32 activations on the GPU, not under autocast, and the input
        features themselves do not need a gradient."""
        c11 = self.conv11
        return (self.conv_mode != "torch" and x.is_cuda and x.dtype == torch.float32 and not x.requires_grad
                and self._mode() is not None and c11.weight.dtype == torch.float32 and c11.out_channels <= 1024)

    def forward(self, paddedInputTensor):
        h = self.forward_map(paddedInputTensor)
        # [B,C,T',F'] -> [B,T',C*F'] with feature = c*F' + f
        return h.permute(0, 2, 1, 3).reshape(h.size(0), h.size(2), h.size(1) * h.size(3))

    def ragged_training_ok(self, fdim=80):
        """A ragged TRAINING batch (forward_map(..., tables=...)) runs when every layer is on the hand-written kernels in the split8
        mode with channel counts the kernels take without padding."""
        if self.conv_mode != "split8" or torch.is_autocast_enabled() or fdim > 80 or self.conv11.out_channels % 64 != 0 \
                or self.conv11.out_channels > 1024:
            return False
        return all(_hipconv.exact(c.in_channels, c.out_channels) and c.weight.dtype == torch.float32
                   for blk in range(1, self.n_blocks + 1) for c in (getattr(self, "conv%d1" % blk), getattr(self, "conv%d2" % blk))
                   if c is not self.conv11)

    def forward_map(self, paddedInputTensor, tables=None):
        """The front-end up to, not including, the final transpose/contiguous/view (CNNs.py:49-50 / 88-89): the
        [B, C, T', F'] map (channels_last memory on the HIP path).  The pooling kernels can read it in place
        (poolings.nhwc_frames_supported), which is how SpeakerClassifier uses it.
        tables (training on a RAGGED batch, ragged.train_tables): paddedInputTensor is [1, sum T_i, F] — utterances of different
        lengths stacked along time — and tables = (spans, pools, borders) per block; forward and backward then treat every
        utterance as its own image (the reference can only do that one utterance per forward)."""
        B, T, Fdim = paddedInputTensor.shape
        hip = self._hip_front(paddedInputTensor)
        grad = torch.is_grad_enabled()
        spans = pools = borders = [None] * self.n_blocks
        if tables is not None:
            if not (hip and grad and B == 1 and self.ragged_training_ok(Fdim)):
                raise RuntimeError("ragged training batches need the split8 mode on the GPU, gradients enabled and exact channel counts")
            spans, pools, borders = tables
            packed_ragged = self._packed_ragged_training_ok(Fdim) and not paddedInputTensor.requires_grad
        if hip and grad and any(q.requires_grad for q in self.parameters()):
            # a pass that an optimizer step may follow: torch's FUSED optimizers update the parameters without moving their version
            # counters (the packed planes' cache key), so no packed copy made before this pass may be reused after it — neither by
            # the training passes (they pack on every call, conv.PackedConvWeights) nor by the no-grad chains' own caches
            _hipconv.invalidate_packed_weights(self)
        if hip and not grad and self.packed_activations and self._mode() == "split8" and self._packed_chain_ok(Fdim, B, T):
            return self._forward_map_packed(paddedInputTensor)
        if (hip and grad and tables is None and self.packed_training and self._mode() == "split8" and not paddedInputTensor.requires_grad
                and self._packed_training_ok(Fdim, B, T)):
            return self._forward_map_packed_train(paddedInputTensor)
        if tables is not None and packed_ragged:
            return self._forward_map_packed_ragged_train(paddedInputTensor, tables)
        if hip and not grad and self.half_activations and self._mode() == "fp16" and not torch.is_autocast_enabled() \
                and self._packed_chain_ok(Fdim, B, T):
            return self._forward_map_f16(paddedInputTensor)
        if hip and not grad and self.half_activations and self._mode() == "bf16" and not torch.is_autocast_enabled() \
                and self._packed_chain_ok(Fdim, B, T):
            return self._forward_map_bf16(paddedInputTensor)
        fused_first = False
        # split8: one range record per tensor (see _forward_map_packed); rh = the record of h, None where h has none
        x8 = hip and self._mode() == "split8"
        recs = iter(_hipconv.new_ranges(paddedInputTensor.device, 3 * self.n_blocks)) if x8 else None
        rh = None
        if hip:
            c12 = self.conv12
            fused_first = (self.fuse_first_block and not grad and Fdim <= 80 and c12.weight.dtype == torch.float32
                           and _hipconv.exact(c12.in_channels, c12.out_channels))
            if fused_first:   # conv11 + conv12 in one launch: conv11's activations never reach HBM
                key = ("conv12", paddedInputTensor.device.index)
                packed = self._packed.get(key)
                if packed is None:
                    packed = self._packed[key] = _hipconv.PackedConvWeights()
                h = _hipconv.conv3x3_c1_conv3x3_bias_relu(paddedInputTensor, self.conv11.weight, self.conv11.bias,
                                                          c12.weight, c12.bias, packed,
                                                          "split" if self._mode() == "split8" else self._mode())   # no fused form of split8
            elif grad:
                rh = next(recs) if x8 else None
                # conv12 runs with mask_input_grad (below) whenever it is on the HIP kernel: conv11's ReLU mask is then applied there
                h = _hipconv.conv3x3_c1_bias_relu_autograd(paddedInputTensor, *self._c1_padded(), rh,
                                                           grad_premasked=self._use_hip_conv("conv12", paddedInputTensor), span=spans[0])
            else:
                rh = next(recs) if x8 else None
                w11, b11 = self._c1_padded()
                h = _hipconv.conv3x3_c1_bias_relu(paddedInputTensor, w11, b11, y_rec=rh)
        else:
            h = paddedInputTensor.reshape(B, 1, T, Fdim)
            if self.channels_last and h.is_cuda:
                h = h.contiguous(memory_format=torch.channels_last)
        for blk in range(1, self.n_blocks + 1):
            n1, n2 = "conv%d1" % blk, "conv%d2" % blk
            # Backward fusion between neighbours that both run on the HIP kernels (conv.conv3x3_bias_relu docstring): the
            # ReLU mask of convX1 is applied by convX2's data-gradient kernel, that of convX2 by the pool's backward.
            # The kernel that applies a mask also leaves the masked gradient's per-channel sums (the bias gradient of the
            # layer it masks for) in a dict both calls share.
            hip2 = self._use_hip_conv(n2, h)
            pool_hip = hip and h.dtype == torch.float32
            l1, l2 = ({}, {}) if grad else (None, None)
            sp, bd = spans[blk - 1], borders[blk - 1]
            if not (hip and blk == 1):
                h, rh = self._conv_relu(n1, h, grad_premasked=hip2, link_in=l1, x_rec=rh, y_rec=next(recs) if x8 else None, span=sp, borders=bd)
            c2 = getattr(self, n2)
            if (grad and x8 and hip2 and pool_hip and sp is None and blk in self.fuse_pool_training and h.dtype == torch.float32
                    and h.shape[1] == c2.in_channels
                    and _hipconv.pool_fusable("split8", h.shape[0], h.shape[2], h.shape[3], c2.in_channels, c2.out_channels)):
                next(recs)          # (the record of the full-resolution map, which does not exist in this form)
                ry = next(recs)
                h = _hipconv.conv3x3_bias_relu_pool(h, c2.weight, c2.bias, self._pk(n2, h.device), mask_input_grad=True,
                                                    link_out=None if (hip and blk == 1) else l1, x_rec=rh, y_rec=ry)
                rh = ry
                continue
            if not (fused_first and blk == 1):
                h, rh = self._conv_relu(n2, h, grad_premasked=hip2 and pool_hip, mask_input_grad=True, link_in=l2,
                                        link_out=None if (hip and blk == 1) else l1,   # conv11's kernel sums its own bias gradient
                                        x_rec=rh, y_rec=next(recs) if x8 else None, span=sp, borders=bd)
            if pool_hip and h.dtype == torch.float32:
                ry = next(recs) if (x8 and rh is not None) else None
                h = (_hipconv.maxpool2x2_ceil_autograd(h, relu_input=hip2, link_out=l2, x_rec=rh, y_rec=ry, pool_src=pools[blk - 1]) if grad
                     else _hipconv.maxpool2x2_ceil(h, x_rec=rh, y_rec=ry))
                rh = ry
            else:
                h = F.max_pool2d(h, 2, stride=2, ceil_mode=True)
                rh = None
        last = getattr(self, "conv%d2" % self.n_blocks).out_channels
        if h.shape[1] != last:      # the zero-padded network of a narrow `kernel_size`: drop the pad channels of the last map
            h = h[:, :last]
        return h


class VGG3L(_VGG):
    n_blocks = 3


class VGG4L(_VGG):
    n_blocks = 4
