"""Ragged batches: utterances of DIFFERENT lengths through the front-end in ONE set of launches — inference (embed_ragged) and, through
the dense autograd functions of conv.py with the border tables threaded through, training (utterance_vectors_ragged_train).

The reference can only run unequal utterances one per forward (train.py:117-133): there are no padding masks anywhere in
the model, so zero-padding a batch changes the result.  But the convolutions and the pools are position-independent
except at the utterance borders.  Here the utterances are stacked along the time axis into one tall NHWC image
[sum T_i, 80, C]; the kernels take the borders from small integer tables (`dasv_*_ragged_nhwc`, include/dasv_hip.h):
a conv tap that would cross into the neighbouring utterance reads zero, exactly as at the top/bottom of a single image,
and the ceil-mode pool of an odd-length utterance clips its last window.  Every output pixel sees the same operands in
the same order as in a one-utterance forward, so the embeddings equal the B=1 results (up to the split-K summation
order of small maps) while the launches are as large as a full batch — no per-length buckets.

`embed_ragged(net, utterances)` -> [len(utterances), emb] for a SpeakerClassifier with a VGG front-end; the pooling runs
on a zero-padded [B, T'max, D] copy of the (small) front-end output with per-utterance `lengths`.
"""
import torch

from . import _lib
from . import conv as _conv
from . import poolings as _pool

__all__ = ["embed_ragged", "embed_ragged_batches", "ragged_tables", "front_end_ragged", "plan_ragged_groups", "train_tables",
           "utterance_vectors_ragged_train"]


def _upload_int64(values, device):
    """A short Python list as an int64 device tensor WITHOUT blocking the host: a pageable host-to-device copy waits for the
    stream's queued kernels first, which serialised every group of utterances behind the previous one (136 of the 168 ms of a
    1,024-utterance step were spent inside those tiny copies).  Pinned staging (torch's caching host allocator keeps the block
    alive until the copy has run) + non_blocking keeps the host ahead of the GPU."""
    t = torch.tensor(values, dtype=torch.int64)
    if torch.device(device).type != "cuda":
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


def _upload_np(arr, device):
    """A small numpy array as a device tensor without blocking the host (pinned staging, see _upload_int64)."""
    t = torch.from_numpy(arr)
    if torch.device(device).type != "cuda":
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


def ragged_tables(lengths, levels, device, with_index=False):
    """Border tables for `levels` conv blocks.  Returns (spans, pools, lens): spans[l] int32 [rows_l, 2] (first row, one
    past the last row of the owning utterance) for the maps entering block l; pools[l] int32 [rows_{l+1}, 2] (first input
    row, has-second-row) of the pool that closes block l; lens[l] the utterance lengths at level l as a Python list
    (lens[levels] = output lengths).  with_index: a fourth value, int64 [2, rows_levels] = (owner utterance, row inside it) of
    every output frame.

    On the GPU the tables come from ONE launch (dasv_ragged_tables) fed by one tiny non-blocking upload of the lengths.  Round 2
    built them with torch ops — ~60 tiny launches per group (cumsum / repeat_interleave / arange / gather / stack per level),
    11 % of the GPU time of a variable-length step; building them on the host and uploading ~400 KB per group instead made the
    HOST the bottleneck (132 -> 240-330 ms per step: copies of that size do not run ahead of the stream the way tiny ones do)."""
    import numpy as np
    cur = np.asarray([int(n) for n in lengths], dtype=np.int64)
    if cur.size == 0 or cur.min() <= 0:
        raise ValueError("lengths must be a non-empty list of positive frame counts")
    device = torch.device(device)
    lens, rows = [], []
    c = cur
    for _ in range(levels + 1):
        lens.append([int(n) for n in c])
        rows.append(int(c.sum()))
        c = (c + 1) // 2
    if device.type == "cuda":
        lib = _lib.load()
        host = np.ascontiguousarray(cur, dtype=np.int32)
        dev_lens = _upload_np(host, device)
        total = sum(2 * (rows[l] + rows[l + 1]) for l in range(levels))
        flat = torch.empty(total, dtype=torch.int32, device=device)
        fin = torch.empty((2, rows[levels]), dtype=torch.int64, device=device) if with_index else None
        with torch.cuda.device(device):
            rc = lib.dasv_ragged_tables(dev_lens.data_ptr(), host.ctypes.data, int(cur.size), levels, flat.data_ptr(),
                                        None if fin is None else fin.data_ptr(), torch.cuda.current_stream(device).cuda_stream)
        _lib.check(rc, "dasv_ragged_tables")
        spans, pools, off = [], [], 0
        for l in range(levels):
            spans.append(flat[off:off + 2 * rows[l]].view(rows[l], 2))
            off += 2 * rows[l]
            pools.append(flat[off:off + 2 * rows[l + 1]].view(rows[l + 1], 2))
            off += 2 * rows[l + 1]
        return (spans, pools, lens, fin) if with_index else (spans, pools, lens)
    # host arithmetic (CPU tensors: the tests' reference walk, and the specification of the kernel above)
    spans, pools = [], []
    c = cur
    for _ in range(levels):
        start = np.cumsum(c) - c
        span = np.stack([np.repeat(start, c), np.repeat(start + c, c)], axis=1)
        nxt = (c + 1) // 2
        ostart = np.cumsum(nxt) - nxt
        local = np.arange(int(nxt.sum()), dtype=np.int64) - np.repeat(ostart, nxt)          # output row inside its utterance
        first = np.repeat(start, nxt) + 2 * local
        second = (2 * local + 1 < np.repeat(c, nxt)).astype(np.int64)
        spans.append(torch.from_numpy(span.astype(np.int32)).to(device))
        pools.append(torch.from_numpy(np.stack([first, second], axis=1).astype(np.int32)).to(device))
        c = nxt
    if with_index:
        fin = torch.from_numpy(np.stack([np.repeat(np.arange(c.size, dtype=np.int64), c),
                                         np.arange(int(c.sum()), dtype=np.int64) - np.repeat(np.cumsum(c) - c, c)])).to(device)
        return spans, pools, lens, fin
    return spans, pools, lens


def _conv_ragged(x, conv, packed, mode, span, x_packed=False, y_packed=False, x_rec=None, y_rec=None):
    """x [rows, W, Cin] NHWC fp32 -> relu(conv3x3(x) + bias) [rows, W, Cout] with utterance borders from `span`.
    x_packed / y_packed (split8 only): the tensors are PACKED activations (conv.conv3x3_x8).
    x_rec / y_rec (split8): range records of x and y (conv.new_ranges)."""
    lib = _lib.load()
    rows, W, cin = x.shape
    # narrow kernel sizes run as their zero-padded form (conv.supported): x may carry more channels than the layer has inputs
    # (zeros), y gets padded_channels(Cout) channels, the surplus identically zero
    cout = _conv.padded_channels(conv.out_channels)
    pad = (cin, cout) if (cin, cout) != (conv.in_channels, conv.out_channels) else None
    hi, lo = packed.get(conv.weight, mode, conv.bias, pad)
    bias = packed.bias if (pad is not None and conv.bias is not None) else conv.bias
    bias_ptr = None if bias is None else bias.detach().data_ptr()
    y = torch.empty((rows, W, cout), dtype=torch.float32, device=x.device)
    nws = 0 if y_packed else lib.dasv_conv3x3_workspace_bytes(1, rows, W, cin, cout)
    ws = torch.empty(nws, dtype=torch.uint8, device=x.device) if nws else None
    if (x_packed or y_packed) and mode != "split8":
        raise RuntimeError("packed activations exist in the split8 mode only")
    if mode == "split8":
        if x_rec is None:
            if x_packed:
                raise RuntimeError("a packed input needs the range record its producer filled")
            x_rec = _conv.range_of(x)
        with torch.cuda.device(x.device):
            rc = lib.dasv_conv3x3_bias_relu_ragged_x8_nhwc(
                x.data_ptr(), int(x_packed), x_rec.data_ptr(), hi.data_ptr(), lo.data_ptr(),
                bias_ptr, y.data_ptr(), int(y_packed),
                None if y_rec is None else y_rec.data_ptr(), span.data_ptr(), None if ws is None else ws.data_ptr(), nws, rows, W, cin,
                cout, 1, torch.cuda.current_stream(x.device).cuda_stream)
        _lib.check(rc, "dasv_conv3x3_bias_relu_ragged_x8_nhwc")
        return y
    with torch.cuda.device(x.device):
        rc = lib.dasv_conv3x3_bias_relu_ragged_nhwc(
            _conv.MODES[mode], x.data_ptr(), hi.data_ptr(), lo.data_ptr(), bias_ptr,
            y.data_ptr(), span.data_ptr(), None if ws is None else ws.data_ptr(), nws, rows, W, cin, cout, 1,
            torch.cuda.current_stream(x.device).cuda_stream)
    _lib.check(rc, "dasv_conv3x3_bias_relu_ragged_nhwc")
    return y


def _packed_chain_ok(front, rows, W, pools):
    """True when no convolution of the ragged pass over `rows` stacked frames would divide its channel slabs over workgroups."""
    lib = _lib.load()
    for blk in range(1, front.n_blocks + 1):
        for name in ("conv%d1" % blk, "conv%d2" % blk):
            conv = getattr(front, name)
            if conv.in_channels == 1:
                if conv.out_channels % 64 != 0:
                    return False
                continue
            if not _conv.exact(conv.in_channels, conv.out_channels) or W > 80:
                return False
            if lib.dasv_conv3x3_workspace_bytes(1, rows, W, conv.in_channels, conv.out_channels) != 0:
                return False
        rows, W = pools[blk - 1].shape[0], (W + 1) // 2
    return True


def _conv_ragged_f16(x, conv, w_f16, span, x_rec, y_rec):
    """One ragged convolution of the half-precision chain: [rows, W, Cin] fp16 -> [rows, W, Cout] fp16 (scales in the records)."""
    lib = _lib.load()
    rows, W, cin = x.shape
    y = torch.empty((rows, W, conv.out_channels), dtype=torch.float16, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.dasv_conv3x3_bias_relu_ragged_f16_nhwc(
            x.data_ptr(), 1, x_rec.data_ptr(), w_f16.data_ptr(), None if conv.bias is None else conv.bias.detach().data_ptr(),
            y.data_ptr(), 1, y_rec.data_ptr(), span.data_ptr(), None, 0, rows, W, cin, conv.out_channels, 1,
            torch.cuda.current_stream(x.device).cuda_stream)
    _lib.check(rc, "dasv_conv3x3_bias_relu_ragged_f16_nhwc")
    return y


def _front_end_ragged_f16(front, x, spans, pools, lens, dev):
    """front_end_ragged in the fp16-operand mode with fp16 activation storage: conv11 -> f16 -> convX1 -> f16 -> convX2 -> f16 ->
    pool -> f16 ... -> last pool -> fp32.  Every half-precision tensor holds v * 2^E, E in its range record."""
    lib = _lib.load()
    stream = torch.cuda.current_stream(dev).cuda_stream
    W = x.shape[1]
    recs = iter(_conv.new_ranges(dev, 2 + 3 * front.n_blocks))
    rx, rh = next(recs), next(recs)
    c11 = front.conv11
    rows = x.shape[0]
    h = torch.empty((rows, W, c11.out_channels), dtype=torch.float16, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.dasv_range_absmax_f32(x.data_ptr(), x.numel(), rx.data_ptr(), stream), "dasv_range_absmax_f32")
        _lib.check(lib.dasv_conv3x3_c1_bias_relu_ragged_f16_nhwc(
            x.data_ptr(), rx.data_ptr(), c11.weight.detach().contiguous().data_ptr(),
            None if c11.bias is None else c11.bias.detach().data_ptr(), h.data_ptr(), rh.data_ptr(), spans[0].data_ptr(), rows, W,
            c11.out_channels, 1, stream), "dasv_conv3x3_c1_bias_relu_ragged_f16_nhwc")
    for blk in range(1, front.n_blocks + 1):
        span = spans[blk - 1]
        for name in ("conv%d1" % blk, "conv%d2" % blk):
            conv = getattr(front, name)
            if conv.in_channels == 1:
                continue
            key = (name, dev.index)
            packed = front._packed.get(key)
            if packed is None:
                packed = front._packed[key] = _conv.PackedConvWeights()
            hi, _ = packed.get(conv.weight, "fp16", conv.bias)
            ry = next(recs)
            h, rh = _conv_ragged_f16(h, conv, hi, span, rh, ry), ry
        rows_in, rows_out = h.shape[0], pools[blk - 1].shape[0]
        Wo = (W + 1) // 2
        last = blk == front.n_blocks
        y = torch.empty((rows_out, Wo, h.shape[2]), dtype=torch.float32 if last else torch.float16, device=dev)
        ry = next(recs)
        with torch.cuda.device(dev):
            rc = lib.dasv_maxpool2x2_ceil_ragged_f16_nhwc(h.data_ptr(), rh.data_ptr(), y.data_ptr(), 0 if last else 1, ry.data_ptr(),
                                                          pools[blk - 1].data_ptr(), rows_in, rows_out, W, h.shape[2], stream)
        _lib.check(rc, "dasv_maxpool2x2_ceil_ragged_f16_nhwc")
        h, W, rh = y, Wo, ry
    frames = h.permute(0, 2, 1).reshape(h.shape[0], h.shape[2] * h.shape[1])
    return frames, list(lens[-1])


@torch.no_grad()
def front_end_ragged(front, utterances, mode=None, with_index=False):
    """VGG front-end over stacked utterances.  Returns (frames [sum T'_i, D] with the reference's feature order c*F'+f,
    lengths T'_i as a list); with_index: also int64 [2, sum T'_i] = (owner utterance, row inside it) of every frame."""
    out = _front_end_ragged(front, utterances, mode)
    return out if with_index else out[:2]


def _front_end_ragged(front, utterances, mode):
    lib = _lib.load()
    dev = next(front.parameters()).device
    if dev.type != "cuda":
        raise RuntimeError("the ragged front-end runs on the GPU only (libdasv_hip.so); no CPU fallback")
    mode = mode or front.conv_mode
    if mode == "torch":
        raise RuntimeError("the ragged path needs the HIP convolutions (conv_mode split8 / split / bf16 / fp16)")
    x = torch.cat([u.to(dev, torch.float32) for u in utterances], dim=0).contiguous()        # [rows, 80]
    W = x.shape[1]
    spans, pools, lens, fin = ragged_tables([int(u.shape[0]) for u in utterances], front.n_blocks, dev, with_index=True)
    stream = torch.cuda.current_stream(dev).cuda_stream
    # split8: the activations between the layers in the consumer's operand form, as in CNNs._forward_map_packed — when no launch
    # of the chain wants split-K (groups of utterances that fill the chip; a packed output has no split-K form)
    chain = mode == "split8" and getattr(front, "packed_activations", False) and _packed_chain_ok(front, x.shape[0], W, pools)
    if mode == "fp16" and getattr(front, "half_activations", False) and _packed_chain_ok(front, x.shape[0], W, pools):
        # the fp16-operand mode with HALF-PRECISION activation tensors between the layers (CNNs._forward_map_f16; BASELINE
        # config 5 names exactly this: variable-length batches on the fp16 MFMA path)
        return _front_end_ragged_f16(front, x, spans, pools, lens, dev) + (fin,)
    h, h_packed = None, False
    # split8: one range record per tensor of the pass (conv.new_ranges; CNNs._forward_map_packed)
    x8 = mode == "split8"
    recs = iter(_conv.new_ranges(dev, 2 + 3 * front.n_blocks)) if x8 else None
    rh = None
    if x8:
        rx = next(recs)
        with torch.cuda.device(dev):
            _lib.check(lib.dasv_range_absmax_f32(x.data_ptr(), x.numel(), rx.data_ptr(), stream), "dasv_range_absmax_f32")
    for blk in range(1, front.n_blocks + 1):
        span = spans[blk - 1]
        for name in ("conv%d1" % blk, "conv%d2" % blk):
            conv = getattr(front, name)
            first = name.endswith("1")
            if conv.in_channels == 1:
                rows = x.shape[0]
                # (narrow kernel sizes: the first layer zero-padded to the 64-multiple the wide kernels read, CNNs._c1_padded)
                w11, b11 = front._c1_padded() if hasattr(front, "_c1_padded") else (conv.weight, conv.bias)
                w11 = w11.detach().contiguous()
                c11out = w11.shape[0]
                h = torch.empty((rows, W, c11out), dtype=torch.float32, device=dev)
                rh = next(recs) if x8 else None
                wp, bp = w11.data_ptr(), None if b11 is None else b11.detach().data_ptr()
                with torch.cuda.device(dev):
                    if chain:
                        rc = lib.dasv_conv3x3_c1_bias_relu_ragged_x8_nhwc(x.data_ptr(), rx.data_ptr(), wp, bp, h.data_ptr(), rh.data_ptr(),
                                                                          span.data_ptr(), rows, W, c11out, 1, stream)
                    else:
                        rc = lib.dasv_conv3x3_c1_bias_relu_ragged_nhwc(x.data_ptr(), wp, bp, h.data_ptr(), None if rh is None else rh.data_ptr(),
                                                                       span.data_ptr(), rows, W, c11out, 1, stream)
                _lib.check(rc, "dasv_conv3x3_c1_bias_relu_ragged_nhwc")
                h_packed = chain
            elif chain:
                key = (name, dev.index)
                packed = front._packed.get(key)
                if packed is None:
                    packed = front._packed[key] = _conv.PackedConvWeights()
                # convX1 hands convX2 a packed tensor; convX2's output goes to the pool in fp32 (the maximum needs all its bits)
                ry = next(recs)
                h = _conv_ragged(h, conv, packed, mode, span, x_packed=h_packed, y_packed=first, x_rec=rh, y_rec=ry)
                h_packed, rh = first, ry
            else:
                key = (name, dev.index)
                packed = front._packed.get(key)
                if packed is None:
                    packed = front._packed[key] = _conv.PackedConvWeights()
                ry = next(recs) if x8 else None
                h = _conv_ragged(h, conv, packed, mode, span, x_rec=rh, y_rec=ry)
                rh = ry
        rows_in, rows_out = h.shape[0], pools[blk - 1].shape[0]
        Wo = (W + 1) // 2
        y = torch.empty((rows_out, Wo, h.shape[2]), dtype=torch.float32, device=dev)
        pack_out = chain and blk < front.n_blocks          # the last pool feeds the pooling layer: fp32
        fn = lib.dasv_maxpool2x2_ceil_ragged_x8_nhwc if pack_out else lib.dasv_maxpool2x2_ceil_ragged_nhwc
        ry = next(recs) if (x8 and rh is not None) else None
        with torch.cuda.device(dev):
            rc = fn(h.data_ptr(), None if ry is None else rh.data_ptr(), y.data_ptr(), None if ry is None else ry.data_ptr(),
                    pools[blk - 1].data_ptr(), rows_in, rows_out, W, h.shape[2], stream)
        _lib.check(rc, "dasv_maxpool2x2_ceil_ragged_nhwc")
        h, W, h_packed, rh = y, Wo, pack_out, ry
    # [rows, F', C] -> [rows, C*F'] with feature = c*F' + f (CNNs.py:49-50 / 88-89); a zero-padded last layer gives its real channels
    c_last = getattr(front, "conv%d2" % front.n_blocks).out_channels
    if h.shape[2] != c_last:
        h = h[:, :, :c_last]
    frames = h.permute(0, 2, 1).reshape(h.shape[0], c_last * h.shape[1])
    return frames, list(lens[-1]), fin


@torch.no_grad()
def embed_ragged(net, utterances, mode=None):
    """Embeddings [len(utterances), emb] of utterances with arbitrary (different) frame counts, one set of launches."""
    if net.training:
        raise RuntimeError("embed_ragged is an inference path: call net.eval() first")
    import numpy as np
    frames, tl, fin = front_end_ragged(net.front_end, utterances, mode, with_index=True)
    B, tmax, D = len(tl), max(tl), frames.shape[1]
    padded = frames.new_zeros((B, tmax, D))
    padded[fin[0], fin[1]] = frames          # (owner, row) of every frame: made with the border tables (ragged_tables)
    lengths = _upload_np(np.asarray(tl, dtype=np.int32), frames.device)
    pool = net.poolingLayer
    if isinstance(pool, _pool.DoubleMHA):
        vec = pool.pool(padded, lengths=lengths)
    elif isinstance(pool, (_pool.MultiHeadAttention, _pool.Attention)):
        vec = pool(padded, lengths=lengths)[0]
    else:
        raise RuntimeError("unsupported pooling layer %r" % type(pool).__name__)
    hidden = torch.relu(net.fc1(vec))
    return net.b2(torch.relu(net.fc2(hidden)))


def train_tables(lengths, levels, device):
    """Border tables of a ragged TRAINING batch: (spans, pools, borders) per block for CNNs.forward_map(..., tables=...) —
    ragged_tables' spans / pools plus, per level, the first rows of utterances 1 ... n-1 (int64; what conv.conv3x3_wgrad subtracts
    the cross-border products at) — and the output lengths and the (owner, row) index of every output frame."""
    import numpy as np
    spans, pools, lens, fin = ragged_tables(lengths, levels, device, with_index=True)
    borders = [_upload_int64([int(v) for v in np.cumsum(lens[l])[:-1]], device) for l in range(levels)]
    return (spans, pools, borders), list(lens[-1]), fin


def utterance_vectors_ragged_train(net, utterances):
    """Pooled vectors [B, vector_size] of a TRAINING minibatch of utterances of different lengths in ONE set of launches, forward
    and backward (BASELINE.json north_star; the reference can only run unequal utterances one per forward, train.py:117-133): the
    utterances are stacked along time, the front-end's kernels take the borders from tables in both directions (the weight gradient
    subtracts the cross-border products, conv.conv3x3_wgrad), and the pooling runs on a zero-padded [B, T'max, D] copy of the (small)
    front-end output with per-utterance lengths.  Differentiable; same values as one utterance per forward."""
    import numpy as np
    front = net.front_end
    dev = next(net.parameters()).device
    x = torch.cat([u.to(dev, non_blocking=True).float() for u in utterances], dim=0).unsqueeze(0)       # [1, sum T_i, 80]
    tables, tl, fin = train_tables([int(u.shape[0]) for u in utterances], front.n_blocks, dev)
    h = front.forward_map(x, tables=tables)                                     # [1, C, rows', F'] (channels_last)
    frames = h[0].permute(1, 0, 2).reshape(h.shape[2], h.shape[1] * h.shape[3])   # feature = c * F' + f (CNNs.py:88-89)
    B, tmax = len(tl), max(tl)
    padded = frames.new_zeros((B, tmax, frames.shape[1])).index_put((fin[0], fin[1]), frames)
    lengths = _upload_np(np.asarray(tl, dtype=np.int32), dev)
    pool = net.poolingLayer
    if isinstance(pool, _pool.DoubleMHA):
        return pool.pool(padded, lengths=lengths)
    if isinstance(pool, (_pool.MultiHeadAttention, _pool.Attention)):
        return pool(padded, lengths=lengths)[0]
    raise RuntimeError("unsupported pooling layer %r" % type(pool).__name__)


def plan_ragged_groups(lengths, max_frames=64 * 300, rank=0, world=1):
    """Which utterances this rank embeds, and in which groups (pure host arithmetic, the same on every rank: no communication).

    Utterances are dealt to the `world` ranks first — longest first, each to the rank with the fewest frames so far (LPT), so the
    ranks' total frames differ by at most one utterance — and every rank then packs ITS utterances, longest first, into groups
    of up to `max_frames` frames (one ragged pass each; similar lengths share a group).  Dealing whole groups instead would
    leave ranks a full group apart: 26 groups over 8 ranks is 4 against 3 passes.  Returns a list of index lists."""
    lengths = [int(n) for n in lengths]
    order = sorted(range(len(lengths)), key=lambda i: (-lengths[i], i))
    if world > 1:
        load = [0] * world
        mine = []
        for i in order:
            r = min(range(world), key=lambda j: (load[j], j))
            load[r] += lengths[i]
            if r == rank:
                mine.append(i)
    else:
        mine = order
    groups, cur, frames = [], [], 0
    for i in mine:
        if cur and frames + lengths[i] > max_frames:
            groups.append(cur)
            cur, frames = [], 0
        cur.append(i)
        frames += lengths[i]
    if cur:
        groups.append(cur)
    return groups


@torch.no_grad()
def embed_ragged_batches(net, utterances, max_frames=64 * 300, mode=None, rank=0, world=1):
    """Embeddings of many utterances of arbitrary lengths: groups of up to `max_frames` total frames (the footprint of a
    [64,300,80] batch by default), each group one ragged pass.  Returns {index: embedding} for the utterances this rank owns
    (plan_ragged_groups: utterances dealt to the ranks by frames, then grouped per rank; all of them when world == 1)."""
    out = {}
    for g in plan_ragged_groups([u.shape[0] for u in utterances], max_frames, rank, world):
        emb = embed_ragged(net, [utterances[i] for i in g], mode)
        for j, i in enumerate(g):
            out[i] = emb[j]
    return out
