"""Test helper: record the DECISIONS the hand-written training forward of a VGG front-end takes — the sign of every ReLU
(CNNs.py:38-48 / 72-86) and the element every 2x2 ceil-mode max-pool window hands on (CNNs.py:40,44,48 / 74,78,82,86) — in the
form oracle.vgg_forward_decided takes, so that gradients can be compared element by element with the float64 gradients under
the SAME decisions (no flips by construction) while the decisions themselves are compared with the float64 forward's own and the
disagreements counted (oracle/train_oracle.py)."""
import torch
import torch.nn.functional as F


def packed_sign(t):
    """(v > 0) of a PACKED activation tensor [B,C,H,W] (conv.conv3x3_x8: every 256-byte slab of 64 channels starts with the 64 fp16
    values c = v * 2^E), as a bool tensor of the same shape."""
    B, C, H, W = t.shape
    raw = t.detach().permute(0, 2, 3, 1).contiguous().view(torch.uint8).view(B, H, W, C // 64, 256)[..., :128].contiguous()
    return (raw.view(torch.float16).reshape(B, H, W, C) > 0).permute(0, 3, 1, 2)


class DecisionCapture:
    """with DecisionCapture(monkeypatch, front_end) as cap: y = front_end(x) ...; cap.decisions() -> dict for the oracle (CPU).
    One forward pass per capture (call `reset()` between passes)."""

    def __init__(self, monkeypatch, front_end):
        from doubleattentionspeakerverification_amd import conv as C
        self.C, self.mp, self.fe = C, monkeypatch, front_end
        self.events = []

    def reset(self):
        self.events = []

    def __enter__(self):
        C = self.C
        oc, op, o1, ocp = C.conv3x3_bias_relu, C.maxpool2x2_ceil_autograd, C.conv3x3_c1_bias_relu_autograd, C.conv3x3_bias_relu_pool

        def convpool(x, *a, **k):
            # conv + ReLU + pool in one launch: the only record of its decisions are the bytes {arg-max | 4 if maximum > 0} the
            # launch leaves for its own backward (saved on the autograd node)
            y = ocp(x, *a, **k)
            self.events.append(("convpool", (y.grad_fn.saved_tensors[2].detach(), x.shape[2], x.shape[3])))
            return y

        def conv(*a, **k):
            y = oc(*a, **k)
            self.events.append(("conv", packed_sign(y) if k.get("y_packed") else y.detach()))
            return y

        def c1(*a, **k):
            y = o1(*a, **k)
            self.events.append(("conv", packed_sign(y) if k.get("pack") else y.detach()))
            return y

        def pool(h, *a, **k):
            self.events.append(("pool", h.detach()))
            return op(h, *a, **k)

        self.mp.setattr(C, "conv3x3_bias_relu", conv)
        self.mp.setattr(C, "conv3x3_c1_bias_relu_autograd", c1)
        self.mp.setattr(C, "maxpool2x2_ceil_autograd", pool)
        self.mp.setattr(C, "conv3x3_bias_relu_pool", convpool)
        ofb = C.first_block_train

        def first_block(*a, **k):
            # conv11 + conv12 + pool as one autograd node: conv11's packed output and the pool's bytes are saved on the node
            y = ofb(*a, **k)
            _, h, _, idx = y.grad_fn.saved_tensors
            self.events.append(("conv", packed_sign(h)))
            self.events.append(("convpool", (idx.detach(), h.shape[2], h.shape[3])))
            return y

        self.mp.setattr(C, "first_block_train", first_block)
        ofr = C.first_block_ragged_train

        def first_block_ragged(*a, **k):
            # conv11 + conv12 over a ragged batch as one node: conv11's packed output is saved on it, conv12's fp32 map is the result
            y = ofr(*a, **k)
            self.events.append(("conv", packed_sign(y.grad_fn.saved_tensors[1])))
            self.events.append(("conv", y.detach()))
            return y

        self.mp.setattr(C, "first_block_ragged_train", first_block_ragged)
        return self

    def __exit__(self, *exc):
        self.mp.undo()
        return False

    def passes(self):
        """Decisions of every forward pass captured since reset(), in order (a variable-length minibatch is several dense passes)."""
        out, k = [], 0
        while k < len(self.events):
            n = 0
            for _ in range(self.fe.n_blocks):       # a block is conv, conv, pool — or conv, conv+pool
                n += 2 if self.events[k + n + 1][0] == "convpool" else 3
            out.append(self.decisions(self.events[k:k + n]))
            k += n
        return out

    def decisions_stacked(self, lengths):
        """The captured pass ran on a RAGGED batch (utterances of `lengths` frames stacked along time in one image): the decisions
        per utterance, as oracle.classifier_train_forward takes them for a list."""
        fe = self.fe
        whole = self.decisions()
        lens, per_level = [int(n) for n in lengths], []
        for _ in range(fe.n_blocks):
            per_level.append(lens)
            lens = [(n + 1) // 2 for n in lens]
        out = []
        ev = {}
        names = []
        for blk in range(1, fe.n_blocks + 1):
            names += ["conv%d1" % blk, "conv%d2" % blk, "pool%d" % blk]
        for name, (_, t) in zip(names, self.events):
            ev[name] = t
        for i in range(len(lengths)):
            d = {}
            for blk in range(1, fe.n_blocks + 1):
                r0 = sum(per_level[blk - 1][:i])
                r1 = r0 + per_level[blk - 1][i]
                for name in ("conv%d1" % blk, "conv%d2" % blk):
                    d[name] = whole[name][:, :, r0:r1, :]
                c = getattr(fe, "conv%d2" % blk).out_channels
                d["pool%d" % blk] = F.max_pool2d(ev["pool%d" % blk][:, :c, r0:r1, :].contiguous(), 2, stride=2, ceil_mode=True,
                                                 return_indices=True)[1].cpu()
            out.append(d)
        return out

    def decisions(self, events=None):
        """{'convXY': bool [B,C,H,W], 'poolX': int64 [B,C,Ho,Wo]} on the CPU, real channels only (the zero-padded network of a
        narrow kernel_size carries pad channels that are identically zero)."""
        fe = self.fe
        it = iter(self.events if events is None else events)
        out = {}
        for blk in range(1, fe.n_blocks + 1):
            n1, n2, pn = "conv%d1" % blk, "conv%d2" % blk, "pool%d" % blk
            kind, t = next(it)
            assert kind == "conv", kind
            out[n1] = (t[:, :getattr(fe, n1).out_channels] > 0).cpu()
            kind, t = next(it)
            c = getattr(fe, n2).out_channels
            if kind == "convpool":
                code, H, W = t
                code = code.cpu().permute(0, 3, 1, 2)[:, :c].long()                     # [B, C, Ho, Wo]
                Ho, Wo = code.shape[2], code.shape[3]
                ho = torch.arange(Ho).view(1, 1, Ho, 1)
                wo = torch.arange(Wo).view(1, 1, 1, Wo)
                pos = code & 3
                out[pn] = (2 * ho + (pos >> 1)) * W + 2 * wo + (pos & 1)
                # the ReLU's sign is known for the element each window hands on (nothing else of that map has any effect)
                mask = torch.zeros(code.shape[0], c, H * W, dtype=torch.bool)
                out[n2] = mask.scatter_(2, out[pn].flatten(2), (code & 4).flatten(2) > 0).view(code.shape[0], c, H, W)
                continue
            assert kind == "conv", kind
            out[n2] = (t[:, :c] > 0).cpu()
            kind, t = next(it)
            assert kind == "pool", kind
            # ATen's arg-max (first maximum in window order) is the kernels' own tie-breaking: bit-exact pools are tested in
            # test_hip_maxpool_and_conv11_backward_match_torch
            out[pn] = F.max_pool2d(t[:, :c].contiguous(), 2, stride=2, ceil_mode=True, return_indices=True)[1].cpu()
        assert next(it, None) is None
        return out
