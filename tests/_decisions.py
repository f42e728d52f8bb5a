"""Test helper: record the DECISIONS the hand-written training forward of a VGG front-end takes — the sign of every ReLU
(CNNs.py:38-48 / 72-86) and the element every 2x2 ceil-mode max-pool window hands on (CNNs.py:40,44,48 / 74,78,82,86) — in the
form oracle.vgg_forward_decided takes, so that gradients can be compared element by element with the float64 gradients under
the SAME decisions (no flips by construction) while the decisions themselves are compared with the float64 forward's own and the
disagreements counted (oracle/train_oracle.py)."""
import torch
import torch.nn.functional as F


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
        oc, op, o1 = C.conv3x3_bias_relu, C.maxpool2x2_ceil_autograd, C.conv3x3_c1_bias_relu_autograd

        def conv(*a, **k):
            y = oc(*a, **k)
            self.events.append(("conv", y.detach()))
            return y

        def c1(*a, **k):
            y = o1(*a, **k)
            self.events.append(("conv", y.detach()))
            return y

        def pool(h, *a, **k):
            self.events.append(("pool", h.detach()))
            return op(h, *a, **k)

        self.mp.setattr(C, "conv3x3_bias_relu", conv)
        self.mp.setattr(C, "conv3x3_c1_bias_relu_autograd", c1)
        self.mp.setattr(C, "maxpool2x2_ceil_autograd", pool)
        return self

    def __exit__(self, *exc):
        self.mp.undo()
        return False

    def passes(self):
        """Decisions of every forward pass captured since reset(), in order (a variable-length minibatch is several dense passes)."""
        n = 3 * self.fe.n_blocks
        assert len(self.events) % n == 0, len(self.events)
        return [self.decisions(self.events[k:k + n]) for k in range(0, len(self.events), n)]

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
        events = self.events if events is None else events
        names = []
        for blk in range(1, fe.n_blocks + 1):
            names += [("conv", "conv%d1" % blk), ("conv", "conv%d2" % blk), ("pool", "pool%d" % blk)]
        assert [k for k, _ in events] == [k for k, _ in names], [k for k, _ in events]
        out = {}
        c = None
        for (kind, name), (_, t) in zip(names, events):
            if kind == "conv":
                c = getattr(fe, name).out_channels
                out[name] = (t[:, :c] > 0).cpu()
            else:
                # ATen's arg-max (first maximum in window order) is the kernels' own tie-breaking: bit-exact pools are tested in
                # test_hip_maxpool_and_conv11_backward_match_torch
                out[name] = F.max_pool2d(t[:, :c].contiguous(), 2, stride=2, ceil_mode=True, return_indices=True)[1].cpu()
        return out
