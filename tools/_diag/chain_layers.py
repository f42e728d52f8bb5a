#!/usr/bin/env python3
"""Per-launch timing of the inference front-end chain as shipped (CNNs._forward_map_packed / _forward_map_f16 on [64,300,80]): HIP events
around every conv / pool / conv11 call of the chain, median of 10 passes.  GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from doubleattentionspeakerverification_amd import conv as C
from doubleattentionspeakerverification_amd.CNNs import VGG4L

mode = sys.argv[1] if len(sys.argv) > 1 else "split8"
dev = torch.device("cuda:0")
torch.manual_seed(1)
net = VGG4L(1024).to(dev).eval()
net.conv_mode = mode
x = torch.randn(64, 300, 80, device=dev)
rec = []


def wrap(name, fn):
    def timed(*a, **k):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        y = fn(*a, **k)
        e.record()
        t = a[0]
        rec.append(("%s %s%s" % (name, tuple(t.shape), " +pool" if k.get("pool") else ""), s, e))
        return y
    return timed


for n in ("conv3x3_x8", "conv3x3_f16", "maxpool2x2_ceil", "conv3x3_c1_bias_relu"):
    setattr(C, n, wrap(n, getattr(C, n)))
rows = {}
with torch.no_grad():
    for it in range(13):
        del rec[:]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        net.forward_map(x)
        e1.record()
        torch.cuda.synchronize()
        if it >= 3:
            for i, (k, s, e) in enumerate(rec):
                rows.setdefault((i, k), []).append(s.elapsed_time(e))
            rows.setdefault((99, "whole pass"), []).append(e0.elapsed_time(e1))
for (i, k), v in sorted(rows.items()):
    v = sorted(v)
    print("%-60s %7.3f ms" % (k, v[len(v) // 2]))
