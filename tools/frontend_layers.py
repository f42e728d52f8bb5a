#!/usr/bin/env python3
"""Per-layer timing of the VGG4L/1024 front-end through PyTorch-ROCm (MIOpen) on [64,300,80]: events around each op."""
import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from doubleattentionspeakerverification_amd.CNNs import VGG4L
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = VGG4L(1024).to(dev).eval()
x = torch.randn(64, 300, 80, device=dev)
for dt in (torch.float32, torch.bfloat16):
    rows = {}
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=(dt == torch.bfloat16)):
        for it in range(4):
            h = x.reshape(64, 1, 300, 80).contiguous(memory_format=torch.channels_last)
            for blk in range(1, 5):
                for j in (1, 2):
                    conv = getattr(net, "conv%d%d" % (blk, j))
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record(); h = F.relu(conv(h)); e.record(); torch.cuda.synchronize()
                    fl = 2 * 9 * conv.in_channels * conv.out_channels * h.size(2) * h.size(3) * 64
                    if it == 3: rows["conv%d%d %4d->%4d %3dx%2d" % (blk, j, conv.in_channels, conv.out_channels, h.size(2), h.size(3))] = (s.elapsed_time(e), fl)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(); h = F.max_pool2d(h, 2, stride=2, ceil_mode=True); e.record(); torch.cuda.synchronize()
                if it == 3: rows["pool%d" % blk] = (s.elapsed_time(e), 0)
    print("dtype", dt)
    tot = 0
    for k, (ms, fl) in rows.items():
        tot += ms
        print("  %-28s %7.3f ms  %7.1f TFLOP/s" % (k, ms, fl / ms / 1e9 if fl else 0))
    print("  total %.3f ms" % tot)
