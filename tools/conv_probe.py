#!/usr/bin/env python3
"""Probe: hand-written conv3x3+bias+ReLU (split / bf16 modes) vs PyTorch-ROCm fp32 on the VGG4L/1024 layer shapes."""
import os, sys, time, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from doubleattentionspeakerverification_amd import conv

dev = torch.device("cuda:0")
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
layers = [("conv12", 128, 128, 300, 80), ("conv21", 128, 256, 150, 40), ("conv22", 256, 256, 150, 40),
          ("conv31", 256, 512, 75, 20), ("conv32", 512, 512, 75, 20), ("conv41", 512, 1024, 38, 10), ("conv42", 1024, 1024, 38, 10)]

def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

tot = {"ref": 0, "split": 0, "bf16": 0}
for name, cin, cout, H, W in layers:
    x = torch.relu(torch.randn(B, cin, H, W, device=dev)).contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=dev) * (2.0 / (9 * cin)) ** 0.5)   # leaf Parameter: packed planes are cached
    b = torch.randn(cout, device=dev) * 0.1
    pk = conv.PackedConvWeights()
    with torch.no_grad():
        ref = F.relu(F.conv2d(x, w, b, padding=1))
        ys = conv.conv3x3_bias_relu(x, w, b, pk, "split")
        yb = conv.conv3x3_bias_relu(x, w, b, pk, "bf16")
        es = float((ys - ref).abs().max() / ref.abs().max()); eb = float((yb - ref).abs().max() / ref.abs().max())
        t_ref = timeit(lambda: F.relu(F.conv2d(x, w, b, padding=1)))
        t_s = timeit(lambda: conv.conv3x3_bias_relu(x, w, b, pk, "split"))
        t_b = timeit(lambda: conv.conv3x3_bias_relu(x, w, b, pk, "bf16"))
    fl = 2 * 9 * cin * cout * H * W * B
    tot["ref"] += t_ref; tot["split"] += t_s; tot["bf16"] += t_b
    print("%s %4d->%4d %3dx%2d | torch fp32 %6.3f ms %6.1f TF | split %6.3f ms %6.1f TF (x3 MFMA: %6.1f) err %.1e | bf16 %6.3f ms %6.1f TF err %.1e"
          % (name, cin, cout, H, W, t_ref, fl / t_ref / 1e9, t_s, fl / t_s / 1e9, 3 * fl / t_s / 1e9, es, t_b, fl / t_b / 1e9, eb), flush=True)
print("total: torch fp32 %.2f ms | split %.2f ms | bf16 %.2f ms" % (tot["ref"], tot["split"], tot["bf16"]))
