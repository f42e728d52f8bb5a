#!/usr/bin/env python3
"""Probe: hand-written conv3x3 weight gradient vs ATen/MIOpen fp32 on the VGG4L/1024 layer shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from doubleattentionspeakerverification_amd import conv
dev = torch.device("cuda:0")
torch.manual_seed(0)
B = 64
layers = [("conv12", 128, 128, 300, 80), ("conv21", 128, 256, 150, 40), ("conv22", 256, 256, 150, 40),
          ("conv31", 256, 512, 75, 20), ("conv32", 512, 512, 75, 20), ("conv41", 512, 1024, 38, 10), ("conv42", 1024, 1024, 38, 10)]
def timeit(fn, n=4):
    for _ in range(2): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
tot = [0, 0, 0]
for name, cin, cout, H, W in layers:
    x = torch.relu(torch.randn(B, cin, H, W, device=dev)).contiguous(memory_format=torch.channels_last)
    g = torch.randn(B, cout, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, 3, 3, device=dev)
    def aten():
        return torch.ops.aten.convolution_backward(g, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    ref = aten()
    mine = conv.conv3x3_wgrad(x, g, "split")
    err = float((mine - ref).abs().max() / ref.abs().max())
    t_a, t_s, t_b = timeit(aten), timeit(lambda: conv.conv3x3_wgrad(x, g, "split")), timeit(lambda: conv.conv3x3_wgrad(x, g, "bf16"))
    fl = 2 * 9 * cin * cout * H * W * B
    tot[0] += t_a; tot[1] += t_s; tot[2] += t_b
    print("%s %4d->%4d %3dx%2d | aten fp32 %6.3f ms %6.1f TF | split %6.3f ms %6.1f TF (x3: %6.1f) diff-vs-aten %.1e | bf16 %6.3f ms %6.1f TF"
          % (name, cin, cout, H, W, t_a, fl / t_a / 1e9, t_s, fl / t_s / 1e9, 3 * fl / t_s / 1e9, err, t_b, fl / t_b / 1e9), flush=True)
print("total: aten %.2f ms | split %.2f ms | bf16 %.2f ms" % tuple(tot))
