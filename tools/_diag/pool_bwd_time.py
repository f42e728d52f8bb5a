"""The pool's backward scatter (arg-max bytes -> PACKED full-resolution gradient) alone, at the four pools of [64,300,80]: time by events
with the output rotating over three buffers, against the bytes it moves."""
import torch
from doubleattentionspeakerverification_amd import conv

dev = "cuda"
for (H, W, C) in [(300, 80, 64), (150, 40, 128), (75, 20, 256), (38, 10, 512)]:
    B, Ho, Wo = 64, (H + 1) // 2, (W + 1) // 2
    gy = (torch.randn(B, C, Ho, Wo, device=dev) * 1e-3).contiguous(memory_format=torch.channels_last)
    idx = torch.randint(0, 8, (B * Ho * Wo * C,), device=dev, dtype=torch.uint8)
    for want_sum in (True, False):
        for _ in range(3):
            conv._pool_scatter(idx, gy, B, H, W, C, want_sum, True)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        keep = []
        a.record()
        for _ in range(12):
            keep.append(conv._pool_scatter(idx, gy, B, H, W, C, want_sum, True)[0])
            keep = keep[-3:]
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / 12 * 1e3
        by = B * H * W * C * 4 + B * Ho * Wo * C * 5
        print(f"[{B},{H},{W},{C}] colsum={want_sum}: {us:7.1f} us  {by / us / 1e6:5.2f} TB/s")
