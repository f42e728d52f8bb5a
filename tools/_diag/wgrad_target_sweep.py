"""Diagnostic: weight-gradient time vs the number of workgroups the pixel range is split for (DASV_WGRAD_TARGET), interleaved."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from doubleattentionspeakerverification_amd import conv
dev = torch.device("cuda:0")
layers = [("conv12", 128, 128, 300, 80), ("conv21", 128, 256, 150, 40), ("conv22", 256, 256, 150, 40),
          ("conv31", 256, 512, 75, 20), ("conv32", 512, 512, 75, 20), ("conv41", 512, 1024, 38, 10), ("conv42", 1024, 1024, 38, 10)]
targets = [128, 256, 384, 512, 640]
tot = {t: 0.0 for t in targets}
for name, cin, cout, H, W in layers:
    x = torch.relu(torch.randn(64, cin, H, W, device=dev)).contiguous(memory_format=torch.channels_last)
    g = torch.randn(64, cout, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    best = {t: 1e9 for t in targets}
    for t in targets:
        os.environ["DASV_WGRAD_TARGET"] = str(t)
        for _ in range(3): conv.conv3x3_wgrad(x, g, "split")
    torch.cuda.synchronize()
    for rnd in range(3):
        for t in targets:
            os.environ["DASV_WGRAD_TARGET"] = str(t)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(4): conv.conv3x3_wgrad(x, g, "split")
            e.record(); torch.cuda.synchronize()
            best[t] = min(best[t], s.elapsed_time(e) / 4)
    print(name, " ".join("%d:%.3f" % (t, best[t]) for t in targets), flush=True)
    for t in targets: tot[t] += best[t]
print("total", " ".join("%d:%.2f" % (t, tot[t]) for t in targets))
