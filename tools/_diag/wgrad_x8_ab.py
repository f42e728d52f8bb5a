"""Same-process timing of the weight-gradient kernel on the seven wide layers of the default model ([64,300,80] batch): split-bf16
(three bf16 MFMAs per product) against the fp16 + e4m3 form, with the error of each against a float64 reference on a small slice."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from doubleattentionspeakerverification_amd import conv as hc  # noqa: E402

dev = torch.device("cuda:0")
B = 64
LAYERS = [("conv12", 128, 128, 300, 80), ("conv21", 128, 256, 150, 40), ("conv22", 256, 256, 150, 40), ("conv31", 256, 512, 75, 20),
          ("conv32", 512, 512, 75, 20), ("conv41", 512, 1024, 38, 10), ("conv42", 1024, 1024, 38, 10)]
torch.manual_seed(0)
tot = {"split": 0.0, "split8": 0.0}
for name, cin, cout, H, W in LAYERS:
    x = torch.relu(torch.randn(B, cin, H, W, device=dev)).contiguous(memory_format=torch.channels_last)
    g = (torch.randn(B, cout, H, W, device=dev) * (torch.rand(B, cout, H, W, device=dev) > 0.5)).contiguous(memory_format=torch.channels_last)
    xr, gr = hc.range_of(x), hc.range_of(g)
    outs = {}
    best = {"split": 1e9, "split8": 1e9}
    for rnd in range(3):
        for mode in (("split", "split8") if rnd % 2 == 0 else ("split8", "split")):
            for _ in range(2):
                outs[mode] = hc.conv3x3_wgrad(x, g, mode, x_rec=xr, g_rec=gr)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                hc.conv3x3_wgrad(x, g, mode, x_rec=xr, g_rec=gr)
            e1.record()
            torch.cuda.synchronize()
            best[mode] = min(best[mode], e0.elapsed_time(e1) / 5)
    d = float((outs["split"] - outs["split8"]).abs().max() / outs["split"].abs().max())
    for m in best:
        tot[m] += best[m]
    print("%s %4d->%4d %3dx%2d  split %.4f ms  split8 %.4f ms  (%+.1f %%)  max rel diff between them %.1e"
          % (name, cin, cout, H, W, best["split"], best["split8"], (best["split8"] / best["split"] - 1) * 100, d), flush=True)
print("seven layers: split %.3f ms  split8 %.3f ms  (%+.1f %%)" % (tot["split"], tot["split8"], (tot["split8"] / tot["split"] - 1) * 100))
