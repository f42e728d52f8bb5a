"""The weight-gradient launches (fp16 + e4m3 form) on random and on all-zero operands: same instruction stream and traffic.  What
zeros gain is clock the chip withholds on real data (see conv_power_test.py)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from doubleattentionspeakerverification_amd import conv as hc  # noqa: E402

dev = torch.device("cuda:0")
B = 64
LAYERS = [("conv12", 128, 128, 300, 80), ("conv22", 256, 256, 150, 40), ("conv32", 512, 512, 75, 20), ("conv42", 1024, 1024, 38, 10)]
torch.manual_seed(0)
for name, cin, cout, H, W in LAYERS:
    data = {}
    for tag in ("random", "zeros"):
        x = torch.relu(torch.randn(B, cin, H, W, device=dev)).contiguous(memory_format=torch.channels_last)
        g = torch.randn(B, cout, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        if tag == "zeros":
            x.zero_()
            g.zero_()
        data[tag] = (x, g, hc.range_of(x), hc.range_of(g))
    best = {"random": 1e9, "zeros": 1e9}
    for rnd in range(4):
        for tag in (("random", "zeros") if rnd % 2 == 0 else ("zeros", "random")):
            x, g, xr, gr = data[tag]
            for _ in range(2):
                hc.conv3x3_wgrad(x, g, "split8", x_rec=xr, g_rec=gr)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                hc.conv3x3_wgrad(x, g, "split8", x_rec=xr, g_rec=gr)
            e1.record()
            torch.cuda.synchronize()
            best[tag] = min(best[tag], e0.elapsed_time(e1) / 5)
    print("%s: random data %.4f ms   all-zero data %.4f ms   (zeros %+.1f %%)" % (name, best["random"], best["zeros"], (best["zeros"] / best["random"] - 1) * 100), flush=True)
