"""Is the split8 convolution held back by the chip's power management?  The same launches (same instruction stream, same memory
traffic) on random ReLU activations and on ALL-ZERO activations and weights: the matrix pipe draws far less on zeros, so a launch
that gets faster on zeros was running at a lowered clock, and one that does not is limited by its own stalls."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from doubleattentionspeakerverification_amd import _lib, conv as hc  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
B = 64
LAYERS = [("conv12", 128, 128, 300, 80), ("conv22", 256, 256, 150, 40), ("conv32", 512, 512, 75, 20), ("conv42", 1024, 1024, 38, 10)]
torch.manual_seed(0)


def prep(cin, cout, H, W, zero):
    xf = torch.relu(torch.randn(B, 2 * H, 2 * W, cin, device=dev))
    w = torch.randn(cout, cin, 3, 3, device=dev) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn(cout, device=dev) * 0.1
    if zero:
        xf.zero_()
        w.zero_()
    recs = hc.new_ranges(dev, 3)
    recs[0] = hc.range_of(xf) if not zero else recs[0]
    xn = torch.empty(B, H, W, cin, device=dev)
    assert lib.dasv_maxpool2x2_ceil_x8_nhwc(xf.data_ptr(), recs[0].data_ptr(), xn.data_ptr(), recs[1].data_ptr(), B, 2 * H, 2 * W, cin, st) == 0
    wh, wl = hc.PackedConvWeights().get(w, "split8", b)
    y = torch.empty(B, H, W, cout, device=dev)

    def run():
        return lib.dasv_conv3x3_bias_relu_x8_nhwc(xn.data_ptr(), 1, recs[1].data_ptr(), wh.data_ptr(), wl.data_ptr(), b.data_ptr(), y.data_ptr(), 0,
                                                  recs[2].data_ptr(), None, 0, B, H, W, cin, cout, 1, st)
    return run, (xn, wh, wl, b, y, recs)


for name, cin, cout, H, W in LAYERS:
    runs = {}
    keep = []
    for tag, zero in (("random", False), ("zeros", True)):
        runs[tag], k = prep(cin, cout, H, W, zero)
        keep.append(k)
        assert runs[tag]() == 0
    torch.cuda.synchronize()
    best = {"random": 1e9, "zeros": 1e9}
    for rnd in range(4):
        for tag in (("random", "zeros") if rnd % 2 == 0 else ("zeros", "random")):
            for _ in range(5):
                runs[tag]()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                runs[tag]()
            e1.record()
            torch.cuda.synchronize()
            best[tag] = min(best[tag], e0.elapsed_time(e1) / 20)
    print("%s: random data %.4f ms   all-zero data %.4f ms   (zeros %+.1f %%)" % (name, best["random"], best["zeros"], (best["zeros"] / best["random"] - 1) * 100), flush=True)
