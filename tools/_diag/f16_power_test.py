"""The fp16-operand convolution on half-precision activations: random data against all-zero data, same launches (see conv_power_test.py)."""
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
for name, cin, cout, H, W in LAYERS:
    runs, keep = {}, []
    for tag in ("random", "zeros"):
        x = torch.relu(torch.randn(B, H, W, cin, device=dev)).half()
        w = torch.randn(cout, cin, 3, 3, device=dev) * (2.0 / (9 * cin)) ** 0.5
        b = torch.randn(cout, device=dev) * 0.1
        if tag == "zeros":
            x.zero_()
            w.zero_()
        recs = hc.new_ranges(dev, 2)
        recs[0][0] = int(x.float().abs().max().view(torch.int32))
        hi, _ = hc.PackedConvWeights().get(w, "fp16", b)
        y = torch.empty(B, H, W, cout, device=dev, dtype=torch.float16)
        keep.append((x, hi, b, y, recs))

        def run(x=x, hi=hi, b=b, y=y, recs=recs):
            return lib.dasv_conv3x3_bias_relu_f16_nhwc(x.data_ptr(), 1, recs[0].data_ptr(), hi.data_ptr(), b.data_ptr(), y.data_ptr(), 1,
                                                       recs[1].data_ptr(), None, 0, B, H, W, cin, cout, 1, st)
        runs[tag] = run
        assert run() == 0
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
