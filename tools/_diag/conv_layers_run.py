"""The seven split8 layers of the default model on [64,300,80] shapes, packed input, three launches each (for rocprofv3 --pmc passes).
argv[1] (optional): a variant library built from the current sources."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from doubleattentionspeakerverification_amd import _lib, conv as hc  # noqa: E402

vp, ci, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
new = _lib.load()
lib = new
if len(sys.argv) > 1:
    lib = ctypes.CDLL(sys.argv[1])
    lib.dasv_conv3x3_bias_relu_x8_nhwc.restype = ci
    lib.dasv_conv3x3_bias_relu_x8_nhwc.argtypes = [vp, ci, vp, vp, vp, vp, vp, ci, vp, vp, sz, ci, ci, ci, ci, ci, ci, vp]
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
B = 64
LAYERS = [("conv12", 128, 128, 300, 80, 0), ("conv21", 128, 256, 150, 40, 1), ("conv22", 256, 256, 150, 40, 0),
          ("conv31", 256, 512, 75, 20, 1), ("conv32", 512, 512, 75, 20, 0), ("conv41", 512, 1024, 38, 10, 1),
          ("conv42", 1024, 1024, 38, 10, 0)]
torch.manual_seed(0)
for name, cin, cout, H, W, ypk in LAYERS:
    xf = torch.relu(torch.randn(B, 2 * H, 2 * W, cin, device=dev))
    w = torch.randn(cout, cin, 3, 3, device=dev) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn(cout, device=dev) * 0.1
    recs = hc.new_ranges(dev, 3)
    recs[0] = hc.range_of(xf)
    xn = torch.empty(B, H, W, cin, device=dev)
    assert new.dasv_maxpool2x2_ceil_x8_nhwc(xf.data_ptr(), recs[0].data_ptr(), xn.data_ptr(), recs[1].data_ptr(), B, 2 * H, 2 * W, cin, st) == 0
    whn, wln = hc.PackedConvWeights().get(w, "split8", b)
    yn = torch.empty(B, H, W, cout, device=dev)
    for _ in range(3):
        assert lib.dasv_conv3x3_bias_relu_x8_nhwc(xn.data_ptr(), 1, recs[1].data_ptr(), whn.data_ptr(), wln.data_ptr(), b.data_ptr(),
                                                  yn.data_ptr(), ypk, recs[2].data_ptr(), None, 0, B, H, W, cin, cout, 1, st) == 0
    torch.cuda.synchronize()
print("done")
