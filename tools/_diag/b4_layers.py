import os, sys, torch
sys.path.insert(0, "/root/repo")
from doubleattentionspeakerverification_amd import conv as hc, _lib
lib = _lib.load()
dev = torch.device("cuda:0")
torch.manual_seed(0)
layers = [("conv12", 128, 128, 300, 80), ("conv21", 128, 256, 150, 40), ("conv22", 256, 256, 150, 40),
          ("conv31", 256, 512, 75, 20), ("conv32", 512, 512, 75, 20), ("conv41", 512, 1024, 38, 10), ("conv42", 1024, 1024, 38, 10)]
def t(fn, n=20):
    with torch.no_grad():
        for _ in range(3): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for B in (4, 2):
    tot = {"split": 0, "split8": 0}
    for name, cin, cout, H, W in layers:
        x = torch.relu(torch.randn(B, cin, H, W, device=dev)).contiguous(memory_format=torch.channels_last)
        w = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=dev) * 0.03)
        b = torch.zeros(cout, device=dev)
        r = {}
        for m in tot:
            pk = hc.PackedConvWeights()
            r[m] = t(lambda: hc.conv3x3_bias_relu(x, w, b, pk, m))
            tot[m] += r[m]
        print(B, name, "ws bytes", lib.dasv_conv3x3_workspace_bytes(B, H, W, cin, cout), " split %.3f  split8 %.3f" % (r["split"], r["split8"]))
    print(B, "total", tot)
