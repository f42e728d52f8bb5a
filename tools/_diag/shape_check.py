"""Diagnostic: forward / weight-gradient / data-gradient accuracy of the conv kernels vs float64 on small-batch shapes."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from doubleattentionspeakerverification_amd import conv as C
DEV = "cuda:0"
torch.manual_seed(0)
for (B, cin, cout, H, W) in [(2, 512, 512, 12, 20), (2, 256, 512, 12, 20), (2, 512, 1024, 6, 10), (2, 256, 256, 23, 40), (2, 128, 128, 45, 80), (2, 128, 256, 23, 40)]:
    x = torch.relu(torch.randn(B, cin, H, W, device=DEV)).contiguous(memory_format=torch.channels_last)
    g = torch.randn(B, cout, H, W, device=DEV).contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, 3, 3, device=DEV) * 0.05
    xd, gd, wd = x.double().cpu(), g.double().cpu(), w.double().cpu().requires_grad_(True)
    xd.requires_grad_(True)
    y = torch.nn.functional.conv2d(xd, wd, None, padding=1)
    (y * gd).sum().backward()
    gw = C.conv3x3_wgrad(x, g, "split").double().cpu()
    wt = w.transpose(0, 1).flip(2, 3).contiguous()
    hi, lo = C.PackedConvWeights().get(wt, "split")
    gx = C._launch(g, hi, lo, None, cin, "split", relu=0).double().cpu()
    yf = C.conv3x3_bias_relu(x, w, None, None, "split").double().cpu()
    yr = torch.relu(y.detach())
    print((B, cin, cout, H, W), "fwd %.2e wgrad %.2e dgrad %.2e" % (
        float((yf - yr).abs().max() / yr.abs().max()),
        float((gw - wd.grad).abs().max() / wd.grad.abs().max()), float((gx - xd.grad).abs().max() / xd.grad.abs().max())))
