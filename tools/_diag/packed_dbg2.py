import os, sys, torch
sys.path.insert(0, "/root/repo")
from doubleattentionspeakerverification_amd.CNNs import VGG4L
from doubleattentionspeakerverification_amd import conv as hc
DEV = "cuda:0"
torch.manual_seed(7)
fe = VGG4L(1024).to(DEV).eval()
for B, T in ((24, 300), (2, 61)):
    mel = torch.randn(B, T, 80, device=DEV) * 2.0
    with torch.no_grad():
        h = hc.conv3x3_c1_bias_relu(mel, fe.conv11.weight, fe.conv11.bias)
        hp = hc.conv3x3_c1_bias_relu(mel, fe.conv11.weight, fe.conv11.bias, pack=True)
        for name in ("conv12", "conv21", "conv22", "conv31", "conv32", "conv41", "conv42"):
            c = getattr(fe, name)
            yf = hc.conv3x3_x8(h, c.weight, c.bias, None)
            yp = hc.conv3x3_x8(hp, c.weight, c.bias, None, x_packed=True)
            d = (yp - yf).abs()
            bad = (d > 0).nonzero()
            print(B, T, name, "max diff %.3e of %.3e, %d of %d differ" % (float(d.max()), float(yf.abs().max()), bad.shape[0], d.numel()),
                  "first bad (b,c,h,w):", bad[0].tolist() if bad.shape[0] else None, "last:", bad[-1].tolist() if bad.shape[0] else None)
            if name.endswith("2"):
                h = hc.maxpool2x2_ceil(yf); hp = hc.maxpool2x2_ceil(yf, pack=True)
            else:
                h = yf; hp = hc.conv3x3_x8(hp, c.weight, c.bias, None, x_packed=True, y_packed=True)
