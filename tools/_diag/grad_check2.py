"""Diagnostic: every HIP conv layer's weight/bias gradient inside a full VGG4L backward vs ATen on the same saved tensors."""
import sys, os, copy, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from doubleattentionspeakerverification_amd.CNNs import VGG4L
from doubleattentionspeakerverification_amd import conv as C
DEV = "cuda:0"
torch.manual_seed(5)
net = VGG4L(1024).to(DEV).train()
rs = np.random.RandomState(61)
x = torch.from_numpy(rs.randn(2, 45, 80).astype(np.float32)).to(DEV)
cap = {}
orig = C.conv3x3_bias_relu
def wrapped(xx, w, b, packed=None, mode="split", a=False, c=False):
    y = orig(xx, w, b, packed, mode, a, c)
    name = [k for k, v in net.named_parameters() if v is w][0]
    cap[name] = dict(x=xx.detach(), y=y.detach())
    y.register_hook(lambda g, name=name: cap[name].__setitem__("gy", g.detach().clone()))
    return y
C.conv3x3_bias_relu = wrapped
net.conv_mode = "split"
y = net(x)
gout = torch.from_numpy(rs.randn(*y.shape).astype(np.float32)).to(DEV)
(y * gout).sum().backward()
for name, d in cap.items():
    conv = getattr(net, name.split(".")[0])
    g = d["gy"] * (d["y"] > 0)
    gx, gw, gb = torch.ops.aten.convolution_backward(g, d["x"], conv.weight.detach(), [conv.out_channels], [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, True])
    print(name, "wgrad vs aten on same tensors %.2e  bias %.2e  | x nonfinite %d, g nonfinite %d, |g|max %.3e |x|max %.3e" % (
        float((conv.weight.grad - gw).abs().max() / gw.abs().max()), float((conv.bias.grad - gb).abs().max() / gb.abs().max()),
        int((~torch.isfinite(d["x"])).sum()), int((~torch.isfinite(g)).sum()), float(g.abs().max()), float(d["x"].abs().max())))
