import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from doubleattentionspeakerverification_amd import conv
from doubleattentionspeakerverification_amd.CNNs import VGG4L
DEV = "cuda:0"
torch.manual_seed(0)
for cin, cout in ((2, 2), (32, 64), (64, 32), (64, 128), (64, 64)):
    x = torch.randn(2, cin, 10, 6, device=DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(cout, cin, 3, 3, device=DEV) * 0.2).requires_grad_(True)
    b = (torch.randn(cout, device=DEV) * 0.1).requires_grad_(True)
    gy = torch.randn(2, cout, 10, 6, device=DEV)
    y = conv.conv3x3_bias_relu(x, w, b, None, "split")[:, :cout]
    (y * gy).sum().backward()
    g1 = [t.grad.clone() for t in (x, w, b)]
    for t in (x, w, b):
        t.grad = None
    yr = torch.relu(torch.nn.functional.conv2d(x, w, b, padding=1))
    (yr * gy).sum().backward()
    print(cin, cout, "fwd", float((y - yr).abs().max() / yr.abs().max()),
          [float((a - r).abs().max() / r.abs().max()) for a, r in zip(g1, (x.grad, w.grad, b.grad))])
for mode in ("split", "split8"):
    torch.manual_seed(5)
    net = VGG4L(256).to(DEV).train()
    net.conv_mode = mode
    ref = copy.deepcopy(net)
    ref.conv_mode = "torch"
    x = torch.randn(2, 45, 80, device=DEV)
    gout = None
    grads = {}
    for tag, m in (("hip", net), ("torch", ref)):
        y = m(x)
        if gout is None:
            gout = torch.randn_like(y)
        (y * gout).sum().backward()
        grads[tag] = {k: v.grad.clone() for k, v in m.named_parameters()}
    print(mode, {k: "%.1e" % float((grads["hip"][k] - grads["torch"][k]).abs().max() / grads["torch"][k].abs().max()) for k in grads["torch"]})
