"""Twenty-five steps of the default model on ONE fixed synthetic batch (B = 16, 50 speakers), once on the hand-written kernels
(split8 forward / data / weight gradients, pool index bytes, fused AM-softmax + CE) and once with PyTorch-ROCm convolutions, same
seeds: the two loss curves must start together and fall alike — a smoke test of the whole training path beyond the unit tests."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from doubleattentionspeakerverification_amd.model import SpeakerClassifier  # noqa: E402
from doubleattentionspeakerverification_amd.dist import DataParallelTrainer as Trainer  # noqa: E402

dev = torch.device("cuda:0")
p = argparse.Namespace(front_end="VGG4L", kernel_size=1024, pooling_method="DoubleMHA", heads_number=32, mask_prob=0.3,
                       embedding_size=400, num_spkrs=50, scalingFactor=30.0, marginFactor=0.4, annealing=False)
rs = np.random.RandomState(0)
x = torch.from_numpy(rs.standard_normal((16, 200, 80)).astype(np.float32)).to(dev)
y = torch.from_numpy(rs.randint(0, 50, size=16)).to(dev)
curves = {}
for mode in ("split8", "torch"):
    torch.manual_seed(0)
    net = SpeakerClassifier(p, dev).to(dev)
    net.front_end.conv_mode = mode
    tr = Trainer(net, lr=1e-4)
    torch.manual_seed(1)                      # the head masks of the training-mode pooling draw from this stream
    losses = []
    for step in range(25):
        loss, _ = tr.train_step(x, y, step=step)
        losses.append(float(loss))
    curves[mode] = losses
    print("%-6s loss: %s" % (mode, " ".join("%.3f" % v for v in losses[::3])))
a, b = np.array(curves["split8"]), np.array(curves["torch"])
print("first step rel diff %.2e, max rel diff over 25 steps %.2e; fall %.3f -> %.3f (hand-written) / %.3f -> %.3f (PyTorch-ROCm convolutions)"
      % (abs(a[0] - b[0]) / b[0], float(np.max(np.abs(a - b) / b)), a[0], a[-1], b[0], b[-1]))
assert np.all(np.isfinite(a)) and a[-1] < a[0] and abs(a[0] - b[0]) / b[0] < 1e-4
print("TRAIN_SANITY_OK")
