"""Same-box A/B of the variable-length (ragged) embedding pass in conv_mode "fp16": fp32 activation storage against the
half-precision activation chain (ragged._front_end_ragged_f16).  1024 utterances, T uniform in [150, 800], groups of <= 19200 frames."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from doubleattentionspeakerverification_amd.model import SpeakerClassifier  # noqa: E402
from doubleattentionspeakerverification_amd.ragged import embed_ragged_batches  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
import argparse  # noqa: E402
p = argparse.Namespace(front_end="VGG4L", kernel_size=1024, pooling_method="DoubleMHA", heads_number=32, mask_prob=0.3,
                       embedding_size=400, num_spkrs=5994, scalingFactor=30.0, marginFactor=0.4, annealing=False)
net = SpeakerClassifier(p, dev).to(dev).eval()
rs = np.random.RandomState(0)
lens = rs.randint(150, 801, size=1024)
utts = [torch.from_numpy((2.0 * rs.standard_normal((int(n), 80))).astype(np.float32)).to(dev) for n in lens]
utts = [u - u.mean(dim=0, keepdim=True) for u in utts]
net.front_end.conv_mode = "fp16"


def run(half, n=3):
    net.front_end.half_activations = half
    embed_ragged_batches(net, utts, max_frames=64 * 300)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        e = embed_ragged_batches(net, utts, max_frames=64 * 300)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, e


best = {False: 1e9, True: 1e9}
for rnd in range(3):
    for half in ((False, True) if rnd % 2 == 0 else (True, False)):
        ms, e = run(half)
        best[half] = min(best[half], ms)
        if half:
            e16 = e
        else:
            e32 = e
print("fp32 storage %.2f ms (%.0f utt/s)   fp16 storage %.2f ms (%.0f utt/s)   embeddings max rel diff %.2e"
      % (best[False], 1024e3 / best[False], best[True], 1024e3 / best[True], max(float((e16[i] - e32[i]).abs().max() / e32[i].abs().max()) for i in e32)))
