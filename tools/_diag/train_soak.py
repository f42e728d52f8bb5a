"""Diagnostic: 120 training steps of the default model on a fixed synthetic batch through DataParallelTrainer (all-HIP
front-end): the loss must stay finite and go down."""
import argparse, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from doubleattentionspeakerverification_amd.model import SpeakerClassifier
from doubleattentionspeakerverification_amd.dist import DataParallelTrainer
dev = torch.device("cuda:0")
p = argparse.Namespace(front_end="VGG4L", kernel_size=1024, pooling_method="DoubleMHA", heads_number=32, mask_prob=0.3,
                       embedding_size=400, num_spkrs=5994, scalingFactor=30.0, marginFactor=0.4, annealing=False)
torch.manual_seed(1234)
net = SpeakerClassifier(p, dev).to(dev).train()
rs = np.random.RandomState(1)
x = torch.from_numpy(rs.standard_normal((64, 300, 80)).astype(np.float32)).to(dev)
y = torch.from_numpy(rs.randint(0, 5994, size=(64,))).to(dev)
tr = DataParallelTrainer(net, lr=1e-4, weight_decay=1e-3)
hist = []
for step in range(120):
    loss, _ = tr.train_step(x, y, step=step)
    if step % 10 == 0 or step == 119:
        hist.append(float(loss))
        print(step, hist[-1], flush=True)
assert all(np.isfinite(hist)) and hist[-1] < hist[0] - 1.0, hist
print("SOAK_OK")
