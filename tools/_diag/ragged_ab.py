import argparse, sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from doubleattentionspeakerverification_amd.model import SpeakerClassifier
from doubleattentionspeakerverification_amd.ragged import embed_ragged_batches
DEV = "cuda:0"
p = argparse.Namespace(front_end="VGG4L", kernel_size=1024, pooling_method="DoubleMHA", heads_number=32, mask_prob=0.3,
                       embedding_size=400, num_spkrs=5994, scalingFactor=30.0, marginFactor=0.4, annealing=False)
torch.manual_seed(1234)
net = SpeakerClassifier(p, torch.device(DEV)).to(DEV).eval()
g = np.random.RandomState(1234)
lens = g.randint(150, 801, size=1024)
utts = [torch.randn(int(n), 80, device=DEV) for n in lens]
def run():
    torch.cuda.synchronize(); t = time.time()
    out = embed_ragged_batches(net, utts)
    torch.cuda.synchronize(); return time.time() - t, out
run()
for rnd in range(3):
    for flag in (True, False):
        net.front_end.packed_activations = flag
        dt, out = run()
        print("packed=%s  %.1f ms  %.0f utt/s" % (flag, dt * 1e3, 1024 / dt), flush=True)
net.front_end.packed_activations = True; _, a = run()
net.front_end.packed_activations = False; _, b = run()
print("identical:", all(torch.equal(a[i], b[i]) for i in a))
