import argparse, sys, time, numpy as np, torch, cProfile, pstats
sys.path.insert(0, "/root/repo")
from doubleattentionspeakerverification_amd.model import SpeakerClassifier
from doubleattentionspeakerverification_amd import ragged
DEV = "cuda:0"
p = argparse.Namespace(front_end="VGG4L", kernel_size=1024, pooling_method="DoubleMHA", heads_number=32, mask_prob=0.3,
                       embedding_size=400, num_spkrs=5994, scalingFactor=30.0, marginFactor=0.4, annealing=False)
torch.manual_seed(1234)
net = SpeakerClassifier(p, torch.device(DEV)).to(DEV).eval()
g = np.random.RandomState(1234)
lens = g.randint(150, 801, size=1024)
utts = [torch.randn(int(n), 80, device=DEV) for n in lens]
ragged.embed_ragged_batches(net, utts); torch.cuda.synchronize()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    out = ragged.embed_ragged_batches(net, utts)
    t1 = time.time(); torch.cuda.synchronize(); t2 = time.time()
    print("enqueue %.1f ms, total %.1f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3), flush=True)
pr = cProfile.Profile(); pr.enable()
out = ragged.embed_ragged_batches(net, utts); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
