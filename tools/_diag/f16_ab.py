"""Same-process A/B of the fp16-operand inference step: fp16 activation storage (CNNs._forward_map_f16) against fp32 storage."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from doubleattentionspeakerverification_amd.model import SpeakerClassifier
p = argparse.Namespace(front_end="VGG4L", kernel_size=1024, pooling_method="DoubleMHA", heads_number=32, mask_prob=0.3,
                       embedding_size=400, num_spkrs=5994, scalingFactor=30.0, marginFactor=0.4, annealing=False)
torch.manual_seed(1234)
dev = torch.device("cuda:0")
net = SpeakerClassifier(p, dev).to(dev).eval()
net.front_end.conv_mode = "fp16"
x = torch.randn(64, 300, 80, device=dev)
best = {True: 1e9, False: 1e9}
with torch.no_grad():
    for rnd in range(6):
        for half in ((True, False) if rnd % 2 == 0 else (False, True)):
            net.front_end.half_activations = half
            for _ in range(3):
                net.getEmbedding(x)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                net.getEmbedding(x)
            e1.record()
            torch.cuda.synchronize()
            best[half] = min(best[half], e0.elapsed_time(e1) / 10)
print("fp16 operands, [64,300,80]: fp16 storage %.3f ms (%.0f utt/s)   fp32 storage %.3f ms (%.0f utt/s)"
      % (best[True], 64e3 / best[True], best[False], 64e3 / best[False]))
# per-kernel view of one step each
from torch.profiler import profile, ProfilerActivity
for half in (True, False):
    net.front_end.half_activations = half
    with torch.no_grad(), profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(3):
            net.getEmbedding(x)
        torch.cuda.synchronize()
    print("half_activations =", half)
    for ev in sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:6]:
        print("   %-70s n=%3d  avg %.1f us" % (ev.key[:70], ev.count, ev.device_time_total / ev.count))
