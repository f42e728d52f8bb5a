"""Time of the split8 weight packs (forward planes; data-gradient planes with and without the transposed scratch copy)."""
import torch
from doubleattentionspeakerverification_amd import conv, _lib


def t(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for cout, cin in [(128, 64), (128, 128), (256, 128), (256, 256), (512, 256), (512, 512), (1024, 512)]:
    w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    b = torch.zeros(cout, device="cuda")
    row = [f"{cout}x{cin}"]
    if cin % 64 == 0:
        lib = _lib.load()
        n = lib.dasv_conv3x3_packed_weight_bytes_x8(cout, cin)
        hi, lo = torch.empty(n, dtype=torch.uint8, device="cuda"), torch.empty(n, dtype=torch.uint8, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        row.append("fwd %.1f us" % t(lambda: lib.dasv_conv3x3_pack_weights_x8(w.data_ptr(), b.data_ptr(), hi.data_ptr(), lo.data_ptr(), cout, cin, st)))
    for thr in (1 << 40, 0):
        conv._DGRAD_SCRATCH_ABOVE = thr
        row.append(("dgrad direct %.1f us" if thr else "dgrad via scratch %.1f us") % t(lambda: conv._pack_dgrad_x8(w)))
    print("  ".join(row))
