import time, torch, numpy as np
dev = torch.device("cuda:0")
x = torch.randn(64, 1024, 1024, device=dev)
def busy():
    for _ in range(20): y = x @ x
def t(fn, n=20):
    torch.cuda.synchronize(); busy(); t0 = time.time()
    for _ in range(n): fn()
    dt = (time.time() - t0) / n * 1e3; torch.cuda.synchronize(); return dt
buf = torch.empty(1 << 18, dtype=torch.int32).pin_memory()
arr = np.arange(100000, dtype=np.int32)
print("pin_memory() same size     %.3f ms" % t(lambda: torch.from_numpy(arr).pin_memory()))
print("copy_ into pinned slice    %.3f ms" % t(lambda: buf[:100000].copy_(torch.from_numpy(arr))))
print("slice.to(non_blocking)     %.3f ms" % t(lambda: buf[:100000].to(dev, non_blocking=True)))
print("full.to(non_blocking)      %.3f ms" % t(lambda: buf.to(dev, non_blocking=True)))
print("pageable .to()             %.3f ms" % t(lambda: torch.from_numpy(arr).to(dev)))
d = torch.empty(100000, dtype=torch.int32, device=dev)
print("d.copy_(slice, nb)         %.3f ms" % t(lambda: d.copy_(buf[:100000], non_blocking=True)))
def ev():
    e = torch.cuda.Event(); e.record(torch.cuda.current_stream(dev)); return e
print("event create+record        %.3f ms" % t(ev))
print("is_pinned(slice)", buf[:100000].is_pinned())
small = torch.tensor(list(range(40)), dtype=torch.int64)
print("small pin_memory().to(nb)  %.3f ms" % t(lambda: small.pin_memory().to(dev, non_blocking=True)))
