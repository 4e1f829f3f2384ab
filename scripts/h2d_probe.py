"""How long do the runner's host<->device statements take on this box when the GPU is idle / busy? (runner_loop attribution)"""
import time, torch
torch.cuda.set_device(0)
x = torch.rand(640000, 3)
idx = torch.randint(0, 640000, (4096,))
torch.cuda.synchronize()
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print("cpu gather x[idx]                    %.3f ms" % t(lambda: x[idx, ...]))
g = x[idx, ...]
print("pageable (4096,3) .cuda(), GPU idle  %.3f ms" % t(lambda: g.cuda()))
p = g.pin_memory()
print("pinned .cuda(non_blocking)           %.3f ms" % t(lambda: p.cuda(non_blocking=True)))
big = torch.empty(64 << 20, device="cuda")
def busy_then_copy():
    for _ in range(20): big.mul_(1.0001)      # ~ a few ms of queued GPU work
    return g.cuda()
def busy_only():
    for _ in range(20): big.mul_(1.0001)
    torch.cuda.synchronize()
print("20 x 256 MB kernels + synchronize    %.3f ms" % t(busy_only, 10))
print("20 x 256 MB kernels + pageable .cuda %.3f ms" % t(busy_then_copy, 10))
s = torch.zeros((), device="cuda")
print(".item() on an idle GPU               %.3f ms" % t(lambda: s.item()))
import numpy as np
print("np.random.choice(640000, 4096)       %.3f ms" % t(lambda: np.random.choice(640000, size=[4096], replace=False), 5))
