#!/usr/bin/env python3
"""The four HBM-bound stage kernels at frame scale, a few launches each, for `rocprofv3 --pmc` / `--kernel-trace`
(bench.py:hbm_stages_at times the same calls with HIP events).  usage: python3 scripts/probe_stages.py [n_rays] [reps]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-nerf_amd")]
import torch  # noqa: E402
from torch_nerf.amd import _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 640000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
Sc, Sf = 64, 128
S = Sc + Sf
dev = torch.device("cuda", 0)
lib = _lib.load()
g = torch.Generator(device=dev).manual_seed(5)
o, d = torch.randn((n, 3), device=dev, generator=g), torch.randn((n, 3), device=dev, generator=g)
t_bins = torch.linspace(2.0, 6.0, Sc + 1, device=dev)[:-1].contiguous()
ps = 4.0 / Sc
u1, u2, u3 = (torch.rand((n, k), device=dev, generator=g) for k in (Sc, Sf, Sf))
w = torch.rand((n, Sc), device=dev, generator=g)
sigma = torch.rand((n, S), device=dev, generator=g) * 3
rad = torch.rand((n, S, 3), device=dev, generator=g)
delta = torch.full((n, S), 4.0 / S, device=dev)
g_rgb = torch.randn((n, 3), device=dev, generator=g)
pts, dirs = torch.empty((n, S, 3), device=dev), torch.empty((n, S, 3), device=dev)
dl, wo, gs = (torch.empty((n, S), device=dev) for _ in range(3))
rgb, gc = torch.empty((n, 3), device=dev), torch.empty((n, S, 3), device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(reps):
    assert lib.nerf_sample_stratified(P(o), P(d), n, Sc, P(t_bins), ps, P(u1), None, P(pts), P(dirs), P(dl), st) == 0
    assert lib.nerf_sample_hierarchical(P(o), P(d), n, Sc, Sf, P(t_bins), ps, P(w), P(u1), P(u2), P(u3), None, None, P(pts),
                                        P(dirs), P(dl), st) == 0
    assert lib.nerf_composite_forward(P(sigma), P(rad), P(delta), n, S, P(rgb), P(wo), st) == 0
    assert lib.nerf_composite_backward(P(sigma), P(rad), P(delta), P(g_rgb), None, n, S, P(gs), P(gc), st) == 0
    big = torch.empty((n * S * 6,), device=dev)          # reference rates on the same box: a pure write and a copy
    big.fill_(1.0)
    big[: big.numel() // 2].copy_(big[big.numel() // 2:])
torch.cuda.synchronize()
print("ok", n, reps)
