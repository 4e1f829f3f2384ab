"""Time one training step (coarse + fine forward with record, loss, backward) and its pieces."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-nerf_amd")]
import numpy as np, torch
from torch_nerf.amd import ops, synth

flat = synth.nerf_flat_params(seed=9, sigma_bias=0.5, sigma_gain=20.0)
fp = torch.from_numpy(flat).cuda()
packed = ops.mlp_pack(fp)
FWD, BWD = 2 * 593408, 2 * 1151104


def timeit(fn, K=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(K):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K


for M in (4096 * 64, 4096 * 192):
    pts = torch.rand(M, 3, device="cuda") * 8 - 4
    dirs = torch.rand(M, 3, device="cuda") * 2 - 1
    gs = torch.randn(M, device="cuda"); gc = torch.randn(M, 3, device="cuda")
    sigma, rgb, saved = ops.mlp_forward(packed, pts, dirs, False, save=True)
    ms = timeit(lambda: ops.mlp_backward(packed, fp, pts, dirs, False, sigma, rgb, saved, gs, gc))
    print(f"M={M} backward: {ms:.3f} ms  {M*BWD/ms/1e9:.1f} TFLOP/s (algorithmic) frac={M*BWD/ms/1e9/157.3:.3f}", flush=True)
