"""Time the fused MLP forward at the bench shapes and print error stats vs the oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-nerf_amd")]
import numpy as np, torch
from torch_nerf.amd import ops, synth

flat = synth.nerf_flat_params(seed=9, sigma_bias=0.5, sigma_gain=20.0)
packed = ops.mlp_pack(torch.from_numpy(flat).cuda())
FLOP = 2 * 593408
for M in (4096 * 64, 4096 * 192):
    pts = (torch.rand(M, 3, device="cuda") * 8 - 4)
    dirs = (torch.rand(M, 3, device="cuda") * 2 - 1)
    for save in (False, True):
        for _ in range(3):
            ops.mlp_forward(packed, pts, dirs, False, save=save)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        K = 10
        for _ in range(K):
            out = ops.mlp_forward(packed, pts, dirs, False, save=save)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / K
        print(f"M={M} save={save}: {ms:.3f} ms  {M*FLOP/ms/1e9:.1f} TFLOP/s  frac={M*FLOP/ms/1e9/157.3:.3f}", flush=True)
if "--check" in sys.argv:
    from oracle import oracle as O
    M = 2048
    pts = np.random.RandomState(0).uniform(-4, 4, (M, 3)).astype(np.float32)
    dirs = np.random.RandomState(1).uniform(-1, 1, (M, 3)).astype(np.float32)
    s, r = ops.mlp_forward(packed, torch.from_numpy(pts).cuda(), torch.from_numpy(dirs).cuda(), False)
    so, ro = O.mlp_forward(flat, O.posenc(pts, 10), O.posenc(dirs, 4))
    print("max |dsigma|", np.abs(s.cpu().numpy() - so).max(), "max |drgb|", np.abs(r.cpu().numpy() - ro).max(), "sigma range", so.min(), so.max())

if "--bf16" in sys.argv:
    pb = ops.mlp_pack_bf16(torch.from_numpy(flat).cuda())
    for M in (4096 * 64, 4096 * 192):
        pts = (torch.rand(M, 3, device="cuda") * 8 - 4)
        dirs = (torch.rand(M, 3, device="cuda") * 2 - 1)
        for _ in range(3):
            ops.mlp_forward_bf16(pb, pts, dirs)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            out = ops.mlp_forward_bf16(pb, pts, dirs)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"bf16 M={M}: {ms:.3f} ms  {M*FLOP/ms/1e9:.1f} TFLOP/s  (x{157.3:.0f} fp32 peak = {M*FLOP/ms/1e9/157.3:.2f}; of 2500 bf16 peak = {M*FLOP/ms/1e9/2500:.3f})", flush=True)
