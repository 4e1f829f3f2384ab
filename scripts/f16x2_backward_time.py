"""Backward of one launch, fp32 kernels vs NeRF.f16x2_training's split-f16 kernels, at the bench's two launch sizes.
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "torch-nerf_amd"))
from torch_nerf.amd import ops, synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
flat = torch.from_numpy(synth.nerf_flat_params(seed=3, sigma_bias=1.0, sigma_gain=30.0)).cuda()
pk32, pkx = ops.mlp_pack(flat), ops.mlp_pack_f16x2(flat)
for M in (4096 * 64, 4096 * 192):
    g = torch.Generator(device="cuda").manual_seed(M)
    pts = torch.rand((M, 3), device="cuda", generator=g) * 8 - 4
    dirs = torch.nn.functional.normalize(torch.randn((M, 3), device="cuda", generator=g), dim=-1)
    gs = torch.randn(M, device="cuda", generator=g)
    gc = torch.randn((M, 3), device="cuda", generator=g)
    s, c, rec = ops.mlp_forward(pk32, pts, dirs, False, save=True)
    for name, kw in (("fp32 backward", {}), ("f16x2 backward", {"packed_f16x2": pkx})):
        fn = lambda: ops.mlp_backward(pk32, flat, pts, dirs, False, s, c, rec, gs, gc, **kw)
        out = fn(); del out
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            out = fn(); del out
        e1.record(); torch.cuda.synchronize()
        print(f"M {M:7d} {name:15s} {e0.elapsed_time(e1) / reps:8.4f} ms")
