"""Profiling target: the split-f16 fused MLP kernel alone, 6 launches at the fine-pass size (786 432 samples)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-nerf_amd")]
import torch
from torch_nerf.amd import ops, synth
flat = torch.from_numpy(synth.nerf_flat_params(seed=3, sigma_bias=1.0, sigma_gain=30.0)).cuda()
pk = ops.mlp_pack_f16x2(flat)
M = 4096 * 192
g = torch.Generator(device="cuda").manual_seed(0)
pts = torch.rand(M, 3, device="cuda", generator=g) * 8 - 4
dirs = torch.rand(M, 3, device="cuda", generator=g) * 2 - 1
for _ in range(6):
    ops.mlp_forward_f16x2(pk, pts, dirs)
torch.cuda.synchronize()
