"""Time the bf16 fused MLP kernel alone (M = 4096 x 64 and 4096 x 192).  NERF_AMD_LIB selects a variant build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-nerf_amd")]
import torch
from torch_nerf.amd import ops, synth

flat = torch.from_numpy(synth.nerf_flat_params(seed=9, sigma_bias=0.5, sigma_gain=20.0)).cuda()
packed = ops.mlp_pack_bf16(flat)
FLOP = 2 * 593408
for M in (4096 * 64, 4096 * 192):
    pts = torch.rand(M, 3, device="cuda") * 8 - 4
    dirs = torch.rand(M, 3, device="cuda") * 2 - 1
    for _ in range(3):
        ops.mlp_forward_bf16(packed, pts, dirs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    K = 20
    e0.record()
    for _ in range(K):
        ops.mlp_forward_bf16(packed, pts, dirs)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / K
    print(f"{os.environ.get('NERF_AMD_LIB', 'default')}: M={M} {ms:.4f} ms  {M * FLOP / ms / 1e9:.1f} TFLOP/s  frac={M * FLOP / ms / 1e9 / 2500:.3f}", flush=True)
