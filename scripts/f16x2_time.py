"""Time mlp_forward_f16x2 at the bench's launch sizes next to the fp32 and bf16 kernels; max error vs the fp32 kernel.
    python scripts/f16x2_time.py [reps]"""
import sys
import numpy as np
import torch
sys.path.insert(0, "torch-nerf_amd")
from torch_nerf.amd import ops, synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
flat = torch.from_numpy(synth.nerf_flat_params(seed=3, sigma_bias=1.0, sigma_gain=30.0)).cuda()
pk32, pk16, pkx = ops.mlp_pack(flat), ops.mlp_pack_bf16(flat), ops.mlp_pack_f16x2(flat)
for M in (4096 * 64, 4096 * 192):
    g = torch.Generator(device="cuda").manual_seed(M)
    pts = torch.rand((M, 3), device="cuda", generator=g) * 8 - 4
    dirs = torch.nn.functional.normalize(torch.randn((M, 3), device="cuda", generator=g), dim=-1)
    res = {}
    for name, fn in (("fp32", lambda: ops.mlp_forward(pk32, pts, dirs, encoded=False)),
                     ("bf16", lambda: ops.mlp_forward_bf16(pk16, pts, dirs)),
                     ("f16x2", lambda: ops.mlp_forward_f16x2(pkx, pts, dirs))):
        out = fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[name] = (e0.elapsed_time(e1) / reps, out)
    flop = M * 1186816.0
    s32, c32 = res["fp32"][1]
    for name, (ms, (s, c)) in res.items():
        mult = 3.0 if name == "f16x2" else 1.0
        peak = 157.3e12 if name == "fp32" else 2.5e15
        print(f"M {M:7d} {name:6s} {ms:8.4f} ms  {flop / ms / 1e9:8.1f} TFLOP/s algorithmic  frac {mult * flop / (ms * 1e-3) / peak:.3f}"
              f"  max|rgb-fp32| {float((c - c32).abs().max()):.2e}  max rel|sigma| {float(((s - s32).abs() / s32.abs().clamp(min=1)).max()):.2e}")
