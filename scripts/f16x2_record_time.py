"""Time the record (training) forward: fp32 kernel vs the split-f16 kernel, at the bench's two launch sizes."""
import sys
import torch
sys.path.insert(0, "torch-nerf_amd")
from torch_nerf.amd import ops, synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
flat = torch.from_numpy(synth.nerf_flat_params(seed=3, sigma_bias=1.0, sigma_gain=30.0)).cuda()
pk32, pkx = ops.mlp_pack(flat), ops.mlp_pack_f16x2(flat)
for M in (4096 * 64, 4096 * 192):
    g = torch.Generator(device="cuda").manual_seed(M)
    pts = torch.rand((M, 3), device="cuda", generator=g) * 8 - 4
    dirs = torch.nn.functional.normalize(torch.randn((M, 3), device="cuda", generator=g), dim=-1)
    for name, fn in (("fp32 record", lambda: ops.mlp_forward(pk32, pts, dirs, False, save=True)),
                     ("f16x2 record", lambda: ops.mlp_forward_f16x2(pkx, pts, dirs, save=True)),
                     ("f16x2 plain", lambda: ops.mlp_forward_f16x2(pkx, pts, dirs))):
        out = fn(); del out
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            out = fn(); del out
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"M {M:7d} {name:13s} {ms:8.4f} ms   record bytes {M * 10400 / 1e9:.2f} GB -> {M * 10400 / ms / 1e9:.2f} TB/s")
