"""Render throughput (coarse + fine pass, fp32 and bf16) versus ray-batch size, through the fused
nerf_render_rays entry point.  One line per batch size."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-nerf_amd")]
import torch
from torch_nerf.amd import ops, synth

dev = torch.device("cuda", 0)
flat_c = torch.from_numpy(synth.nerf_flat_params(seed=3, sigma_bias=1.0, sigma_gain=30.0)).to(dev)
flat_f = torch.from_numpy(synth.nerf_flat_params(seed=4, sigma_bias=1.0, sigma_gain=30.0)).to(dev)
packs = {False: (ops.mlp_pack(flat_c), ops.mlp_pack(flat_f)), True: (ops.mlp_pack_bf16(flat_c), ops.mlp_pack_bf16(flat_f))}
t_bins = torch.linspace(2.0, 6.0, 65, device=dev)[:-1]
ps = 4.0 / 64
for n in (256, 1024, 4096, 16384, 65536, 262144):
    o = torch.tensor([0.0, 0.0, 4.0], device=dev).repeat(n, 1) + 0.01 * torch.randn(n, 3, device=dev)
    d = torch.nn.functional.normalize(torch.randn(n, 3, device=dev) * 0.2 + torch.tensor([0.0, 0.0, -1.0], device=dev), dim=-1)
    u1c, u1, u2, u3 = (torch.rand(n, k, device=dev) for k in (64, 64, 128, 128))
    row = [f"rays {n:7d}"]
    for bf16 in (False, True):
        pc, pf = packs[bf16]
        def step():
            _, w = ops.render_rays(pc, o, d, t_bins, ps, u1c, bf16=bf16)
            ops.render_rays(pf, o, d, t_bins, ps, u1, weights=w, u2=u2, u3=u3, bf16=bf16)
        for _ in range(3):
            step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        K = max(3, min(50, int(2e6 / n)))
        for _ in range(K):
            step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
        row.append(f"{'bf16' if bf16 else 'fp32'} {dt * 1e3:8.3f} ms {n / dt / 1e3:9.1f} k rays/s")
    print("   ".join(row), flush=True)
