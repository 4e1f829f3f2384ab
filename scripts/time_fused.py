"""Time the fused render pass (fine: 4096 rays x 192 samples; coarse: x 64) next to the bare MLP kernel at the same
sample count.  NERF_AMD_LIB selects a variant library."""
import os, sys, json, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-nerf_amd")]
from torch_nerf.amd import ops, shard, synth
n, Sc, Sf = 4096, 64, 128
flat = torch.from_numpy(synth.nerf_flat_params(seed=4, sigma_bias=1.0, sigma_gain=30.0)).cuda()
pk = ops.mlp_pack(flat)
g = torch.Generator(device="cuda").manual_seed(0)
o = torch.randn(n, 3, device="cuda", generator=g); d = torch.randn(n, 3, device="cuda", generator=g)
t_bins = torch.linspace(2.0, 6.0, Sc + 1, device="cuda")[:-1]
u1c, u1, u2, u3 = shard.ray_draws(3, 0, n, Sc, Sf, "cuda")
_, w = ops.render_rays(pk, o, d, t_bins, 4.0 / Sc, u1c)
def timeit(fn, reps=15):
    for _ in range(3): fn()
    ev = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); ev.append((a, b))
    torch.cuda.synchronize()
    return float(np.median([x.elapsed_time(y) for x, y in ev]))
pts = torch.rand(n * (Sc + Sf), 3, device="cuda") * 4
dirs = torch.rand(n * (Sc + Sf), 3, device="cuda")
out = {"fine_fused_ms": timeit(lambda: ops.render_rays(pk, o, d, t_bins, 4.0 / Sc, u1, weights=w.clone(), u2=u2, u3=u3)),
       "coarse_fused_ms": timeit(lambda: ops.render_rays(pk, o, d, t_bins, 4.0 / Sc, u1c)),
       "mlp_786432_ms": timeit(lambda: ops.mlp_forward(pk, pts, dirs, encoded=False)),
       "mlp_262144_ms": timeit(lambda: ops.mlp_forward(pk, pts[: n * Sc], dirs[: n * Sc], encoded=False)),
       "clone_ms": timeit(lambda: w.clone())}
print(os.environ.get("NERF_AMD_LIB", "default").split("/")[-1], json.dumps({k: round(v, 4) for k, v in out.items()}))
