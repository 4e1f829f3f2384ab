"""Cycle timeline of workgroup 0 of the fused render pass (needs an -DX_FUSED_TIMELINE build:
NERF_AMD_LIB=torch-nerf_amd/lib/variants/tl.so).  Prints the phases of the first groups of a fine pass."""
import ctypes, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-nerf_amd")]
from torch_nerf.amd import _lib, ops, shard, synth
n, Sc, Sf = 4096, 64, 128
pk = ops.mlp_pack(torch.from_numpy(synth.nerf_flat_params(seed=4, sigma_bias=1.0, sigma_gain=30.0)).cuda())
g = torch.Generator(device="cuda").manual_seed(0)
o = torch.randn(n, 3, device="cuda", generator=g); d = torch.randn(n, 3, device="cuda", generator=g)
t_bins = torch.linspace(2.0, 6.0, Sc + 1, device="cuda")[:-1]
u1c, u1, u2, u3 = shard.ray_draws(3, 0, n, Sc, Sf, "cuda")
_, w = ops.render_rays(pk, o, d, t_bins, 4.0 / Sc, u1c)
lib = _lib.load()
lib.nerf_debug_stamps.restype = ctypes.c_int
buf = (ctypes.c_ulonglong * 512)()
lib.nerf_debug_stamps(buf, 1)
ops.render_rays(pk, o, d, t_bins, 4.0 / Sc, u1, weights=w.clone(), u2=u2, u3=u3)
k = lib.nerf_debug_stamps(buf, 1)
st = np.array(buf[:k], dtype=np.int64)
names = ["bunch start", "floor+jitter", "pdf", "cdf", "search", "sort", "sampling done", "barrier",
         "group 0 tiles", "barrier", "group 0 integral", "group 1 tiles", "barrier", "group 1 integral"]
per = len(names)
print(f"{k} stamps; fine pass, workgroup 0, wave 0 (ray 0 of each bunch of 4 rays = 2 groups of 2 rays); cycles per phase")
for bi in range(min(3, (k - 1) // per)):
    s = st[bi * per:(bi + 1) * per + 1]
    print(f"bunch {bi}: " + ", ".join(f"{names[i + 1]} {int(s[i + 1] - s[i])}" for i in range(per - 1)),
          f"| to next bunch {int(s[per] - s[per - 1])}")
