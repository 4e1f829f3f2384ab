"""Per-workgroup durations of the dW kernel (NERF_DW_TIMING dump): balance of the (layer, slice) plan."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-nerf_amd")]
from torch_nerf.amd import ops, synth
M = 4096 * 192
flat = torch.from_numpy(synth.nerf_flat_params(seed=9, sigma_bias=0.5, sigma_gain=20.0)).cuda()
packed = ops.mlp_pack(flat)
pts = torch.rand(M, 3, device="cuda") * 8 - 4; dirs = torch.rand(M, 3, device="cuda") * 2 - 1
gs = torch.randn(M, device="cuda"); gc = torch.randn(M, 3, device="cuda")
sigma, rgb, saved = ops.mlp_forward(packed, pts, dirs, False, save=True)
for _ in range(2): ops.mlp_backward(packed, flat, pts, dirs, False, sigma, rgb, saved, gs, gc)
os.environ["NERF_DW_TIMING"] = "/tmp/dw_timing.txt"
ops.mlp_backward(packed, flat, pts, dirs, False, sigma, rgb, saved, gs, gc)
torch.cuda.synchronize()
del os.environ["NERF_DW_TIMING"]
rows = np.loadtxt("/tmp/dw_timing.txt")
print("item  a_w  x_w slices   mean_ms   max_ms   min_ms")
for k in sorted(set(rows[:, 0].astype(int))):
    r = rows[rows[:, 0] == k]
    t = r[:, 4] / 1e5   # 100 MHz ticks -> ms
    print(f"{k:4d} {int(r[0,1]):4d} {int(r[0,2]):4d} {len(r):6d} {t.mean():9.3f} {t.max():8.3f} {t.min():8.3f}")
t = rows[:, 4] / 1e5
print(f"all: {len(t)} workgroups, mean {t.mean():.3f} ms, max {t.max():.3f} ms, sum/256 {t.sum()/256:.3f} ms")
