"""Per-workgroup durations of the dW kernel (NERF_DW_TIMING dump): balance of the (layer, slice) plan."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-nerf_amd")]
from torch_nerf.amd import ops, synth
M = 4096 * 192
flat = torch.from_numpy(synth.nerf_flat_params(seed=9, sigma_bias=0.5, sigma_gain=20.0)).cuda()
packed = ops.mlp_pack(flat)
X2 = "--f16x2" in sys.argv          # the split-f16 backward's dW kernel (mlp_bwd_dw_x2_kernel) instead of the fp32 one
kw = {"packed_f16x2": ops.mlp_pack_f16x2(flat)} if X2 else {}
pts = torch.rand(M, 3, device="cuda") * 8 - 4; dirs = torch.rand(M, 3, device="cuda") * 2 - 1
gs = torch.randn(M, device="cuda"); gc = torch.randn(M, 3, device="cuda")
sigma, rgb, saved = ops.mlp_forward(packed, pts, dirs, False, save=True)
for _ in range(2): ops.mlp_backward(packed, flat, pts, dirs, False, sigma, rgb, saved, gs, gc, **kw)
os.environ["NERF_DW_TIMING"] = "/tmp/dw_timing.txt"
ops.mlp_backward(packed, flat, pts, dirs, False, sigma, rgb, saved, gs, gc, **kw)
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
# workgroups strictly inside one item (their neighbours start in the same item): time ~ tiles x true cost, tiles ~ 1 / planned cost
first = rows[:, 0].astype(int)
inner = np.array([0 < b < len(first) - 1 and first[b - 1] == first[b] == first[b + 1] for b in range(len(first))])
shapes = sorted(set((int(r[1]), int(r[2])) for r in rows))
ref = None
print("shape      inner  median_ms  planned  suggested (256 x 256 = 7350)")
est = {}
for aw, xw in shapes:
    sel = inner & (rows[:, 1] == aw) & (rows[:, 2] == xw)
    if not sel.any():
        continue
    est[(aw, xw)] = (np.median(rows[sel, 4]) / 1e5, rows[sel, 5][0], int(sel.sum()))
if (256, 256) in est:
    t0, c0, _ = est[(256, 256)]
    for (aw, xw), (tm, c, n) in est.items():
        print(f"{aw:3d} x {xw:3d} {n:6d} {tm:10.3f} {int(c):8d}  {7350 * (tm * c) / (t0 * c0):8.0f}")
