"""Per-workgroup durations of the layered family's dW list kernel (NERF_DW_TIMING dump) on a non-default-encoder network:
balance of the plan (csrc/mlp_backward.hip:plan_dw_items).  usage: python3 scripts/dw_list_timing.py [coord_l12|dir_l5]"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-nerf_amd")]
from torch_nerf.amd import ops, synth
tag = sys.argv[1] if len(sys.argv) > 1 else "coord_l12"
e_p, e_d, lp, ld = {"coord_l12": (75, 27, 12, 4), "dir_l5": (63, 33, 10, 5)}[tag]
M = 4096 * 192
net = ops.Net(e_p, e_d, 256, lp, True, ld, True)
flat = torch.from_numpy(synth.nerf_flat_params(seed=9, pos_dim=e_p, view_dir_dim=e_d, sigma_bias=0.5, sigma_gain=20.0)).cuda()
pts = torch.rand(M, 3, device="cuda") * 8 - 4; dirs = torch.rand(M, 3, device="cuda") * 2 - 1
gs = torch.randn(M, device="cuda"); gc = torch.randn(M, 3, device="cuda")
sigma, rgb, rec = ops.mlp_layered_forward(flat, pts, dirs, net, encoded=False, record=True)
run = lambda: ops.mlp_layered_backward(flat, pts, dirs, net, sigma, rgb, rec, gs, gc)
for _ in range(2): run()
os.environ["NERF_DW_TIMING"] = "/tmp/dw_list_timing.txt"
run(); torch.cuda.synchronize()
del os.environ["NERF_DW_TIMING"]
rows = np.loadtxt("/tmp/dw_list_timing.txt")
print("item  a_w  x_w flags slices   mean_ms   max_ms   min_ms  planned")
for k in sorted(set(rows[:, 0].astype(int))):
    r = rows[rows[:, 0] == k]; t = r[:, 4] / 1e5
    print(f"{k:4d} {int(r[0,1]):4d} {int(r[0,2]):4d} {int(r[0,6]):5d} {len(r):6d} {t.mean():9.3f} {t.max():8.3f} {t.min():8.3f} {int(r[0,5]):8d}")
t = rows[:, 4] / 1e5
print(f"all: {len(t)} workgroups, mean {t.mean():.3f} ms, max {t.max():.3f} ms")
first = rows[:, 0].astype(int)
inner = np.array([0 < b < len(first) - 1 and first[b - 1] == first[b] == first[b + 1] for b in range(len(first))])
ref = None
for k in sorted(set(first)):
    sel = inner & (first == k)
    if sel.any() and rows[sel, 1][0] == 256 and rows[sel, 2][0] == 256 and rows[sel, 6][0] <= 1:
        ref = (np.median(rows[sel, 4]), rows[sel, 5][0]); break
print("item  inner median_ms planned suggested (plain 256 x 256 = 7350)")
for k in sorted(set(first)):
    sel = inner & (first == k)
    if sel.any() and ref:
        tm, c = np.median(rows[sel, 4]), rows[sel, 5][0]
        print(f"{k:4d} {int(sel.sum()):6d} {tm/1e5:9.3f} {int(c):7d} {7350 * tm * c / (ref[0] * ref[1]):9.0f}")
