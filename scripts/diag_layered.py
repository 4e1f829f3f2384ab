"""Per-tensor worst deviations of the layered family's gradients from the CPU oracle (forced masks)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-nerf_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from torch_nerf.amd import ops, synth
from oracle import oracle as O
from helpers import layered_masks
O.build()
dims = tuple(int(v) for v in sys.argv[1].split(",")); M = int(sys.argv[2])
e_p, e_d, feat = dims
rng = np.random.RandomState(M + feat)
pe = rng.uniform(-1, 1, (M, e_p)).astype(np.float32); de = rng.uniform(-1, 1, (M, e_d)).astype(np.float32)
gs, gc = rng.standard_normal(M).astype(np.float32), rng.standard_normal((M, 3)).astype(np.float32)
flat = synth.nerf_flat_params(seed=21, pos_dim=e_p, view_dir_dim=e_d, feat_dim=feat, sigma_bias=0.3, sigma_gain=6.0)
spec = ops.Net.dims_only(*dims)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
fp = dev(flat)
sigma, rgb, rec = ops.mlp_layered_forward(fp, dev(pe), dev(de), spec, record=True)
masks = layered_masks(rec, sigma, M, spec)
want_g, _, _, _ = O.mlp_backward_ex(flat, pe, de, gs, gc, F=feat, force_masks=masks)
outs = []
for rep in range(2):
    got, _, _ = ops.mlp_layered_backward(fp, dev(pe), dev(de), spec, sigma, rgb, rec, dev(gs), dev(gc))
    outs.append(got.cpu().numpy())
print("deterministic:", np.array_equal(outs[0], outs[1]))
G, W = synth.split_flat_params(outs[0], *dims), synth.split_flat_params(want_g, *dims)
for k in G:
    a, b = G[k], W[k]
    rms = np.sqrt(np.mean(b.astype(np.float64) ** 2)) + 1e-30
    err = np.abs(a - b) / (2e-5 * np.abs(b) + 2e-5 * rms)
    idx = np.unravel_index(np.argsort(err.reshape(-1))[-3:], err.shape)
    print(f"{k:14s} rms {rms:.3e} worst x bound {err.max():.2f}  99.9% {np.quantile(err, 0.999):.2f}  at {[tuple(int(v[j]) for v in idx) for j in range(3)]}")
