#!/usr/bin/env python3
"""Training steps (record forward, integrator + MLP backward, FusedAdam) on a non-default-encoder scene pair, for
`rocprofv3 --kernel-trace --stats`.  usage: python3 scripts/probe_encoders_train.py [coord_l12|dir_l5|sh]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-nerf_amd")]
import torch  # noqa: E402
import bench  # noqa: E402
import torch_nerf.src.network as network  # noqa: E402
import torch_nerf.src.scene as scene  # noqa: E402
from torch_nerf.src.signal_encoder import PositionalEncoder, SHEncoder  # noqa: E402
from torch_nerf.amd import synth  # noqa: E402
from torch_nerf.amd.optim import FusedAdam  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "coord_l12"
ce, de = {"coord_l12": (PositionalEncoder(3, 12, True), PositionalEncoder(3, 4, True)),
          "dir_l5": (PositionalEncoder(3, 10, True), PositionalEncoder(3, 5, True)),
          "sh": (SHEncoder(3, 4), SHEncoder(3, 4))}[tag]
device = torch.device("cuda", 0)
renderer = bench.build_scene(device)[0]
scenes = []
for seed in (3, 4):
    flat = synth.nerf_flat_params(seed=seed, pos_dim=ce.out_dim, view_dir_dim=de.out_dim, sigma_bias=1.0, sigma_gain=30.0)
    net = network.NeRF(ce.out_dim, de.out_dim)
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.split_flat_params(flat, ce.out_dim, de.out_dim, 256).items()})
    scenes.append(scene.PrimitiveCube(net.to(device), {"coord_enc": ce, "dir_enc": de}))
opt = FusedAdam([p for sc in scenes for p in sc.radiance_field.parameters()], lr=5e-4, eps=1e-8)
mse, gt = torch.nn.MSELoss(), torch.rand((bench.RAYS, 3), device=device)
pix = torch.arange(bench.RAYS, device=device)
for _ in range(5):
    opt.zero_grad(set_to_none=True)
    c_rgb, c_idx, c_w = renderer.render_scene(scenes[0], bench.RAYS, 64, False, 0, pixel_indices=pix)
    f_rgb, _, _ = renderer.render_scene(scenes[1], bench.RAYS, (64, 128), False, 0, pixel_indices=c_idx, weights=c_w)
    (mse(gt, c_rgb) + mse(gt, f_rgb)).backward()
    opt.step()
torch.cuda.synchronize()
print("ok", tag)
