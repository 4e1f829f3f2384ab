#!/usr/bin/env python3
"""One non-default-encoder scene through the headline step (coarse + fine render_scene, 4096 rays), a few steps, for
`rocprofv3 --kernel-trace --stats`: where the kernel chain's time goes.  usage: python3 scripts/probe_encoders.py [coord_l12|dir_l5|sh]"""
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
pix = torch.arange(bench.RAYS, device=device)
with torch.no_grad():
    for _ in range(6):
        bench.render_step(renderer, scenes[0], scenes[1], pix, 0)
torch.cuda.synchronize()
print("ok", tag)
