#!/usr/bin/env python3
"""The whole 800x800 frame through the CLASS API (the two render_scene calls of runner_utils.py:872-908, num_ray_batch as the
runners pass it) for a non-default encoder setting, next to shard.render_frame on the same scenes and draws.
usage: python3 scripts/frame_via_api_encoders.py [coord_l12|dir_l5|sh]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-nerf_amd")]
import torch  # noqa: E402
import bench  # noqa: E402
import torch_nerf.src.network as network  # noqa: E402
import torch_nerf.src.scene as scene  # noqa: E402
from torch_nerf.src.signal_encoder import PositionalEncoder, SHEncoder  # noqa: E402
from torch_nerf.amd import shard, synth  # noqa: E402

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
H = W = bench.H
with torch.no_grad():
    for rep in range(2):
        torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats()
        t0 = time.perf_counter()
        c_rgb, idx, c_w = renderer.render_scene(scenes[0], H * W, 64, False, 0, num_ray_batch=H * W // 4096)
        f_rgb, _, _ = renderer.render_scene(scenes[1], H * W, (64, 128), False, 0, pixel_indices=idx, weights=c_w,
                                            num_ray_batch=H * W // 4096)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(tag, "frame via render_scene: %.1f ms, peak %.2f GiB, finite %s, range [%.3f, %.3f]" % (
        dt * 1e3, torch.cuda.max_memory_allocated() / 2**30, bool(torch.isfinite(f_rgb).all()), float(f_rgb.min()), float(f_rgb.max())))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    img = shard.render_frame(renderer.camera, scenes[0], scenes[1], 64, 128, False, seed=1, single_rank=True)
    torch.cuda.synchronize()
    print(tag, "frame via shard.render_frame: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
