"""Full 800x800 frame through the drop-in VolumeRenderer.render_scene API (the way runners/render.py:81-100 does it),
timed against shard.render_frame."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-nerf_amd")]
import torch
import bench

dev = torch.device("cuda", 0)
renderer, scene_c, scene_f, nets, flats, cam, focal, pose = bench.build_scene(dev)
H = W = 800
for batches in (None, 16):
    with torch.no_grad():
        for it in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            c_rgb, c_idx, c_w = renderer.render_scene(scene_c, H * W, 64, False, 0, num_ray_batch=batches)
            f_rgb, _, _ = renderer.render_scene(scene_f, H * W, (64, 128), False, 0, pixel_indices=c_idx, weights=c_w,
                                                num_ray_batch=batches)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"render_scene API, num_ray_batch={batches}: {dt * 1e3:.1f} ms  ({H * W / dt / 1e3:.1f} k rays/s), peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
