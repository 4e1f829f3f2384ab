"""Render one full 800x800 frame (BASELINE configs[4] on however many GPUs are visible to the launcher)
with shard.render_frame and report the frame time; optionally write a PNG.

    python scripts/render_frame.py [--bf16] [--llff] [--png out.png]        # --llff: BASELINE configs[3], 1008x756 NDC
    python scripts/render_frame.py --llff --spiral 4 [--png out_%03d.png]   # 4 evenly spaced poses of the LLFF spiral path
    python -m torch.distributed.run --nproc-per-node 8 scripts/render_frame.py
"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-nerf_amd")]
import torch
import torch.distributed as dist
import torch_nerf.src.network as network
import torch_nerf.src.renderer.cameras as cameras
from torch_nerf.amd import shard, synth, image

ap = argparse.ArgumentParser()
ap.add_argument("--bf16", action="store_true")
ap.add_argument("--png", default=None)
ap.add_argument("--size", type=int, default=800)
ap.add_argument("--llff", action="store_true", help="LLFF fern geometry: 1008x756, forward-facing pose, NDC rays, t in [0,1]")
ap.add_argument("--spiral", type=int, default=0, help="with --llff: render this many poses of the spiral render path "
                "(synth.llff_spiral_poses = load_llff.py:519-559 on a synthetic forward-facing pose set)")
args = ap.parse_args()
world = int(os.environ.get("WORLD_SIZE", "1"))
local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
torch.cuda.set_device(local)
if world > 1:
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
if args.llff:   # W=1008, H=756, f ~ 815, near-identity 3x4 pose, ndc=True with (t_n, t_f) = (0, 1)
    H, W, focal, ndc = 756, 1008, 815.0, True
    cam = cameras.PerspectiveCamera({"f_x": focal, "f_y": focal, "img_width": W, "img_height": H},
                                    torch.from_numpy(synth.llff_like_pose()), 0.0, 1.0)
else:
    H = W = args.size
    focal, ndc = float(synth.blender_focal(W)), False
    cam = cameras.PerspectiveCamera({"f_x": focal, "f_y": focal, "img_width": W, "img_height": H},
                                    torch.from_numpy(synth.pose_spherical(37.0, -30.0, 4.0)), 2.0, 6.0)
nets = []
for seed in (3, 4):
    flat = synth.nerf_flat_params(seed=seed, sigma_bias=1.0, sigma_gain=30.0)
    net = network.NeRF(63, 27)
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.split_flat_params(flat).items()})
    net = net.cuda()
    net.bf16_inference = args.bf16
    nets.append(net)


def frame():
    if args.bf16:
        return shard.render_frame(cam, nets[0], nets[1], 64, 128, ndc, seed=1, bf16=True)
    return shard.render_frame(cam, nets[0], nets[1], 64, 128, ndc, seed=1)


frame(); torch.cuda.synchronize()
if args.llff and args.spiral > 0:   # row f3: walk the render path the LLFF loader would hand to runners/render.py
    import numpy as np
    poses, bounds = synth.llff_like_pose_set(20, seed=0)
    path = synth.llff_spiral_poses(poses, bounds)   # (not recentred: the path stays one unit off the z = 0 plane, see synth)
    for k in np.linspace(0, len(path), args.spiral, endpoint=False).astype(int):
        cam = cameras.PerspectiveCamera({"f_x": focal, "f_y": focal, "img_width": W, "img_height": H},
                                        torch.from_numpy(path[k]), 0.0, 1.0)
        t0 = time.perf_counter(); img = frame(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if int(os.environ.get("RANK", "0")) == 0:
            print(f"spiral pose {k:3d}/{len(path)}: {dt*1e3:.1f} ms, mean colour {[round(c, 4) for c in img.mean(0).tolist()]}, "
                  f"finite {bool(torch.isfinite(img).all())}", flush=True)
            if args.png:
                image.save_png(args.png % k if "%" in args.png else args.png, img.view(H, W, 3))
t0 = time.perf_counter(); img = frame(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
if int(os.environ.get("RANK", "0")) == 0:
    print(f"{W}x{H} {'LLFF-NDC' if args.llff else 'Blender'} frame, 64+128 samples, {'bf16' if args.bf16 else 'fp32'}, {world} GPU(s): {dt*1e3:.1f} ms  "
          f"({H*W/dt/1e3:.1f} k rays/s)", flush=True)
    if args.png:
        image.save_png(args.png, img.view(H, W, 3))
if world > 1:
    dist.destroy_process_group()
