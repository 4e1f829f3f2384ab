#!/usr/bin/env python3
"""Train the coarse + fine NeRF on the procedural scene with the device-resident training step and
report PSNR on held-out views.

    python scripts/train_procedural.py --steps 2000 --size 200 --out gpurun_out/train_proc
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 scripts/train_procedural.py ...

Schedule and hyper-parameters are the reference's (configs/train_params/nerf.yaml, configs/renderer: 4096
pixels, 64 + 128 samples, Adam 5e-4 -> 5e-5 over 300 k iterations, centre-crop batches at the start --
the reference crops for 10 epochs, here for the first --crop-steps steps).  One JSON line at the end.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-nerf_amd")]

import numpy as np
import torch
import torch.distributed as dist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--size", type=int, default=200)
    ap.add_argument("--views", type=int, default=24)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--crop-steps", type=int, default=200)
    ap.add_argument("--eval-every", type=int, default=500)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--f16x2-training", action="store_true", help="record forward of every step on the split-f16 kernel "
                    "and its backward on the split-f16 kernels (NeRF.f16x2_training, round 6)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "train_proc"))
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group(args.backend, **({"device_id": dev} if args.backend == "nccl" else {}))

    from torch_nerf.amd import image, procedural, shard, train
    from torch_nerf.amd.optim import FusedAdam
    from torch_nerf.src.network import NeRF
    from torch_nerf.src.renderer.cameras import PerspectiveCamera

    size = args.size
    images, poses, focal = procedural.make_views(args.views, size, size, dev)
    held_images, held_poses, _ = procedural.make_views(3, size, size, dev, theta_offset=360.0 / args.views / 2)
    data = train.DeviceImages(images, poses, size, size, focal)

    def camera(pose):
        return PerspectiveCamera({"f_x": focal, "f_y": focal, "img_width": size, "img_height": size}, pose, 2.0, 6.0)

    torch.manual_seed(args.seed)                        # same initial weights on every rank
    coarse, fine = NeRF(63, 27).to(dev), NeRF(63, 27).to(dev)
    coarse.f16x2_training = fine.f16x2_training = bool(args.f16x2_training)
    opt = FusedAdam(list(coarse.parameters()) + list(fine.parameters()), lr=5e-4, eps=1e-8)
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, pow(0.00005 / 0.0005, 1 / 300000))
    gen = torch.Generator(device=dev).manual_seed(args.seed)   # identical on every rank: all agree on the batch
    view_rng = np.random.RandomState(args.seed)

    def evaluate(tag):
        psnrs = []
        for i in range(held_images.shape[0]):
            frame = shard.render_frame(camera(held_poses[i]), coarse, fine, 64, 128, False, seed=1234)
            mse = torch.mean((frame - held_images[i]) ** 2).item()
            psnrs.append(-10.0 * np.log10(mse))
            if rank == 0 and i == 0:
                os.makedirs(args.out, exist_ok=True)
                image.save_png(os.path.join(args.out, f"held0_{tag}.png"), frame.view(size, size, 3))
                if tag == "step0":
                    image.save_png(os.path.join(args.out, "held0_truth.png"), held_images[0].view(size, size, 3))
        return float(np.mean(psnrs))

    log = [{"step": 0, "psnr_heldout": evaluate("step0")}]
    window = []
    torch.cuda.synchronize()
    t0, train_s = time.perf_counter(), 0.0
    for step in range(args.steps):
        view = int(view_rng.randint(len(data)))
        pix = train.choose_pixels(size, size, args.rays, gen, centre_crop=step < args.crop_steps)
        c_sse, f_sse = train.train_step(camera(data.poses[view]), coarse, fine, opt, data.images[view], pix, 64, 128,
                                        False, seed=args.seed, step=step, scheduler=sched)
        window.append(f_sse)
        if (step + 1) % args.eval_every == 0 or step + 1 == args.steps:
            torch.cuda.synchronize()
            train_s += time.perf_counter() - t0
            sse = torch.stack(window).sum()
            if world > 1:
                dist.all_reduce(sse)
            entry = {"step": step + 1, "train_mse_fine": sse.item() / (3 * args.rays * len(window)),
                     "psnr_heldout": evaluate(f"step{step + 1}"), "lr": opt.param_groups[0]["lr"]}
            log.append(entry)
            window = []
            if rank == 0:
                print(json.dumps(entry), flush=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
    # round 6: the TRAINED networks on the split-f16 kernel (operands split in two f16 parts, csrc/mlp_forward_f16x2.hip) --
    # trained weights are not nn.Linear's initial distribution: per-layer scales, activations and densities differ
    split = None
    if world == 1:
        per_view = []
        for i in range(held_images.shape[0]):
            a = shard.render_frame(camera(held_poses[i]), coarse, fine, 64, 128, False, seed=1234, single_rank=True)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            b = shard.render_frame(camera(held_poses[i]), coarse, fine, 64, 128, False, seed=1234, single_rank=True, f16x2=True)
            torch.cuda.synchronize()
            dt_x = time.perf_counter() - t1
            err = (a - b).abs().max(dim=1).values
            per_view.append({"psnr_fp32": float(-10.0 * np.log10(torch.mean((a - held_images[i]) ** 2).item())),
                             "psnr_f16x2": float(-10.0 * np.log10(torch.mean((b - held_images[i]) ** 2).item())),
                             "median_abs_diff": float(err.median().item()), "max_abs_diff": float(err.max().item()),
                             "pixels_beyond_1e-5": int((err > 1e-5).sum().item()), "pixels": int(err.numel()),
                             "finite": bool(torch.isfinite(b).all()), "f16x2_frame_ms": dt_x * 1e3})
        split = {"what": "held-out views of the trained networks: fp32 kernels vs the split-f16 kernel, same rays and draws",
                 "views": per_view,
                 "max_abs_weight": max(float(p.detach().abs().max()) for n in (coarse, fine) for p in n.parameters())}
    result = {"what": "procedural scene, coarse + fine NeRF, device-resident training step", "world": world,
              "f16x2_training": bool(args.f16x2_training),
              "size": size, "views": args.views, "rays_per_step": args.rays, "steps": args.steps,
              "train_seconds": train_s, "ms_per_step": 1e3 * train_s / args.steps,
              "rays_per_s": args.rays * args.steps / train_s, "psnr_heldout_start": log[0]["psnr_heldout"],
              "psnr_heldout_end": log[-1]["psnr_heldout"], "f16x2_on_trained_networks": split, "log": log}
    if rank == 0:
        os.makedirs(args.out, exist_ok=True)
        with open(os.path.join(args.out, "train_procedural.json"), "w") as f:
            json.dump(result, f, indent=1)
        print(json.dumps({k: v for k, v in result.items() if k != "log"}), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
