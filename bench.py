#!/usr/bin/env python3
"""Headline benchmark: rays/s on 4096-ray x (64 coarse + 128 fine)-sample batches.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: one rank per GPU over RCCL.  Under torch.distributed.run the ranks are used as given; started as a
     plain `python bench.py --gpus N` the script launches its own N ranks as a CHILD torch.distributed.run
     process before anything touches the GPU and leaves with the child's return code)

Workload (BASELINE.json configs[1]): Blender-lego geometry, 800x800, focal 1111.1, t in [2,6],
no NDC, fp32, synthetic pose / pixels / weights (there is no dataset on the GPU box).
One step = one VolumeRenderer.render_scene coarse pass (64 samples, coarse net) + one fine pass
(64+128 sorted samples, fine net) over a 4096-ray batch per GPU, through the drop-in class API,
forward only (rendering), inputs resident in HBM.  With N GPUs every rank renders its own
4096-ray slab of the frame (weak scaling) and one all-gather assembles the N*4096 colours.

The JSON line also carries
  roofline     : the dominant kernel (fused posenc+MLP, fine-pass launch) against the fp32 MFMA
                 peak, timed live with HIP events on the launch stream inside the timed region
  cpu_baseline : the eager-PyTorch CPU port of the reference path (oracle/torch_port.py) on the
                 host cores, same batch, bounded sample; cpu_baseline_1thread: the same with the
                 reference's own default torch.set_num_threads(1) (runners/runner_utils.py:427)
  frame        : BASELINE configs[4] -- one 800x800 frame, rays sharded over the ranks, the RCCL all-gather
                 included (STRONG scaling: fixed 640 000 rays), image hash checked against the 1-rank image
  hbm_stages   : the HBM-bound stages (sampling, integral fwd/bwd) at full size against 8 TB/s
  ms_per_step_median : median of per-step HIP-event times (the headline value stays total rays / total time)
"""
import argparse
import datetime
import hashlib
import json
import os
import socket
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "torch-nerf_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

RAYS, N_COARSE, N_FINE = 4096, 64, 128
H = W = 800
NEAR, FAR = 2.0, 6.0
MLP_FLOP_PER_SAMPLE = 2 * 593408
FP32_MFMA_PEAK_TFLOPS = 157.3             # MI355X_MICROARCH.md: 256 CU x 2.4 GHz x 256 FLOP/clk/CU
BF16_MFMA_PEAK_TFLOPS = 2500.0            # MI355X_MICROARCH.md: dense bf16 MFMA (no 2:1 sparsity)
HBM_PEAK_GBS = 8000.0                     # MI355X_MICROARCH.md: HBM3E spec peak
# fused render pass, mean of the coarse and the fine launch: per ray o, d (24 B) + rgb (12 B) + draws and weights
# (coarse: u1 256 B in, weights 256 B out; fine: u1 256 + u2, u3 1024 + weights 256 in and 256 floored back, 768 out)
# + ONE pass over the 2.57 MB packed weight stream per launch
ALGORITHMIC_BYTES_PER_LAUNCH = (4096 * (36 + 512) + 4096 * (36 + 256 + 1024 + 512 + 768) + 2 * 2569216) // 2


def build_scene(device):
    import torch_nerf.src.network as network
    import torch_nerf.src.scene as scene
    import torch_nerf.src.renderer.cameras as cameras
    import torch_nerf.src.renderer.integrators.quadrature_integrator as integrators
    import torch_nerf.src.renderer.ray_samplers as ray_samplers
    from torch_nerf.src.renderer.volume_renderer import VolumeRenderer
    from torch_nerf.src.signal_encoder import PositionalEncoder
    from torch_nerf.amd import synth

    focal = float(synth.blender_focal(W))
    pose = torch.from_numpy(synth.pose_spherical(37.0, -30.0, 4.0))
    cam = cameras.PerspectiveCamera({"f_x": focal, "f_y": focal, "img_width": W, "img_height": H}, pose, NEAR, FAR)
    enc = {"coord_enc": PositionalEncoder(3, 10, True), "dir_enc": PositionalEncoder(3, 4, True)}
    nets, flats = [], []
    for seed in (3, 4):
        flat = synth.nerf_flat_params(seed=seed, sigma_bias=1.0, sigma_gain=30.0)
        net = network.NeRF(63, 27)
        net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.split_flat_params(flat).items()})
        nets.append(net.to(device))
        flats.append(flat)
    renderer = VolumeRenderer(integrators.QuadratureIntegrator(), ray_samplers.StratifiedSampler(), cam)
    return renderer, scene.PrimitiveCube(nets[0], enc), scene.PrimitiveCube(nets[1], enc), nets, flats, cam, focal, pose


def render_step(renderer, scene_c, scene_f, pix, device_index):
    """Exactly the two calls runners/train.py:172-201 / runner_utils.py:890-908 make per batch."""
    c_rgb, c_idx, c_w = renderer.render_scene(scene_c, RAYS, N_COARSE, False, device_index, pixel_indices=pix)
    f_rgb, _, f_w = renderer.render_scene(scene_f, RAYS, (N_COARSE, N_FINE), False, device_index,
                                          pixel_indices=c_idx, weights=c_w)
    return c_rgb, f_rgb


def cpu_model():
    """`model name` of the host CPU (what lscpu prints), for the cpu_baseline objects."""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_cores():
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(np.ceil(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    return cores


def train_leg(renderer, scene_c, scene_f, nets, pix, device, local_rank, steps, warmup, world):
    """Secondary figure: full training step (runners/train.py:120-218 without the .item() syncs):
    coarse + fine forward with activation record, MSE coarse + MSE fine, backward through the
    integrator and both MLPs (hand-written kernels), fused Adam (one launch over both networks) with
    the reference's ExponentialLR.  With world > 1 every rank trains on its own 4096 rays (weak scaling)
    and the optimizer all-reduces the joined 4.77 MB gradient blob once per step."""
    from torch_nerf.amd.optim import FusedAdam
    params = [p for net in nets for p in net.parameters()]
    opt = FusedAdam(params, lr=5e-4, eps=1e-8)                                  # configs/train_params/nerf.yaml
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, pow(0.00005 / 0.0005, 1 / 300000))
    mse = torch.nn.MSELoss()
    gt = torch.rand((RAYS, 3), device=device)

    def step(s):
        opt.zero_grad(set_to_none=True)
        c_rgb, c_idx, c_w = renderer.render_scene(scene_c, RAYS, N_COARSE, False, local_rank, pixel_indices=pix[s])
        f_rgb, _, _ = renderer.render_scene(scene_f, RAYS, (N_COARSE, N_FINE), False, local_rank,
                                            pixel_indices=c_idx, weights=c_w)
        loss = mse(gt, c_rgb) + mse(gt, f_rgb)
        loss.backward()
        opt.step()
        sched.step()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    from torch_nerf.amd import ops
    for s in range(warmup):
        step(s)
    fence()
    ops.KERNEL_EVENTS = []
    t0 = time.perf_counter()
    for s in range(warmup, warmup + steps):
        step(s % len(pix))
    fence()
    dt = time.perf_counter() - t0
    events, ops.KERNEL_EVENTS = ops.KERNEL_EVENTS, None
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    flop = RAYS * (N_COARSE + N_COARSE + N_FINE) * 2 * (593408 + 1151104)   # fwd + bwd, per rank
    # roofline of the step's MFMA kernels from HIP events on the launch stream: record-mode forward launches and
    # backward enqueues (dX chain + dW GEMMs + the two thin reduction kernels), against the fp32 MFMA peak
    fwd = [(M, a.elapsed_time(b)) for tag, M, a, b in events if tag == "mlp_forward"]
    bwd = [(M, a.elapsed_time(b)) for tag, M, a, b in events if tag == "mlp_backward"]
    fwd_ms, bwd_ms = sum(t for _, t in fwd), sum(t for _, t in bwd)
    fwd_tf = sum(M for M, _ in fwd) * 2 * 593408 / (fwd_ms * 1e-3) / 1e12
    bwd_tf = sum(M for M, _ in bwd) * 2 * 1151104 / (bwd_ms * 1e-3) / 1e12
    both = (sum(M for M, _ in fwd) * 2 * 593408 + sum(M for M, _ in bwd) * 2 * 1151104) / ((fwd_ms + bwd_ms) * 1e-3) / 1e12
    roofline = {"bound": "mfma", "achieved": round(both, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(both / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": None,
                "kernel": "mlp_forward_kernel<false,true> (record mode) + mlp_bwd_dx_kernel + mlp_bwd_dw_kernel",
                "forward_record": {"ms_per_step": round(fwd_ms / steps, 3), "TFLOPs": round(fwd_tf, 1),
                                   "frac": round(fwd_tf / FP32_MFMA_PEAK_TFLOPS, 4)},
                "backward": {"ms_per_step": round(bwd_ms / steps, 3), "TFLOPs": round(bwd_tf, 1),
                             "frac": round(bwd_tf / FP32_MFMA_PEAK_TFLOPS, 4)},
                "other_ms_per_step": round(dt / steps * 1e3 - (fwd_ms + bwd_ms) / steps, 3)}
    # round 6, opt-in variant (NeRF.f16x2_training): the same step with all three thirds on the f16 pipe -- the RECORDING
    # forward, the reverse chain (dX, one power-of-two scale per sample) and the dW GEMMs (one per gradient plane)
    split = None
    if world == 1:
        for net in nets:
            net.f16x2_training = True
        for s in range(2):
            step(s)
        fence()
        ops.KERNEL_EVENTS = []
        t0 = time.perf_counter()
        for s in range(warmup, warmup + steps):
            step(s % len(pix))
        fence()
        dtx = time.perf_counter() - t0
        ev, ops.KERNEL_EVENTS = ops.KERNEL_EVENTS, None
        for net in nets:
            net.f16x2_training = False
        fx = [a.elapsed_time(b) for tag, M, a, b in ev if tag == "mlp_forward"]
        bx = [a.elapsed_time(b) for tag, M, a, b in ev if tag == "mlp_backward"]
        # what binds this step is HBM, not the matrix pipe: per sample the record goes out once (10 400 B) and comes back once
        # (288 B of ReLU bits to the reverse chain, 10 112 B of activations to the dW GEMMs), the gradient planes likewise
        # (9 748 B out of the reverse chain, 9 748 B into the GEMMs), 32 B of sigma / rgb and their gradients: DESIGN.md 4.8
        x2_bytes = 10400 + 288 + 10112 + 2 * 9748 + 32
        x2_samples = sum(M for tag, M, a, b in ev if tag == "mlp_forward")
        x2_gbs = x2_samples * x2_bytes / ((sum(fx) + sum(bx)) * 1e-3) / 1e9
        split = {"ms_per_step": dtx / steps * 1e3, "rays_per_s": RAYS * steps / dtx, "speedup_vs_fp32_step": dt / dtx,
                 "forward_record_ms_per_step": round(sum(fx) / steps, 3), "backward_ms_per_step": round(sum(bx) / steps, 3),
                 "roofline": {"bound": "hbm", "achieved": round(x2_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(x2_gbs / HBM_PEAK_GBS, 4), "traffic": None,
                              "algorithmic_bytes_per_sample": x2_bytes,
                              "kernel": "mlp_forward_f16x2_kernel<2,1,true> + mlp_bwd_dx_f16x2_kernel + mlp_bwd_dw_x2_kernel "
                                        "(+ thin rows, reduction), HIP events around the forward and the backward of every launch"},
                 "what": "NeRF.f16x2_training: record forward, reverse chain (dX, one power-of-two scale per sample) and dW "
                         "GEMMs (one per gradient plane) on the split-f16 kernels; same record, gradient planes, partial "
                         "tiles and fixed-order reduction as the fp32 step, fp32 accumulation, fused Adam"}
    return {"rays_per_s": world * RAYS * steps / dt, "ms_per_step": dt / steps * 1e3, "steps": steps, "f16x2_training": split,
            "what": "fwd+bwd+fused Adam+ExponentialLR, both networks, 4096 rays x (64 + 192) samples per GPU"
                    + (f", gradient all-reduce over {world} ranks" if world > 1 else ""),
            "mfma_frac_of_peak": flop / (dt / steps) / 1e12 / FP32_MFMA_PEAK_TFLOPS, "roofline": roofline}


def bf16_leg(renderer, scene_c, scene_f, nets, pix, local_rank, steps, warmup):
    """Secondary figure (BASELINE configs[2]): the same render step with bf16 weights / layer inputs on the
    bf16 MFMA path, its PSNR against the fp32 step on identical pixels and draws, and the roofline object of
    its dominant kernel (HIP events around every launch in the timed region, as for the fp32 kernel)."""
    from torch_nerf.amd import ops

    def run(s, seed):
        torch.manual_seed(seed)
        return render_step(renderer, scene_c, scene_f, pix[s], local_rank)[1]

    with torch.no_grad():
        ref = run(0, 99)
        for net in nets:
            net.bf16_inference = True
        got = run(0, 99)
        mse = torch.mean((got.double() - ref.double()) ** 2).item()
        for s in range(warmup):
            run(s, s)
        torch.cuda.synchronize()
        # three rounds of `steps` steps, the median round reported: a 1 ms step is 12 launches, and one descheduling of
        # the host thread (CFS throttling on a shared box, seen once: 1.98 ms) would otherwise be the figure
        rounds = []
        for rnd in range(3):
            ops.KERNEL_EVENTS = []
            t0 = time.perf_counter()
            for s in range(warmup, warmup + steps):
                render_step(renderer, scene_c, scene_f, pix[s % len(pix)], local_rank)
            torch.cuda.synchronize()
            rounds.append((time.perf_counter() - t0, ops.KERNEL_EVENTS))
        ops.KERNEL_EVENTS = None
        dt, events = sorted(rounds, key=lambda r: r[0])[1]
        for net in nets:
            net.bf16_inference = False
    durs = [(M, e0.elapsed_time(e1)) for tag, M, e0, e1 in events if tag == "mlp_forward_bf16"]
    total_ms = sum(ms for _, ms in durs)
    achieved = sum(M for M, _ in durs) * MLP_FLOP_PER_SAMPLE / (total_ms * 1e-3) / 1e12
    fine = [ms for M, ms in durs if M == RAYS * (N_COARSE + N_FINE)]
    roofline = {"bound": "mfma", "achieved": round(achieved, 1), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / BF16_MFMA_PEAK_TFLOPS, 4), "traffic": None,
                "kernel": ops.DOMINANT_KERNEL_BF16 + ", 2 launches/step", "launches": len(durs),
                "ms_per_launch": round(total_ms / len(durs), 4),
                "fine_ms_per_launch": round(float(np.mean(fine)), 4) if fine else None}
    return {"rays_per_s": RAYS * steps / dt, "ms_per_step": dt / steps * 1e3, "steps": steps,
            "ms_per_step_rounds": [round(r[0] / steps * 1e3, 4) for r in rounds],
            "psnr_vs_fp32_db": round(10.0 * np.log10(1.0 / mse), 2) if mse > 0 else None,
            "max_abs_err_vs_fp32": (got - ref).abs().max().item(), "roofline": roofline,
            "what": "render step with bf16 weights + bf16 layer inputs (v_mfma_f32_32x32x16_bf16, fp32 accumulate)"}


def f16x2_leg(renderer, scene_c, scene_f, nets, pix, local_rank, steps, warmup):
    """Secondary figure (round 6): the same render step with the MLP on the f16 matrix pipe, every operand split in two
    f16 parts (three v_mfma_f32_16x16x32_f16 per k-step, fp32 accumulate): the fp32 BOUND -- 1e-5 abs, tests/
    test_gpu_f16x2.py -- at a multiple of the fp32 MFMA kernel's speed.  The headline stays the fp32-MFMA kernel:
    BASELINE configs[1] names fp32.  roofline: 3x the algorithmic MLP FLOPs (the three part products the kernel
    issues) against the dense f16 MFMA peak; `equivalent_fp32_TFLOPs` = the algorithmic FLOPs alone."""
    from torch_nerf.amd import ops

    def run(s, seed):
        torch.manual_seed(seed)
        return render_step(renderer, scene_c, scene_f, pix[s], local_rank)

    with torch.no_grad():
        ref_c, ref = run(0, 99)
        for net in nets:
            net.f16x2_inference = True
        got_c, got = run(0, 99)
        for s in range(warmup):
            run(s, s)
        torch.cuda.synchronize()
        rounds = []
        for rnd in range(3):
            ops.KERNEL_EVENTS = []
            t0 = time.perf_counter()
            for s in range(warmup, warmup + steps):
                render_step(renderer, scene_c, scene_f, pix[s % len(pix)], local_rank)
            torch.cuda.synchronize()
            rounds.append((time.perf_counter() - t0, ops.KERNEL_EVENTS))
        ops.KERNEL_EVENTS = None
        dt, events = sorted(rounds, key=lambda r: r[0])[1]
        for net in nets:
            net.f16x2_inference = False
    durs = [(M, e0.elapsed_time(e1)) for tag, M, e0, e1 in events if tag == "mlp_forward_f16x2"]
    total_ms = sum(ms for _, ms in durs)
    algorithmic = sum(M for M, _ in durs) * MLP_FLOP_PER_SAMPLE / (total_ms * 1e-3) / 1e12
    achieved = 3.0 * algorithmic
    fine = [ms for M, ms in durs if M == RAYS * (N_COARSE + N_FINE)]
    err = (got - ref).abs().max(dim=1).values
    roofline = {"bound": "mfma", "achieved": round(achieved, 1), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / BF16_MFMA_PEAK_TFLOPS, 4), "traffic": None,
                "kernel": ops.DOMINANT_KERNEL_F16X2 + ", 2 launches/step", "launches": len(durs),
                "ms_per_launch": round(total_ms / len(durs), 4),
                "fine_ms_per_launch": round(float(np.mean(fine)), 4) if fine else None,
                "equivalent_fp32_TFLOPs": round(algorithmic, 1),
                "vs_fp32_mfma_peak": round(algorithmic / FP32_MFMA_PEAK_TFLOPS, 3)}
    return {"rays_per_s": RAYS * steps / dt, "ms_per_step": dt / steps * 1e3, "steps": steps, "dtype": "f16x2",
            "ms_per_step_rounds": [round(r[0] / steps * 1e3, 4) for r in rounds],
            "coarse_max_abs_err_vs_fp32": (got_c - ref_c).abs().max().item(),
            "fine_median_abs_err_vs_fp32": err.median().item(),
            "fine_pixels_beyond_1e-5": int((err > 1e-5).sum().item()),
            "fine_max_abs_err_vs_fp32": err.max().item(), "roofline": roofline,
            "what": "render step with every MLP operand split in two f16 parts (lo.hi + hi.lo + hi.hi on "
                    "v_mfma_f32_16x16x32_f16, fp32 accumulate): the fp32 1e-5 bound on the f16 matrix pipe; fine pixels "
                    "beyond 1e-5 of the fp32 step are rays where a 1e-7 change of a coarse weight moved a fine sample "
                    "across a cdf boundary (the coarse pass has no such step: its bound is the kernel's)"}


def cpu_baseline(flats, focal, pose, device):
    """Eager-torch CPU port on the host cores; returns the JSON objects (all cores, 1 thread) + PSNR of HIP vs port."""
    from oracle import torch_port as TP
    from torch_nerf.amd import ops, shard, synth

    cores = host_cores()
    params = [{k: torch.from_numpy(v.copy()) for k, v in synth.split_flat_params(f).items()} for f in flats]

    def run(n_rays):
        pix = torch.from_numpy(synth.pixel_batch(0, H, W, n_rays))
        draws = tuple(d.cpu() for d in shard.ray_draws(7, 0, n_rays, N_COARSE, N_FINE, "cpu"))
        t0 = time.perf_counter()
        with torch.no_grad():
            out = TP.render_batch(params[0], params[1], pix, H, W, focal, pose, NEAR, FAR, N_COARSE, N_FINE, draws)
        return time.perf_counter() - t0, pix, draws, out

    # ---- the reference's own default: torch.set_num_threads(1) (runners/runner_utils.py:427); ~8 s of CPU work
    torch.set_num_threads(1)
    run(32)
    t_probe, *_ = run(64)
    n1 = int(min(1024, max(64, 64 * (8.0 / max(t_probe, 1e-3))))) // 64 * 64
    secs1, *_ = run(n1)
    one = {"value": n1 / secs1, "unit": "rays/s", "cores": 1, "kind": "port", "cpu_model": cpu_model(),
           "sample": f"{n1} of the 4096 rays of one batch, coarse 64 + fine 64+128, forward, eager PyTorch CPU port "
                     f"(oracle/torch_port.py), torch.set_num_threads(1) as runners/runner_utils.py:427 sets it, "
                     f"{secs1:.1f} s"}

    # ---- all host cores the cgroup grants
    torch.set_num_threads(cores)
    run(64)                                   # warm-up (thread pool, MKL)
    t_probe, *_ = run(256)
    n = int(min(RAYS, max(256, 256 * (12.0 / max(t_probe, 1e-3)))))   # aim at ~12 s of CPU work
    n = max(256, (n // 256) * 256)
    secs, pix, draws, (c_rgb, c_w, f_rgb, f_w, idx) = run(n)
    if secs < 8.0:                            # fast host: repeat the full batch to reach ~10 s
        reps = int(min(8, max(1, round(10.0 / secs) - 1)))
        extra = [run(n)[0] for _ in range(reps)]
        secs = (secs + sum(extra)) / (1 + reps)
    # the HIP path on the same rays and draws: PSNR / max error of the pixel colours
    k4 = (np.float32(focal), np.float32(focal), W / 2.0, H / 2.0)
    o, d = ops.generate_rays(H, W, k4, pose, False, focal, NEAR, device, pix=pix.to(device))
    t_bins = torch.linspace(NEAR, FAR, N_COARSE + 1, device=device)[:-1]
    ps = (FAR - NEAR) / N_COARSE
    u1c, u1, u2, u3 = (x.to(device) for x in draws)
    pc, pf = (ops.mlp_pack(torch.from_numpy(f).to(device)) for f in flats)
    g_c, g_w = ops.render_rays(pc, o, d, t_bins, ps, u1c)
    g_f, _ = ops.render_rays(pf, o, d, t_bins, ps, u1, weights=g_w, u2=u2, u3=u3)
    err = (g_f.cpu() - f_rgb).abs().max().item()
    mse = torch.mean((g_f.cpu().double() - f_rgb.double()) ** 2).item()
    psnr = float("inf") if mse == 0 else 10.0 * np.log10(1.0 / mse)
    base = {"value": n / secs, "unit": "rays/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
            "cpus_visible": os.cpu_count(),
            # how the port compares with the reference it stands in for (the reference cannot travel to this box):
            # scripts/port_vs_reference.py, build container, alternating passes -> profiles/r06_port_vs_reference.json
            "port_vs_reference": "1.00x (0.99 - 1.00) of the imported reference's rays/s on the build container: 1024 rays, "
                                 "8 threads, 7 alternating passes each, medians 727 vs 729 rays/s",
            "sample": f"{n} of the 4096 rays of one batch, coarse 64 + fine 64+128, forward, eager PyTorch CPU "
                      f"port of the reference path (oracle/torch_port.py), {cores} threads, {secs:.1f} s per pass"}
    quality = {"psnr_vs_cpu_port_db": (None if psnr == float("inf") else round(psnr, 2)),
               "max_abs_pixel_err_vs_cpu_port": err, "coarse_max_abs_err": (g_c.cpu() - c_rgb).abs().max().item()}
    return base, one, quality


def hbm_stages(device, reps=20):
    """The HBM-bound stages at the training batch (4096 rays: latency-bound, a lower bound on the streaming rate) and at
    frame scale (65 536 and 640 000 rays = a whole 800x800 frame in one launch: the streaming rate itself)."""
    out = hbm_stages_at(device, RAYS, reps)
    for n in (65536, H * W):
        out[f"rays_{n}"] = hbm_stages_at(device, n, max(5, reps // 2))
    return out


def hbm_stages_at(device, n_rays, reps=20):
    """the stages outside the MLP are HBM-bound and reported separately against 8 TB/s.
    Each kernel at `n_rays` rays (64 or 64+128 samples) straight through the C ABI on preallocated
    buffers: `reps` launches back to back behind a GPU-side sleep (so that the host is ahead and the events see
    kernel time + launch boundary only), HIP events around the run; bytes = the algorithmic traffic of the
    stage.  (Inference runs these stages inside the fused render kernel; training runs them
    as kernels.)"""
    import ctypes
    from torch_nerf.amd import _lib
    lib = _lib.load()
    n, Sc, Sf = n_rays, N_COARSE, N_FINE
    S = Sc + Sf
    g = torch.Generator(device=device).manual_seed(5)
    o = torch.randn((n, 3), device=device, generator=g)
    d = torch.randn((n, 3), device=device, generator=g)
    t_bins = torch.linspace(NEAR, FAR, Sc + 1, device=device)[:-1].contiguous()
    ps = (FAR - NEAR) / Sc
    u1, u2, u3 = (torch.rand((n, k), device=device, generator=g) for k in (Sc, Sf, Sf))
    w = torch.rand((n, Sc), device=device, generator=g)
    sigma = torch.rand((n, S), device=device, generator=g) * 3
    rad = torch.rand((n, S, 3), device=device, generator=g)
    delta = torch.full((n, S), 4.0 / S, device=device)
    g_rgb = torch.randn((n, 3), device=device, generator=g)
    pts, dirs = torch.empty((n, S, 3), device=device), torch.empty((n, S, 3), device=device)
    dl, wo, gs = torch.empty((n, S), device=device), torch.empty((n, S), device=device), torch.empty((n, S), device=device)
    rgb, gc = torch.empty((n, 3), device=device), torch.empty((n, S, 3), device=device)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    cases = {
        "stratified": (lambda: lib.nerf_sample_stratified(P(o), P(d), n, Sc, P(t_bins), ps, P(u1), None, P(pts), P(dirs),
                                                          P(dl), st), n * Sc * 32 + n * 24,
                       "u1 4 B in; pts 12 + dirs 12 + delta 4 B out per sample; o, d per ray"),
        "hierarchical": (lambda: lib.nerf_sample_hierarchical(P(o), P(d), n, Sc, Sf, P(t_bins), ps, P(w), P(u1), P(u2),
                                                              P(u3), None, None, P(pts), P(dirs), P(dl), st),
                         n * (Sc * 12 + Sf * 8 + S * 28 + 24),
                         "weights 4 B read + 4 B written back, u1 4 B per coarse sample; u2, u3 4 B per fine sample; "
                         "pts 12 + dirs 12 + delta 4 B out per sorted sample; o, d per ray"),
        "composite_fwd": (lambda: lib.nerf_composite_forward(P(sigma), P(rad), P(delta), n, S, P(rgb), P(wo), st),
                          n * S * 24 + n * 12,
                          "sigma 4 + radiance 12 + delta 4 B in, weights 4 B out per sample; rgb 12 B per ray"),
        "composite_bwd": (lambda: lib.nerf_composite_backward(P(sigma), P(rad), P(delta), P(g_rgb), None, n, S, P(gs),
                                                              P(gc), st), n * S * 36 + n * 12,
                          "sigma 4 + radiance 12 + delta 4 B in, g_sigma 4 + g_radiance 12 B out per sample; "
                          "g_rgb 12 B per ray"),
    }
    out = {}
    for name, (fn, nbytes, what) in cases.items():
        for _ in range(3):
            assert fn() == 0, name
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(4_000_000)          # ~2 ms of GPU time: the launches below queue up behind it
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / reps
        gbs = nbytes / (t * 1e-3) / 1e9
        out[name] = {"us": round(t * 1e3, 2), "bytes": nbytes, "achieved_GBs": round(gbs, 1),
                     "frac_of_8TBs": round(gbs / HBM_PEAK_GBS, 4), "traffic": what}
    out["note"] = (f"{n} rays x {Sc} (stratified) / {Sc}+{Sf} (others) samples, mean of {reps} back-to-back launches "
                   "(kernel + launch boundary)" + ("; one-wave-per-ray kernels of 5-45 us are latency-bound at this batch "
                   "size, so the fraction is a lower bound on their streaming rate (rays_65536 / rays_640000 hold the "
                   "frame-scale figures)" if n <= 4096 else ""))
    return out


def frame_leg(nets, cam, rank, world, device, dist_on=None):
    """BASELINE configs[4]: one 800x800 frame (640 000 rays, 64+128), pixel ranges sharded over the ranks, each rank
    renders coarse+fine for its range and ONE all-gather assembles the image (shard.render_frame).  Strong scaling.
    With world > 1 EVERY rank first renders the whole frame alone (no collective, nobody idles inside the process
    group's watchdog window); the gathered image must then equal that one-rank image bit for bit on every rank."""
    from torch_nerf.amd import shard
    dist_on = world > 1 if dist_on is None else dist_on

    def fence():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize()

    solo = None
    if dist_on:     # doubles as the warm-up; draws are a function of the global ray index, so the bits must agree
        solo = shard.render_frame(cam, nets[0], nets[1], N_COARSE, N_FINE, False, seed=1, single_rank=True)
        torch.cuda.synchronize()
    shard.render_frame(cam, nets[0], nets[1], N_COARSE, N_FINE, False, seed=1)      # warm-up (collective included)
    fence()
    stats = {}
    t0 = time.perf_counter()
    img = shard.render_frame(cam, nets[0], nets[1], N_COARSE, N_FINE, False, seed=1, stats=stats)
    fence()
    dt = time.perf_counter() - t0
    mine = [stats["render_events"][0].elapsed_time(stats["render_events"][1]), float(stats["launches"]), float(stats["rays"])]
    per_rank = [mine]
    if dist_on:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
        every = torch.empty((world, 3), device=device, dtype=torch.float64)      # each rank's own rendering, gather excluded
        dist.all_gather_into_tensor(every, torch.tensor([mine], device=device, dtype=torch.float64))
        per_rank = every.tolist()
    digest = hashlib.sha256(img.cpu().numpy().tobytes()).hexdigest()[:16]
    gather_ms = None
    if dist_on:    # the frame's ONE collective on its own: the same (H*W/world, 3) slabs, rendering excluded
        per = (H * W + world - 1) // world
        slab = torch.zeros((per, 3), device=device)
        full = torch.empty((world * per, 3), device=device)
        times = []
        for _ in range(4):
            fence()
            t0 = time.perf_counter()
            dist.all_gather_into_tensor(full, slab)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        t = torch.tensor([min(times[1:])], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        gather_ms = t.item() * 1e3
    out = {"ms": dt * 1e3, "rays_per_s": H * W / dt, "rays": H * W, "n_gpus": world, "scaling": "strong",
           "gather_ms": gather_ms,
           "ms_per_rank": [round(r[0], 3) for r in per_rank],       # GPU time of each rank's own launches (no collective)
           "launches_per_rank": [int(r[1]) for r in per_rank], "rays_per_rank": [int(r[2]) for r in per_rank],
           "image_sha256_16": digest,
           "what": f"{W}x{H} frame, 64+128 samples, fp32, contiguous pixel ranges over {world} rank(s), "
                   "all-gather of the (H*W/world, 3) slabs included"}
    if dist_on:
        out["image_sha256_16_one_rank"] = hashlib.sha256(solo.cpu().numpy().tobytes()).hexdigest()[:16]
        same = torch.tensor([float(torch.equal(solo, img))], device=device)
        dist.all_reduce(same, op=dist.ReduceOp.MIN)          # equal on EVERY rank
        out["equals_one_rank_image"] = bool(same.item())
    return out


def frame_api_leg(renderer, scene_c, scene_f, cam_pose, focal):
    """The whole 800x800 frame through the CLASS API, call for call as the reference's runners do it:
    frame_api = _visualize_scene (runners/runner_utils.py:872-908): render_scene(default_scene, num_pixels=H*W,
                num_samples=64, num_ray_batch=H*W // 4096) -> render_scene(fine_scene, ..., (64, 128), pixel_indices,
                weights, num_ray_batch) -> reshape / permute to (C, H, W);
    validate  = the per-view body of validate_one_epoch (runners/train.py:285-330): a fresh PerspectiveCamera, the same
                two calls, MSE / PSNR against the ground-truth view.
    Peak device memory of the frame is reported (the API materialises the whole-frame draws and the (N, S) weights)."""
    import torch_nerf.src.renderer.cameras as cameras
    total = H * W
    dev_i = torch.cuda.current_device()

    def frame():
        pred, idx, w = renderer.render_scene(scene_c, num_pixels=total, num_samples=N_COARSE, project_to_ndc=False,
                                             device=dev_i, num_ray_batch=total // RAYS)
        pred, _, _ = renderer.render_scene(scene_f, num_pixels=total, num_samples=(N_COARSE, N_FINE),
                                           project_to_ndc=False, pixel_indices=idx, weights=w, device=dev_i,
                                           num_ray_batch=total // RAYS)
        return pred.reshape(H, W, -1).permute(2, 0, 1)

    out = {}
    with torch.no_grad():
        frame()
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        t0 = time.perf_counter()
        img = frame()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out["frame_api"] = {"ms": dt * 1e3, "rays_per_s": total / dt, "rays": total,
                            "peak_mem_gib": (torch.cuda.max_memory_allocated() - base) / 2 ** 30,
                            "finite": bool(torch.isfinite(img).all()),
                            "what": "runner_utils.py:872-908: two render_scene calls over the whole frame "
                                    "(num_ray_batch = H*W // 4096), fp32, 1 GPU"}
        gt = torch.rand((H, W, 3))
        loss_func = torch.nn.MSELoss()
        t0 = time.perf_counter()
        renderer.camera = cameras.PerspectiveCamera({"f_x": focal, "f_y": focal, "img_width": W, "img_height": H},
                                                    cam_pose, NEAR, FAR)
        pred = frame()[None]
        pixel_gt = gt.reshape(H, W, -1).permute(2, 0, 1)[None].cuda()
        mse = loss_func(pixel_gt, pred).item()
        psnr = -10.0 * np.log10(max(mse, 1e-12))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out["validate"] = {"ms": dt * 1e3, "rays_per_s": total / dt, "psnr_vs_random_gt_db": float(psnr),
                           "what": "train.py:285-330 per view: camera + two whole-frame render_scene calls + MSE / PSNR"}
    return out


def net_variants_leg(device):
    """The rest of the reference's constructor space NeRF(pos_dim, view_dir_dim, feat_dim) (network/nerf.py:24-63), driver-run
    (VERDICT r03 item 1): the layered family on a narrow (register-resident kernels) and a wide (plane-parked kernel)
    network, forward / record forward / backward as fractions of the fp32 MFMA peak on the networks' own algorithmic
    FLOPs, and the default network's call whose INPUTS require grad (record forward + input-gradient dX chain + dW)."""
    from torch_nerf.amd import ops, synth
    M = RAYS * (N_COARSE + N_FINE)          # 786 432: the sample count of the headline's fine pass

    def t(fn, window_ms=120.0):
        """mean duration of fn over ~window_ms of back-to-back calls (a 0.8 ms call timed three times measures the
        clock ramp of an idle chip, not the kernels: 10 % low)"""
        def run(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n
        fn()
        torch.cuda.synchronize()
        once = run(2)
        return run(max(3, min(200, int(window_ms / max(once, 1e-3)))))

    out = {"samples": M, "peak_TFLOPs": FP32_MFMA_PEAK_TFLOPS}
    gs, gc = torch.randn(M, device=device), torch.randn(M, 3, device=device)
    # register-resident kernels: feat 64 (2 x 2 blocks per wavefront), 128 (2 x 4), 256 with three position blocks
    # (coord_encode_level 12: 1 x 8, general reverse chain); plane-parked general kernel: feat 512
    for e_p, e_d, F in ((63, 27, 64), (63, 27, 128), (75, 27, 256), (63, 27, 512)):
        H2 = F // 2
        mac = e_p * F + 4 * F * F + (F + e_p) * F + 2 * F * F + F * (F + 1) + (F + e_d) * H2 + 3 * H2
        net = ops.Net.dims_only(e_p, e_d, F)
        flat = torch.from_numpy(synth.nerf_flat_params(seed=1, pos_dim=e_p, view_dir_dim=e_d, feat_dim=F)).to(device)
        pe, de = torch.randn(M, e_p, device=device), torch.randn(M, e_d, device=device)
        fwd = t(lambda: ops.mlp_layered_forward(flat, pe, de, net))
        sigma, rgb, rec = ops.mlp_layered_forward(flat, pe, de, net, record=True)
        rfwd = t(lambda: ops.mlp_layered_forward(flat, pe, de, net, record=True))
        bwd = t(lambda: ops.mlp_layered_backward(flat, pe, de, net, sigma, rgb, rec, gs, gc))
        frac = lambda flop, ms: round(flop * M / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4)
        out[f"nerf_{e_p}_{e_d}_{F}"] = {"forward_ms": fwd, "forward_frac": frac(2 * mac, fwd), "record_forward_ms": rfwd,
                                        "record_forward_frac": frac(2 * mac, rfwd), "backward_ms": bwd,
                                        "backward_frac": frac(4 * mac, bwd), "flop_per_sample_forward": 2 * mac}
        del sigma, rgb, rec
    # default network, inputs require grad: gradients w.r.t. the encoded inputs out of the fused dX chain
    e_p, e_d, F, H2 = 63, 27, 256, 128
    mac = e_p * F + 4 * F * F + (F + e_p) * F + 2 * F * F + F * (F + 1) + (F + e_d) * H2 + 3 * H2
    ig = 2 * e_p * F + e_d * H2
    flat = torch.from_numpy(synth.nerf_flat_params(seed=1)).to(device)
    packed = ops.mlp_pack(flat)
    pe, de = torch.randn(M, e_p, device=device), torch.randn(M, e_d, device=device)
    sigma, rgb, saved = ops.mlp_forward(packed, pe, de, True, save=True)
    fwd = t(lambda: ops.mlp_forward(packed, pe, de, True, save=True))
    bwd = t(lambda: ops.mlp_backward(packed, flat, pe, de, True, sigma, rgb, saved, gs, gc, want_pos=True, want_dir=True))
    out["input_gradient_call_63_27_256"] = {
        "record_forward_ms": fwd, "backward_with_input_grads_ms": bwd,
        "frac": round((6 * mac + 2 * ig) * M / ((fwd + bwd) * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
        "what": "NeRF.forward on pre-encoded inputs that require grad: record forward + dX chain with g_pos / g_view_dir + dW"}
    return out


def runner_loop_leg(device, local_rank, steps, warmup):
    """The per-batch body of the reference's train_one_epoch (runners/train.py:120-218), statement by statement,
    against the drop-in classes -- what an UNMODIFIED runner pays per step: a fresh PerspectiveCamera per batch,
    np.random.choice(H*W, 4096, replace=False) inside render_scene, the CPU ground-truth gather + .cuda() upload,
    three .item() syncs, torch.optim.Adam + ExponentialLR.  Host pieces are timed one by one (perf_counter around
    the statement; the .item() figures are mostly the GPU work they wait for)."""
    import torch_nerf.src.renderer.cameras as cameras
    renderer, scene_c, scene_f, nets, _, _, focal, _ = build_scene(device)
    from torch_nerf.amd import synth
    # The runners start with torch.set_num_threads(1) (runner_utils.py:427, _init_torch).  It matters: left at its
    # default the CPU pool has one thread per visible core (128 on the GPU box, whose cgroup grants 16 CPUs); the
    # loop's tiny CPU ops (the ground-truth gather) then wake 128 spinning threads, the cgroup runs out of CFS quota
    # and every thread of the process is frozen until the next 100 ms period -- 30-70 ms stalls with an idle GPU in
    # every second step (measured: scripts/runner_loop_timeline.py, nr_throttled in /sys/fs/cgroup/cpu.stat).
    threads_before = torch.get_num_threads()
    torch.set_num_threads(1)
    params = [p for net in nets for p in net.parameters()]
    optimizer = torch.optim.Adam(params, lr=5e-4, eps=1e-8)                      # runner_utils.py:691-695
    scheduler = torch.optim.lr_scheduler.ExponentialLR(optimizer, pow(0.00005 / 0.0005, 1 / 300000))
    loss_func = torch.nn.MSELoss()
    views = [(torch.rand((H, W, 3)), torch.from_numpy(synth.pose_spherical(a, -30.0, 4.0))) for a in (10.0, 130.0, 250.0)]
    np.random.seed(0)
    clock = {k: 0.0 for k in ("camera", "render_coarse_call", "gt_gather_upload", "item_syncs", "render_fine_call",
                              "backward_call", "adam_call")}

    def tick(key, t0):
        clock[key] += time.perf_counter() - t0

    def step(k, timed):
        pixel_gt, extrinsic = views[k % len(views)]
        pixel_gt = pixel_gt.squeeze().reshape(-1, 3)
        loss = 0.0
        optimizer.zero_grad()
        t0 = time.perf_counter()
        renderer.camera = cameras.PerspectiveCamera({"f_x": focal, "f_y": focal, "img_width": W, "img_height": H},
                                                    extrinsic, NEAR, FAR)
        if timed: tick("camera", t0)
        t0 = time.perf_counter()
        coarse_pred, coarse_indices, coarse_weights = renderer.render_scene(
            scene_c, num_pixels=RAYS, num_samples=N_COARSE, project_to_ndc=False, pixel_indices=None,
            device=torch.cuda.current_device())
        if timed: tick("render_coarse_call", t0)
        t0 = time.perf_counter()
        gt_c = pixel_gt[coarse_indices, ...].cuda()
        if timed: tick("gt_gather_upload", t0)
        coarse_loss = loss_func(gt_c, coarse_pred)
        loss += coarse_loss
        t0 = time.perf_counter()
        coarse_loss.item()
        if timed: tick("item_syncs", t0)
        t0 = time.perf_counter()
        fine_pred, fine_indices, _ = renderer.render_scene(
            scene_f, num_pixels=RAYS, num_samples=(N_COARSE, N_FINE), project_to_ndc=False,
            pixel_indices=coarse_indices, weights=coarse_weights, device=torch.cuda.current_device())
        if timed: tick("render_fine_call", t0)
        t0 = time.perf_counter()
        gt_f = pixel_gt[fine_indices, ...].cuda()
        if timed: tick("gt_gather_upload", t0)
        fine_loss = loss_func(gt_f, fine_pred)
        loss += fine_loss
        t0 = time.perf_counter()
        fine_loss.item()
        loss.item()
        if timed: tick("item_syncs", t0)
        t0 = time.perf_counter()
        loss.backward()
        if timed: tick("backward_call", t0)
        t0 = time.perf_counter()
        optimizer.step()
        scheduler.step()
        if timed: tick("adam_call", t0)

    for k in range(warmup):
        step(k, False)
    torch.cuda.synchronize()
    t_choice0 = time.perf_counter()
    for _ in range(5):
        np.random.choice(H * W, size=[RAYS], replace=False)
    choice_ms = (time.perf_counter() - t_choice0) / 5 * 1e3
    t0 = time.perf_counter()
    for k in range(warmup, warmup + steps):
        step(k, True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # the same unmodified loop with NERF_AMD_F16X2_TRAINING=1's effect: every network's training step on the split-f16 kernels
    for net in nets:
        net.f16x2_training = True
    for k in range(2):
        step(k, False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(warmup, warmup + steps):
        step(k, False)
    torch.cuda.synchronize()
    dtx = time.perf_counter() - t0
    for net in nets:
        net.f16x2_training = False
    torch.set_num_threads(threads_before)
    return {"ms_per_step": dt / steps * 1e3, "rays_per_s": RAYS * steps / dt, "steps": steps,
            "torch_num_threads": 1,
            "f16x2_training": {"ms_per_step": dtx / steps * 1e3, "rays_per_s": RAYS * steps / dtx,
                               "what": "the same loop with NERF_AMD_F16X2_TRAINING=1 (NeRF.f16x2_training on both networks)"},
            "host_ms_per_step": {k: round(v / steps * 1e3, 3) for k, v in clock.items()},
            "np_random_choice_ms": round(choice_ms, 3),
            "what": "runners/train.py:120-218 verbatim against the drop-in classes: camera per batch, np.random.choice "
                    "pixel batch, CPU ground-truth gather + upload, three .item() syncs, torch.optim.Adam + "
                    "ExponentialLR; compare with train.ms_per_step (device-resident batch, FusedAdam, no syncs)"}


def configs_leg(nets, flats, device):
    """Driver-run numbers for the remaining BASELINE configs.
    llff      (configs[3]): 1008x756 forward-facing frame, NDC rays, t in [0,1], 64+128, through shard.render_frame.
    coarse400 (configs[0]): 400x400 Blender frame, coarse-only 64 samples: the whole frame on the GPU, and the CPU
                            port (oracle/torch_port.py) on a bounded run of the SAME rays and draws beside it."""
    import torch_nerf.src.renderer.cameras as cameras
    from oracle import torch_port as TP
    from torch_nerf.amd import shard, synth
    out = {}
    # ---- configs[3]
    Hl, Wl, fl = 756, 1008, 815.0
    cam = cameras.PerspectiveCamera({"f_x": fl, "f_y": fl, "img_width": Wl, "img_height": Hl},
                                    torch.from_numpy(synth.llff_like_pose()), 0.0, 1.0)
    shard.render_frame(cam, nets[0], nets[1], N_COARSE, N_FINE, True, seed=2, single_rank=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    img = shard.render_frame(cam, nets[0], nets[1], N_COARSE, N_FINE, True, seed=2, single_rank=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # the same LLFF frame on the split-f16 kernel (round 6)
    shard.render_frame(cam, nets[0], nets[1], N_COARSE, N_FINE, True, seed=2, single_rank=True, f16x2=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    imgl = shard.render_frame(cam, nets[0], nets[1], N_COARSE, N_FINE, True, seed=2, single_rank=True, f16x2=True)
    torch.cuda.synchronize()
    dtl = time.perf_counter() - t0
    errl = (imgl - img).abs().max(dim=1).values
    out["llff_f16x2"] = {"ms": dtl * 1e3, "rays_per_s": Hl * Wl / dtl, "rays": Hl * Wl, "dtype": "f16x2",
                         "median_abs_err_vs_fp32_frame": float(errl.median().item()),
                         "pixels_beyond_1e-5_of_fp32_frame": int((errl > 1e-5).sum().item()),
                         "max_abs_err_vs_fp32_frame": float(errl.max().item()),
                         "what": "the llff frame (configs[3]: 1008x756, NDC rays) with the MLP on the split-f16 kernel"}
    # ---- configs[2]: "Blender ship, 800x800, 64+128, bf16 MLP weights on MFMA": the whole frame on the bf16 path and
    # its PSNR against the fp32 frame of the same networks, pose and draws
    cam8 = cameras.PerspectiveCamera({"f_x": float(synth.blender_focal(W)), "f_y": float(synth.blender_focal(W)),
                                      "img_width": W, "img_height": H},
                                     torch.from_numpy(synth.pose_spherical(37.0, -30.0, 4.0)), NEAR, FAR)
    ref32 = shard.render_frame(cam8, nets[0], nets[1], N_COARSE, N_FINE, False, seed=4, single_rank=True)
    shard.render_frame(cam8, nets[0], nets[1], N_COARSE, N_FINE, False, seed=4, single_rank=True, bf16=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    img16 = shard.render_frame(cam8, nets[0], nets[1], N_COARSE, N_FINE, False, seed=4, single_rank=True, bf16=True)
    torch.cuda.synchronize()
    dt16 = time.perf_counter() - t0
    mse16 = torch.mean((img16.double() - ref32.double()) ** 2).item()
    out["ship_bf16"] = {"ms": dt16 * 1e3, "rays_per_s": H * W / dt16, "rays": H * W,
                        "psnr_vs_fp32_frame_db": float(-10.0 * np.log10(max(mse16, 1e-20))),
                        "max_abs_err_vs_fp32_frame": float((img16 - ref32).abs().max().item()),
                        "what": "Blender geometry 800x800, 64+128, bf16 weights + layer inputs on v_mfma_f32_32x32x16_bf16 "
                                "(fp32 accumulate), whole frame on 1 GPU through shard.render_frame(bf16=True)"}
    # ---- the same frame on the split-f16 kernel (round 6): the fp32 bound on the f16 matrix pipe
    shard.render_frame(cam8, nets[0], nets[1], N_COARSE, N_FINE, False, seed=4, single_rank=True, f16x2=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    imgx = shard.render_frame(cam8, nets[0], nets[1], N_COARSE, N_FINE, False, seed=4, single_rank=True, f16x2=True)
    torch.cuda.synchronize()
    dtx = time.perf_counter() - t0
    errx = (imgx - ref32).abs().max(dim=1).values
    out["lego_f16x2"] = {"ms": dtx * 1e3, "rays_per_s": H * W / dtx, "rays": H * W, "dtype": "f16x2",
                         "median_abs_err_vs_fp32_frame": float(errx.median().item()),
                         "pixels_beyond_1e-5_of_fp32_frame": int((errx > 1e-5).sum().item()),
                         "max_abs_err_vs_fp32_frame": float(errx.max().item()),
                         "psnr_vs_fp32_frame_db": float(-10.0 * np.log10(max(torch.mean((imgx.double() - ref32.double()) ** 2).item(), 1e-20))),
                         "what": "Blender geometry 800x800, 64+128, every MLP operand split in two f16 parts (three "
                                 "v_mfma_f32_16x16x32_f16 per k-step, fp32 accumulate), whole frame on 1 GPU through "
                                 "shard.render_frame(f16x2=True); pixels beyond 1e-5 = rays where a fine sample changed its cdf bin"}
    out["llff"] = {"ms": dt * 1e3, "rays_per_s": Hl * Wl / dt, "rays": Hl * Wl, "finite": bool(torch.isfinite(img).all()),
                   "image_sha256_16": hashlib.sha256(img.cpu().numpy().tobytes()).hexdigest()[:16],
                   "what": "LLFF fern geometry 1008x756, NDC rays, t in [0,1], 64+128, fp32, 1 GPU, full frame"}
    # ---- configs[0]
    Hc = Wc = 400
    fc = float(synth.blender_focal(Wc))
    pose = torch.from_numpy(synth.pose_spherical(37.0, -30.0, 4.0))
    cam = cameras.PerspectiveCamera({"f_x": fc, "f_y": fc, "img_width": Wc, "img_height": Hc}, pose, NEAR, FAR)
    shard.render_frame(cam, nets[0], nets[1], N_COARSE, 0, False, seed=3, single_rank=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    img = shard.render_frame(cam, nets[0], nets[1], N_COARSE, 0, False, seed=3, single_rank=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    cores = host_cores()
    torch.set_num_threads(cores)
    p = {k: torch.from_numpy(v.copy()) for k, v in synth.split_flat_params(flats[0]).items()}

    def cpu(n_rays):
        pix = torch.arange(n_rays)
        u1c = shard.ray_draws(3, 0, n_rays, N_COARSE, 0, "cpu")[0]
        t0 = time.perf_counter()
        with torch.no_grad():
            o, d = TP.rays(pix, Hc, Wc, fc, pose)
            rgb, _, _ = TP.render_pass(p, o, d, NEAR, FAR, N_COARSE, u1c)
        return time.perf_counter() - t0, rgb

    cpu(256)
    t_probe, _ = cpu(1024)
    n = int(min(Hc * Wc, max(1024, 1024 * (10.0 / max(t_probe, 1e-3))))) // 1024 * 1024    # ~10 s of CPU work
    secs, rgb_cpu = cpu(n)
    err = (img[:n].cpu() - rgb_cpu).abs().max().item()
    out["coarse400"] = {"ms": dt * 1e3, "rays_per_s": Hc * Wc / dt, "rays": Hc * Wc,
                        "cpu_port": {"value": n / secs, "unit": "rays/s", "cores": cores, "kind": "port",
                                     "sample": f"the first {n} of the frame's 160000 rays, same draws, {secs:.1f} s"},
                        "max_abs_pixel_err_vs_cpu_port": err, "speedup_vs_cpu_port": (Hc * Wc / dt) / (n / secs),
                        "what": "Blender lego geometry 400x400, coarse-only 64 samples, fp32, whole frame on 1 GPU; "
                                "the reference's own CPU-runnable case restated by oracle/torch_port.py beside it"}
    return out


def encoder_variants_leg(device, local_rank, steps, warmup, renderer=None):
    """Whole-pass rays/s -- sampling + encoder kernel + network kernel + integral, coarse then fine, through
    VolumeRenderer.render_scene exactly like the headline step -- for the encoder settings the reference's yaml can
    name beyond the shipped 10 / 4 (configs/signal_encoder/positional_encoding.yaml:2-3, sh.yaml; runner_utils.py:584-612
    builds NeRF(coord_enc.out_dim, dir_enc.out_dim) behind them).  None of them fits the single-kernel render pass
    (pos_dim > 64, view_dir_dim > 32 or not a PositionalEncoder): each pass is the kernel chain.  `mlp_frac` = the
    algorithmic MLP FLOPs of the step (UNPADDED widths) over the whole step time, of the fp32 MFMA peak; `train` = the
    training step of the `train` leg on the same scenes (fwd + bwd + FusedAdam); `frame` = the 800x800 frame through
    shard.render_frame's chain fallback."""
    import torch_nerf.src.network as network
    import torch_nerf.src.scene as scene
    from torch_nerf.src.signal_encoder import PositionalEncoder, SHEncoder
    from torch_nerf.amd import shard, synth
    if renderer is None:
        renderer = build_scene(device)[0]
    cam = renderer.camera
    # the legs behind this one (bf16, f16x2, train) must draw what they would draw without it: the global generators'
    # states are restored on the way out
    rng_cpu, rng_gpu = torch.get_rng_state(), torch.cuda.get_rng_state(device)
    pix = [((torch.arange(RAYS, device=device) + s * RAYS) % (H * W)) for s in range(warmup + steps)]
    variants = {"coord_l12": (PositionalEncoder(3, 12, True), PositionalEncoder(3, 4, True)),
                "dir_l5": (PositionalEncoder(3, 10, True), PositionalEncoder(3, 5, True)),
                "sh": (SHEncoder(3, 4), SHEncoder(3, 4))}
    out = {}
    for tag, (ce, de) in variants.items():
        scenes = []
        for seed in (3, 4):
            flat = synth.nerf_flat_params(seed=seed, pos_dim=ce.out_dim, view_dir_dim=de.out_dim, sigma_bias=1.0, sigma_gain=30.0)
            net = network.NeRF(ce.out_dim, de.out_dim)
            net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in
                                 synth.split_flat_params(flat, ce.out_dim, de.out_dim, 256).items()})
            scenes.append(scene.PrimitiveCube(net.to(device), {"coord_enc": ce, "dir_enc": de}))
        torch.manual_seed(99)
        with torch.no_grad():
            for s in range(warmup):
                render_step(renderer, scenes[0], scenes[1], pix[s], local_rank)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for s in range(warmup, warmup + steps):
                _, f_rgb = render_step(renderer, scenes[0], scenes[1], pix[s], local_rank)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            shard.render_frame(cam, scenes[0], scenes[1], N_COARSE, N_FINE, False, seed=1, single_rank=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            img = shard.render_frame(cam, scenes[0], scenes[1], N_COARSE, N_FINE, False, seed=1, single_rank=True)
            torch.cuda.synchronize()
            dt_frame = time.perf_counter() - t0
            # round 6: the same step and frame with the networks' no-grad queries on the split-f16 kernel (PositionalEncoder
            # scenes; the SH scene has no raw-point entry)
            split = None
            if scenes[0].raw_net() is not None and scenes[0].raw_net().f16x2_ok:
                for sc in scenes:
                    sc.radiance_field.f16x2_inference = True
                torch.manual_seed(99)
                _, x_rgb0 = render_step(renderer, scenes[0], scenes[1], pix[warmup], local_rank)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for s in range(warmup, warmup + steps):
                    render_step(renderer, scenes[0], scenes[1], pix[s], local_rank)
                torch.cuda.synchronize()
                dtx = (time.perf_counter() - t0) / steps
                for sc in scenes:
                    sc.radiance_field.f16x2_inference = False
                torch.manual_seed(99)
                _, f_rgb0 = render_step(renderer, scenes[0], scenes[1], pix[warmup], local_rank)
                t0 = time.perf_counter()
                imgx = shard.render_frame(cam, scenes[0], scenes[1], N_COARSE, N_FINE, False, seed=1, single_rank=True, f16x2=True)
                torch.cuda.synchronize()
                dtx_frame = time.perf_counter() - t0
                ex = (imgx - img).abs().max(dim=1).values
                split = {"ms_per_step": dtx * 1e3, "rays_per_s": RAYS / dtx, "speedup_vs_fp32_step": dt / dtx,
                         "step_max_abs_err_vs_fp32": float((x_rgb0 - f_rgb0).abs().max().item()),
                         "frame_ms": dtx_frame * 1e3, "frame_pixels_beyond_1e-5": int((ex > 1e-5).sum().item()),
                         "frame_median_abs_err_vs_fp32": float(ex.median().item())}
        macs = sum(o * i for o, i in synth.layer_shapes(ce.out_dim, de.out_dim, 256))
        flop = 2 * macs * RAYS * (2 * N_COARSE + N_FINE)
        # the training step on the same scenes (record forward, integrator + MLP backward, FusedAdam), as the `train` leg
        from torch_nerf.amd.optim import FusedAdam
        opt = FusedAdam([p for sc in scenes for p in sc.radiance_field.parameters()], lr=5e-4, eps=1e-8)
        mse, gt = torch.nn.MSELoss(), torch.rand((RAYS, 3), device=device)

        def train_step(s):
            opt.zero_grad(set_to_none=True)
            c_rgb, c_idx, c_w = renderer.render_scene(scenes[0], RAYS, N_COARSE, False, local_rank, pixel_indices=pix[s])
            f_rgb, _, _ = renderer.render_scene(scenes[1], RAYS, (N_COARSE, N_FINE), False, local_rank,
                                                pixel_indices=c_idx, weights=c_w)
            (mse(gt, c_rgb) + mse(gt, f_rgb)).backward()
            opt.step()

        for s in range(2):
            train_step(s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(2, 2 + steps):
            train_step(s % len(pix))
        torch.cuda.synchronize()
        dt_train = (time.perf_counter() - t0) / steps
        # backward MACs: dW of every layer + dX of every layer input that is an activation (not the encodings)
        bwd_macs = 2 * macs - (2 * ce.out_dim * 256 + de.out_dim * 128)
        out[tag] = {"network": f"NeRF({ce.out_dim}, {de.out_dim}, 256)", "path": "fused family, pre-encoded entry" if scenes[0].radiance_field._net.fused else "layered family",
                    "ms_per_step": dt * 1e3, "rays_per_s": RAYS / dt, "mlp_frac": flop / dt / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                    "finite": bool(torch.isfinite(f_rgb).all()),
                    "train": {"ms_per_step": dt_train * 1e3, "rays_per_s": RAYS / dt_train,
                              "mlp_frac": 2 * (macs + bwd_macs) * RAYS * (2 * N_COARSE + N_FINE) / dt_train / 1e12 / FP32_MFMA_PEAK_TFLOPS},
                    "frame": {"ms": dt_frame * 1e3, "rays_per_s": H * W / dt_frame, "finite": bool(torch.isfinite(img).all())},
                    "f16x2": split}
    out["what"] = ("4096 rays x (64+128), fp32, coarse + fine render_scene per step (the headline step's two calls) behind "
                   "non-default encoders: kernel chain sampling -> encode -> network -> integral; frame = 800x800 via "
                   "shard.render_frame(scenes)")
    torch.set_rng_state(rng_cpu)
    torch.cuda.set_rng_state(rng_gpu, device)
    return out


def traffic_leg():
    """HBM traffic of the step's kernels, measured in THIS run: two child processes re-run a
    short bench (render + bf16 + train legs) under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate
    passes, MI355X_MICROARCH.md: the two do not fit one pass), the rocpd databases are read back and summed per
    dispatch.  Values are RAW counter bytes (KiB x 1024).  The guide's gfx950 note applies: FETCH_SIZE tallies a
    wide (16 B/lane) streaming read -- the LDS-DMA weight / record streams here -- at HALF its bytes and is
    uncalibrated for narrow loads, so `fetch_x2` is given beside it as the upper bound."""
    import shutil
    import sqlite3
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    per = {}          # counter -> kernel name -> [bytes per dispatch]
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out_dir = tempfile.mkdtemp(prefix="nerf_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", counter, "--kernel-trace", "-d", out_dir, "-o", "t", "--", sys.executable,
               os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-frame",
               "--no-stages", "--no-configs", "--no-runner-loop", "--no-traffic"]
        env = dict(os.environ, TMPDIR="/tmp")
        try:
            rc = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                timeout=240).returncode
        except subprocess.TimeoutExpired:
            return {"error": f"rocprofv3 --pmc {counter} pass timed out"}
        dbs = [os.path.join(r, f) for r, _, fs in os.walk(out_dir) for f in fs if f.endswith("_results.db")]
        if rc != 0 or not dbs:
            return {"error": f"rocprofv3 --pmc {counter} pass failed (rc {rc})"}
        cur = sqlite3.connect(dbs[0]).cursor()
        cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        kname = next(c for c in cols if "kernel_name" in c or c == "name")
        cname = next(c for c in cols if "counter_name" in c)
        vname = next(c for c in cols if c in ("value", "counter_value"))
        did = next(c for c in cols if "dispatch_id" in c)
        acc = {}
        for k, c, v, d in cur.execute(f"select {kname}, {cname}, {vname}, {did} from counters_collection"):
            if c == counter:
                acc.setdefault(k, {}).setdefault(d, 0.0)
                acc[k][d] += float(v) * 1024.0            # the counters report KiB
        per[counter] = {k: list(v.values()) for k, v in acc.items()}
        shutil.rmtree(out_dir, ignore_errors=True)

    def kernel(needle, pick=None):
        out = {}
        for counter, key in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
            vals = [v for k, vs in per[counter].items() if needle in k for v in vs]
            if pick is not None and vals:
                vals = pick(vals)
            out[key] = (sum(vals) / len(vals)) if vals else None
            out["dispatches"] = len(vals)
        if out["fetch"] is not None and out["write"] is not None:
            out["bytes_per_launch"] = out["fetch"] + out["write"]
            out["fetch_x2"] = 2 * out["fetch"]
        return out

    big = lambda vals: [v for v in vals if v >= 0.5 * max(vals)]          # the fine-pass launches of a kernel
    return {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of this bench run (2 steps), raw counter bytes",
            "render_fused_kernel": kernel("render_fused_kernel"),
            "render_fused_kernel_fine_launch": kernel("render_fused_kernel", big),
            "mlp_forward_bf16_kernel": kernel("mlp_forward_bf16_kernel"),
            "render_fused_bf16_kernel": kernel("render_fused_bf16"),
            "mlp_forward_kernel_record": kernel("mlp_forward_kernel"),
            "mlp_bwd_dx_kernel": kernel("mlp_bwd_dx_kernel"),
            "mlp_bwd_dw_kernel": kernel("mlp_bwd_dw_kernel")}


def per_rank_times(elapsed: float, steps: int, world: int, device):
    """Every rank's ms per step, gathered to all (a bad N-GPU number must be diagnosable from the line alone)."""
    mine = torch.tensor([elapsed], device=device, dtype=torch.float64)
    every = torch.empty((world,), device=device, dtype=torch.float64)
    dist.all_gather_into_tensor(every, mine)
    ms = (every / steps * 1e3).tolist()
    return {"ms_per_step_min": min(ms), "ms_per_step_max": max(ms), "rank_of_max": int(np.argmax(ms)),
            "ms_per_step": [round(v, 4) for v in ms]}, float(every.max().item())


def nccl_debug_setup(rank: int):
    """RCCL's own warnings (NCCL_DEBUG=WARN: failed IPC handles, unreachable peers, aborted communicators) go to a file
    per rank, so that a failing collective can quote them in the JSON line instead of losing them in a launcher log."""
    os.environ.setdefault("NCCL_DEBUG", "WARN")
    if "NCCL_DEBUG_FILE" not in os.environ:
        os.environ["NCCL_DEBUG_FILE"] = os.path.join(tempfile.gettempdir(), f"nerf_bench_rccl_{os.getpid()}_r{rank}.log")
    return os.environ["NCCL_DEBUG_FILE"]


def nccl_debug_tail(path, limit=1500):
    try:
        with open(path, "rb") as f:
            f.seek(0, 2)
            size = f.tell()
            f.seek(max(0, size - limit))
            return f.read().decode("utf-8", "replace").strip() or None
    except OSError:
        return None


def self_launch(args) -> int:
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as a CHILD process (never exec
    from a process that may have initialised the GPU; this one has not) and hand back its return code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the secondary fwd+bwd+Adam measurement")
    ap.add_argument("--train", action="store_true", help="(accepted for compatibility: the data-parallel training leg "
                    "-- gradient all-reduce inside the optimizer -- now runs by default for every --gpus; the headline "
                    "line is out before it starts, so a failure there cannot erase the measurement; --no-train skips it)")
    ap.add_argument("--fault-rank", type=int, default=-1, help="test hook: this rank exits (code 17) right after the "
                    "rendezvous, before the first data collective -- the others must fail within --dist-timeout, not hang")
    ap.add_argument("--no-bf16", action="store_true", help="skip the secondary bf16-MFMA render measurement")
    ap.add_argument("--no-f16x2", action="store_true", help="skip the secondary split-f16 (fp32-grade on the f16 matrix pipe) render measurement")
    ap.add_argument("--no-frame", action="store_true", help="skip the 800x800 sharded full-frame leg")
    ap.add_argument("--no-stages", action="store_true", help="skip the HBM-bound stage measurements")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 --pmc child passes that measure HBM "
                    "traffic (roofline.traffic is then null)")
    ap.add_argument("--no-configs", action="store_true", help="skip the llff (configs[3]), coarse400 (configs[0]), ship_bf16 (configs[2]) and "
                    "non-default-encoder legs")
    ap.add_argument("--no-runner-loop", action="store_true", help="skip the runners/train.py-shaped training loop leg")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to "
                    "exercise the multi-rank path on a box with fewer GPUs than ranks)")
    ap.add_argument("--force-dist", action="store_true", help="initialise the process group and issue every collective "
                    "of the N-rank path even with ONE rank (RCCL init with device_id, all_gather_into_tensor, all_reduce, "
                    "barrier on a 1-GPU box; tests/test_bench_launch.py)")
    ap.add_argument("--dist-timeout", type=int, default=120, help="process-group timeout in seconds: a stuck "
                    "collective ends the run with rc != 0 instead of hanging it")
    ap.add_argument("--leg-timeout", type=int, default=420, help="with a process group up: seconds the secondary legs "
                    "may take after the headline line is out before the process leaves with rc 3")
    ap.add_argument("--launch-check", action="store_true", help="rendezvous, collectives and the JSON line only, no "
                    "rendering: exercises the self-launch and the N-rank plumbing on a box without GPUs (CPU tests)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))       # nothing in this process has touched the GPU

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher provides WORLD_SIZE={world}")

    if args.launch_check:
        seen = torch.ones(1)
        per_rank = frame = None
        if world > 1:
            dist.init_process_group("gloo" if args.backend != "nccl" or not torch.cuda.is_available() else "nccl",
                                    timeout=datetime.timedelta(seconds=args.dist_timeout))
            if rank == args.fault_rank:      # a rank that dies between the rendezvous and the first collective
                os._exit(17)
            dist.all_reduce(seen)
            dist.barrier()
            # the collectives of the real run, on stand-in data: the per-step slab gather, the per-rank time gather and
            # the frame leg's (world, 3) statistics gather -- same shapes of output as the N-GPU line
            slab, full = torch.zeros((RAYS, 3)), torch.empty((world * RAYS, 3))
            t0 = time.perf_counter()
            for _ in range(max(1, args.steps)):
                dist.all_gather_into_tensor(full, slab)
            per_rank, _ = per_rank_times(time.perf_counter() - t0, max(1, args.steps), world, torch.device("cpu"))
            from torch_nerf.amd import shard
            lo, hi = shard.shard_range(H * W, rank, world)
            every = torch.empty((world, 3), dtype=torch.float64)
            dist.all_gather_into_tensor(every, torch.tensor([[0.0, float(-(-(hi - lo) // 131072)), float(hi - lo)]],
                                                            dtype=torch.float64))
            frame = {"launches_per_rank": [int(r[1]) for r in every.tolist()],
                     "rays_per_rank": [int(r[2]) for r in every.tolist()]}
        if rank == 0:
            print(json.dumps({"metric": "rays/sec at 4096 rays x (64+128) samples", "value": None, "unit": "rays/s",
                              "n_gpus": world, "rccl_ranks_seen": int(seen.item()), "launch_check": True,
                              "per_rank": per_rank, "frame": frame,
                              "steps": args.steps, "warmup": args.warmup}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # CPU-side torch ops: never more threads than the cgroup grants (the default is one per VISIBLE core; over-
    # subscribing a CFS quota freezes the whole process for the rest of the 100 ms period)
    torch.set_num_threads(max(1, host_cores() // max(1, world)))
    ranks_seen = 1
    dist_on = world > 1 or args.force_dist       # every `if dist_on` below is a collective of the N-rank path
    if dist_on:
        if "MASTER_ADDR" not in os.environ:      # --force-dist outside a launcher: a one-rank rendezvous on loopback
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sk.getsockname()[1]), RANK="0",
                                  WORLD_SIZE="1", LOCAL_RANK=str(local_rank))
        # the host driver of this pool supports dmabuf IPC only: with the legacy mode RCCL's (and torch's) cross-process
        # buffer sharing fails in hipIpcGetMemHandle ("invalid argument").  Exported on the boxes already; kept here
        # for launches from a bare environment
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        tmo = datetime.timedelta(seconds=args.dist_timeout)
        nccl_log = nccl_debug_setup(rank) if args.backend == "nccl" else None
        try:
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=device, timeout=tmo)
            else:
                dist.init_process_group(args.backend, timeout=tmo)
            if rank == args.fault_rank:
                os._exit(17)
            seen = torch.ones(1, device=device)
            dist.all_reduce(seen)                      # every rank really is on the communicator
            ranks_seen = int(seen.item())
        except Exception as exc:  # noqa: BLE001 -- no headline line can follow: leave the evidence on stderr and fail
            print(json.dumps({"bench_failed": "process group / first collective", "rank": rank, "world": world,
                              "error": f"{type(exc).__name__}: {exc}"[:500],
                              "rccl_warnings_tail": nccl_debug_tail(nccl_log) if nccl_log else None}),
                  file=sys.stderr, flush=True)
            raise

    from torch_nerf.amd import ops
    renderer, scene_c, scene_f, nets, flats, cam, focal, pose = build_scene(device)
    total = H * W
    n_steps = args.warmup + args.steps
    # rank r renders slab (step*world + r) of the frame: contiguous 4096-pixel ranges, resident in HBM
    pix = [((torch.arange(RAYS, device=device) + ((s * world + rank) * RAYS)) % total) for s in range(n_steps)]
    # two slabs: the all-gather of step k runs on RCCL's own stream UNDER the render kernels of step k+1 (the design
    # rule: overlap collectives with compute); a slab is reused only after the gather that filled it has completed
    gathered = [torch.empty((world * RAYS, 3), device=device) for _ in range(2)] if dist_on else None
    pending = [None, None]
    torch.manual_seed(1234 + rank)

    def step(s):
        _, f_rgb = render_step(renderer, scene_c, scene_f, pix[s], local_rank)
        if dist_on:
            slot = s & 1
            if pending[slot] is not None:
                pending[slot].wait()
            pending[slot] = dist.all_gather_into_tensor(gathered[slot], f_rgb.contiguous(), async_op=True)   # the frame slab
        return f_rgb

    def fence():
        for slot in (0, 1):                 # every slab of the timed region is assembled before the clock stops
            if dist_on and pending[slot] is not None:
                pending[slot].wait()
                pending[slot] = None
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize()

    step_events = []
    with torch.no_grad():
        for s in range(args.warmup):
            step(s)
        fence()
        ops.KERNEL_EVENTS = []
        t0 = time.perf_counter()
        for s in range(args.warmup, n_steps):
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
            step(s)
            step_events.append(e0)
        e_end = torch.cuda.Event(enable_timing=True)
        e_end.record()
        fence()
        elapsed = time.perf_counter() - t0
        events, ops.KERNEL_EVENTS = ops.KERNEL_EVENTS, None
    step_events.append(e_end)
    per_step = [a.elapsed_time(b) for a, b in zip(step_events[:-1], step_events[1:])]
    per_rank = None
    if dist_on:
        per_rank, elapsed = per_rank_times(elapsed, args.steps, world, device)

    # ---- dominant kernel: the fused render pass (sampling + encode + MLP + integral in one kernel), two launches
    # per step: coarse pass M = 4096 x 64 and fine pass M = 4096 x 192 samples.  achieved = algorithmic MLP FLOPs
    # of all its launches in the timed region / their summed HIP-event durations (so the in-kernel sampling and
    # integration count against the fraction); ms_per_launch = their mean (what rocprofv3 --stats reports as the
    # kernel's average).
    durs = [(M, e0.elapsed_time(e1)) for tag, M, e0, e1 in events if tag == "render_pass"]
    total_ms = sum(ms for _, ms in durs)
    total_flop = sum(M for M, _ in durs) * MLP_FLOP_PER_SAMPLE
    achieved = total_flop / (total_ms * 1e-3) / 1e12
    fine_ms = float(np.mean([ms for M, ms in durs if M == RAYS * (N_COARSE + N_FINE)]))
    coarse_ms = float(np.mean([ms for M, ms in durs if M == RAYS * N_COARSE]))
    traffic = None        # filled by traffic_leg() below, from counters collected in this run
    roofline = {"bound": "mfma", "achieved": round(achieved, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                "kernel": ops.DOMINANT_KERNEL + ", 2 launches/step",
                "launches": len(durs), "ms_per_launch": round(total_ms / len(durs), 4),
                "fine_ms_per_launch": round(fine_ms, 4), "coarse_ms_per_launch": round(coarse_ms, 4),
                "flop_per_launch_avg": total_flop / len(durs)}

    result = {
        "metric": "rays/sec at 4096 rays x (64+128) samples",
        "value": world * RAYS * args.steps / elapsed,
        "unit": "rays/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "ms_per_step_median": float(np.median(per_step)),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "Blender lego 800x800 geometry, 4096-ray batches, 64 coarse + 128 fine, fp32, "
                               "forward render through VolumeRenderer.render_scene (coarse net + fine net)",
                   "rays_per_gpu_per_step": RAYS, "global_rays_per_step": world * RAYS,
                   "parallelism": f"ray-shard x{world}" + (" + all-gather" if dist_on else "")},
        "rccl_ranks_seen": ranks_seen,
        "backend": args.backend if dist_on else None,
        "per_rank": per_rank,
        "roofline": roofline,
    }
    # With a process group up the headline line leaves NOW: a secondary leg that hangs in a collective (or a rank
    # that dies in one) can then no longer erase the measurement.  The enriched line printed after the legs repeats
    # every key of this one -- the last JSON line is the complete record.  A leg that outlives --leg-timeout ends
    # this process with rc 3 (the process-group timeout normally fires first and surfaces as an exception / abort).
    watchdog = None
    if dist_on:
        if rank == 0:
            print(json.dumps(dict(result, partial="headline only; the secondary legs follow on the next line")),
                  flush=True)
        watchdog = threading.Timer(args.leg_timeout, lambda: os._exit(3))
        watchdog.daemon = True
        watchdog.start()

    # secondary legs never take the headline line down with them
    def guarded(name, fn):
        try:
            return fn()
        except Exception as exc:  # noqa: BLE001
            err = {"error": f"{type(exc).__name__}: {exc}"[:300], "leg": name}
            if dist_on and args.backend == "nccl":     # what RCCL itself said (NCCL_DEBUG=WARN, per-rank file)
                err["rccl_warnings_tail"] = nccl_debug_tail(os.environ.get("NCCL_DEBUG_FILE", ""))
            return err

    if not args.no_frame:       # collective: every rank takes part
        result["frame"] = guarded("frame", lambda: frame_leg(nets, cam, rank, world, device, dist_on))
    if rank == 0 and world == 1 and not args.no_frame:
        api = guarded("frame_api", lambda: frame_api_leg(renderer, scene_c, scene_f, pose, focal))
        if "error" in api:
            result["frame_api"] = api
        else:
            result.update(api)
            if isinstance(result.get("frame"), dict) and "ms" in result["frame"]:
                result["frame_api"]["vs_frame_leg"] = result["frame_api"]["ms"] / result["frame"]["ms"]
    if rank == 0 and world == 1 and not args.no_configs:     # before the train leg: that one UPDATES the networks
        result["configs"] = guarded("configs", lambda: configs_leg(nets, flats, device))
    if rank == 0 and world == 1 and not args.no_configs:
        result["encoders"] = guarded("encoders", lambda: encoder_variants_leg(device, local_rank, max(5, args.steps // 4), 2, renderer))
        if isinstance(result.get("configs"), dict) and "error" not in result["encoders"]:
            # the same three whole-pass figures beside the other BASELINE configurations (VERDICT r04 item 4 asked for them
            # there); the full objects -- training step, frame -- stay under `encoders`
            for tag, setting in (("coord_l12", "coord_encode_level: 12"), ("dir_l5", "dir_encode_level: 5"), ("sh", "signal_encoder: sh")):
                e = result["encoders"][tag]
                result["configs"]["enc_" + tag] = {"ms": e["ms_per_step"], "rays_per_s": e["rays_per_s"], "rays": RAYS,
                                                   "mlp_frac": e["mlp_frac"], "network": e["network"],
                                                   "what": f"{setting}: 4096 rays x (64+128), coarse + fine render_scene, whole "
                                                           "pass (sampling + encode + network + integral), fp32"}
    if world == 1 and not args.no_bf16:
        result["bf16"] = guarded("bf16", lambda: bf16_leg(renderer, scene_c, scene_f, nets, pix, local_rank,
                                                          args.steps, 3))
    if world == 1 and not args.no_f16x2:
        result["f16x2"] = guarded("f16x2", lambda: f16x2_leg(renderer, scene_c, scene_f, nets, pix, local_rank,
                                                             args.steps, 3))
    if not args.no_train:       # world > 1: data-parallel (one gradient all-reduce per step inside FusedAdam)
        result["train"] = guarded("train", lambda: train_leg(renderer, scene_c, scene_f, nets, pix, device,
                                                             local_rank, max(3, args.steps // 4), 2, world))
    if rank == 0 and world == 1 and not args.no_runner_loop and not args.no_train:
        result["runner_loop"] = guarded("runner_loop", lambda: runner_loop_leg(device, local_rank,
                                                                               max(3, args.steps // 4), 2))
    if rank == 0 and world == 1 and not args.no_traffic:
        tr = guarded("traffic", traffic_leg)
        result["traffic"] = tr
        if "error" not in tr:
            f32 = tr["render_fused_kernel"]
            result["roofline"]["traffic"] = f32.get("bytes_per_launch")
            result["roofline"]["traffic_detail"] = {"fetch": f32.get("fetch"), "write": f32.get("write"),
                                                    "fetch_x2": f32.get("fetch_x2"),
                                                    "algorithmic_bytes_per_launch": ALGORITHMIC_BYTES_PER_LAUNCH}
            if isinstance(result.get("bf16"), dict) and "roofline" in result["bf16"]:
                b16 = tr["render_fused_bf16_kernel"] if tr["render_fused_bf16_kernel"].get("dispatches") else tr["mlp_forward_bf16_kernel"]
                result["bf16"]["roofline"]["traffic"] = b16.get("bytes_per_launch")
            if isinstance(result.get("train"), dict) and "roofline" in result["train"]:
                parts = [tr[k].get("bytes_per_launch") for k in ("mlp_forward_kernel_record", "mlp_bwd_dx_kernel", "mlp_bwd_dw_kernel")]
                if all(p is not None for p in parts):   # one launch of each per pass; mean over the coarse and fine pass
                    result["train"]["roofline"]["traffic"] = sum(parts)
                    result["train"]["roofline"]["traffic_detail"] = dict(zip(("forward_record", "dx_chain", "dw_gemms"), parts))
    if rank == 0 and world == 1 and not args.no_stages:
        result["hbm_stages"] = guarded("hbm_stages", lambda: hbm_stages(device))
        result["net_variants"] = guarded("net_variants", lambda: net_variants_leg(device))
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out = guarded("cpu_baseline", lambda: cpu_baseline(flats, focal, pose, device))
        if isinstance(out, tuple):
            base, one, quality = out
            result["cpu_baseline"] = base
            result["cpu_baseline_1thread"] = one
            result.update(quality)
            result["speedup_vs_cpu_baseline"] = result["value"] / base["value"]
        else:
            result["cpu_baseline"] = out
    if rank == 0:
        print(json.dumps(result), flush=True)
    if watchdog is not None:
        watchdog.cancel()
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
