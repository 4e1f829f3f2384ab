"""Host-side logic of the training step (rows f1/f2) on CPU: pixel choice, ground-truth residency checks,
the optimizer's refusal to run without the GPU, and -- with world_size 2 over gloo -- that sharding the
batch with the `world / (3 n)` loss scale and averaging gradients reproduces the full-batch gradient."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from torch_nerf.amd import shard, synth, train
from torch_nerf.amd.optim import FusedAdam


def test_centre_crop_indices_follow_the_reference_formula():
    for H, W in ((800, 800), (756, 1008), (11, 7)):
        ci, cj = (H - 1) // 2, (W - 1) // 2                                   # train.py:146-147
        rows = torch.arange(ci - ci // 2, ci + ci // 2)
        cols = torch.arange(cj - cj // 2, cj + cj // 2)
        grid = torch.cartesian_prod(rows, cols)
        want = grid[:, 0] * W + grid[:, 1]
        assert torch.equal(train.centre_crop_indices(H, W, "cpu"), want)


def test_choose_pixels_is_a_seeded_sample_without_replacement():
    g1, g2 = torch.Generator().manual_seed(5), torch.Generator().manual_seed(5)
    a, b = train.choose_pixels(60, 50, 1000, g1), train.choose_pixels(60, 50, 1000, g2)
    assert torch.equal(a, b) and a.dtype == torch.int64
    assert a.unique().numel() == 1000 and 0 <= a.min() and a.max() < 3000
    assert not torch.equal(a, train.choose_pixels(60, 50, 1000, g1))         # the stream advances
    crop = train.choose_pixels(60, 50, 300, g1, centre_crop=True)
    assert set(crop.tolist()) <= set(train.centre_crop_indices(60, 50, "cpu").tolist())


def test_device_images_keeps_views_on_the_gpu_only():
    poses = torch.eye(4).repeat(2, 1, 1)
    with pytest.raises(RuntimeError, match="GPU"):
        train.DeviceImages(torch.zeros(2, 4, 5, 3), poses, 4, 5, 10.0)
    with pytest.raises(ValueError):
        train.DeviceImages(torch.zeros(2, 19, 3), poses, 4, 5, 10.0)


def test_fused_adam_has_no_cpu_path():
    p = torch.nn.Parameter(torch.zeros(8))
    opt = FusedAdam([p], lr=1e-3)
    p.grad = torch.ones(8)
    with pytest.raises(RuntimeError, match="GPU"):
        opt.step()
    with pytest.raises(ValueError):
        FusedAdam([p], lr=-1.0)
    with pytest.raises(ValueError):
        FusedAdam([p], betas=(1.0, 0.999))


def _shard_gradient(rank, world, n, seed):
    """Gradient of this rank's shard with train_step's loss scale, on the CPU port of the path."""
    from oracle import torch_port
    flat_c = synth.nerf_flat_params(seed=3, sigma_bias=1.0, sigma_gain=30.0)
    flat_f = synth.nerf_flat_params(seed=4, sigma_bias=1.0, sigma_gain=30.0)
    pc = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in synth.split_flat_params(flat_c).items()}
    pf = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in synth.split_flat_params(flat_f).items()}
    pix = torch.from_numpy(synth.pixel_batch(seed, 40, 40, n))
    gt = torch.from_numpy(synth.counter_uniform(seed, 9, 3 * n).reshape(n, 3))
    lo, hi = shard.shard_range(n, rank, world)
    draws = shard.ray_draws(seed, lo, hi - lo, 64, 128, "cpu")
    pose = torch.from_numpy(synth.pose_spherical(20.0, -30.0, 4.0))
    c_rgb, _, f_rgb, _, _ = torch_port.render_batch(pc, pf, pix[lo:hi], 40, 40, synth.blender_focal(40), pose, 2.0,
                                                    6.0, 64, 128, draws)
    sse = torch.sum((c_rgb - gt[lo:hi]) ** 2) + torch.sum((f_rgb - gt[lo:hi]) ** 2)
    (sse * (world / (3.0 * n))).backward()
    return list(pc.values()) + list(pf.values())


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        params = _shard_gradient(rank, world, 7, 21)
        shard.allreduce_gradients(params, average=True)
        q.put((rank, np.concatenate([p.grad.reshape(-1).numpy() for p in params])))
    finally:
        dist.destroy_process_group()


def test_sharded_loss_scale_reproduces_full_batch_gradient_world2_gloo():
    torch.set_num_threads(4)
    full = np.concatenate([p.grad.reshape(-1).numpy() for p in _shard_gradient(0, 1, 7, 21)])
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert np.array_equal(got[0], got[1])
    rel = np.linalg.norm(got[0] - full) / np.linalg.norm(full)
    assert np.linalg.norm(full) > 0 and rel < 1e-5, rel
