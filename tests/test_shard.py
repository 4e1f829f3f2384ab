"""Ray-shard logic on CPU: partitioning, G-independent draws, and the all-gather that assembles
the image, exercised with world_size 2 over gloo (the GPU box runs the same code over RCCL)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from torch_nerf.amd import shard, synth


def test_shard_ranges_cover_exactly():
    for total in (1, 7, 640000, 762048, 160000):
        for world in (1, 2, 3, 4, 8):
            edges = [shard.shard_range(total, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1


def test_counter_uniform_matches_numpy_generator_and_slices():
    ref = synth.counter_uniform(123, 2, 5000)
    full = shard.counter_uniform(123, 2, 0, 5000, "cpu").numpy()
    assert np.array_equal(full, ref)
    part = shard.counter_uniform(123, 2, 1234, 1000, "cpu").numpy()
    assert np.array_equal(part, ref[1234:2234])
    assert ref.min() >= 0.0 and ref.max() < 1.0
    # per-ray draws do not depend on which shard asks for them
    a = shard.ray_draws(9, 0, 100, 64, 128, "cpu")
    b = shard.ray_draws(9, 40, 20, 64, 128, "cpu")
    for x, y in zip(a, b):
        assert torch.equal(x[40:60], y)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        image = torch.arange(total * 3, dtype=torch.float32).view(total, 3)  # stand-in for rendered colours
        lo, hi = shard.shard_range(total, rank, world)
        got = shard.gather_image(image[lo:hi].clone(), total)
        q.put((rank, bool(torch.equal(got, image))))
    finally:
        dist.destroy_process_group()


def test_gather_image_world2_gloo():
    for total in (10, 11):  # even and ragged shards
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
        results = dict(q.get(timeout=10) for _ in range(2))
        assert results == {0: True, 1: True}


def _grad_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
        x = torch.arange(20, dtype=torch.float32).view(4, 5) * (rank + 1)
        net(x).sum().backward()
        local = [p.grad.clone() for p in net.parameters()]
        shard.allreduce_gradients(list(net.parameters()))
        q.put((rank, [g.numpy() for g in local], [p.grad.numpy() for p in net.parameters()]))
    finally:
        dist.destroy_process_group()


def test_allreduce_gradients_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = dict()
    for _ in range(2):
        rank, local, reduced = q.get(timeout=120)
        out[rank] = (local, reduced)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for k in range(4):
        mean = (out[0][0][k] + out[1][0][k]) / 2
        np.testing.assert_allclose(out[0][1][k], mean, rtol=1e-6)
        np.testing.assert_allclose(out[1][1][k], mean, rtol=1e-6)


def test_gather_image_single_process_passthrough():
    x = torch.rand(12, 3)
    assert shard.gather_image(x, 12) is x
