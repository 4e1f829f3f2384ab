"""Row f1/f2/f4 on the GPU: the fused Adam kernel against the golden optimizer trajectory and the
oracle, FusedAdam against torch.optim.Adam on the real networks, the sharded training step, and a
short end-to-end training run on the procedural scene."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _nets(seed=0):
    from torch_nerf.src.network import NeRF
    torch.manual_seed(seed)
    return NeRF(63, 27).cuda(), NeRF(63, 27).cuda()


def _tol(want, before):
    """one ulp of the parameter + 2e-6 of the step taken (same bound as the oracle's golden test)"""
    return np.spacing(np.abs(want)) + 2e-6 * np.abs(want - before).max()


def test_adam_kernel_against_golden_and_phases(golden):
    from torch_nerf.amd import ops
    g = golden("f8_adam")
    init_lr, end_lr, num_iter, eps = g["config"]
    n = g["p0"].size
    for phase in (0, 1, 2, 3):                       # element offset into 16-byte aligned arenas
        arena = [torch.zeros(n + 8, device="cuda") for _ in range(4)]
        P, G, M, V = (a[phase:phase + n] for a in arena)
        P.copy_(torch.from_numpy(g["p0"]))
        before = g["p0"]
        for s in range(g["grads"].shape[0]):
            G.copy_(torch.from_numpy(g["grads"][s]))
            ops.adam_step(P, G, M, V, s + 1, float(g["lrs"][s]), eps=float(eps))
            got, want = P.cpu().numpy(), g["params"][s]
            assert np.all(np.abs(got - want) <= _tol(want, before)), (phase, s)
            before = want
        np.testing.assert_allclose(M.cpu().numpy(), g["exp_avg"], rtol=2e-6, atol=1e-12)
        np.testing.assert_allclose(V.cpu().numpy(), g["exp_avg_sq"], rtol=2e-6, atol=1e-30)
        for a in arena:                              # nothing outside the blob was touched
            assert float(a[:phase].abs().sum()) == 0.0 and float(a[phase + n:].abs().sum()) == 0.0


def test_adam_kernel_against_oracle_full_size(oracle):
    from torch_nerf.amd import ops
    rng = np.random.RandomState(5)
    n = 2 * 595844
    p = rng.uniform(-0.1, 0.1, n).astype(np.float32)
    m = (rng.standard_normal(n) * 1e-3).astype(np.float32)
    v = (rng.uniform(0, 1e-5, n)).astype(np.float32)
    g = (rng.standard_normal(n) * 1e-2).astype(np.float32)
    P, G, M, V = (torch.from_numpy(a.copy()).cuda() for a in (p, g, m, v))
    ops.adam_step(P, G, M, V, 137, 3.3e-4, grad_scale=0.5)
    before = p.copy()
    oracle.adam_step(p, (g * np.float32(0.5)).astype(np.float32), m, v, 137, 3.3e-4)
    assert np.all(np.abs(P.cpu().numpy() - p) <= _tol(p, before))
    np.testing.assert_allclose(M.cpu().numpy(), m, rtol=2e-6, atol=1e-12)
    np.testing.assert_allclose(V.cpu().numpy(), v, rtol=2e-6, atol=1e-30)


def test_adam_rejects_bad_arguments():
    from torch_nerf.amd import ops
    P = torch.zeros(16, device="cuda")
    with pytest.raises(RuntimeError):
        ops.adam_step(P.cpu(), P, P.clone(), P.clone(), 1, 1e-3)
    with pytest.raises(ValueError):
        ops.adam_step(P, P[:8], P.clone(), P.clone(), 1, 1e-3)
    with pytest.raises(RuntimeError, match="phase"):
        ops.adam_step(P[1:9], P.clone()[:8], P.clone()[1:9], P.clone()[1:9], 1, 1e-3)
    with pytest.raises(RuntimeError, match="step counts from 1"):
        ops.adam_step(P, P.clone(), P.clone(), P.clone(), 0, 1e-3)


def _fake_backward(nets, seed):
    """Give every parameter a gradient the way the MLP backward does: one flat blob per network."""
    gen = torch.Generator(device="cuda").manual_seed(seed)
    for net in nets:
        params = net._ordered_params()
        flat = torch.randn(sum(p.numel() for p in params), device="cuda", generator=gen) * 1e-2
        off = 0
        for p in params:
            p.grad = flat[off:off + p.numel()].view(p.shape)
            off += p.numel()


def test_fused_adam_tracks_torch_adam_and_shares_the_blob():
    from torch_nerf.amd.optim import FusedAdam
    mine, ref = _nets(3), _nets(3)
    for a, b in zip(mine, ref):
        b.load_state_dict(a.state_dict())
    params = [p for net in mine for p in net.parameters()]
    opt = FusedAdam(params, lr=5e-4, eps=1e-8)
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, 0.9)
    ref_params = [p for net in ref for p in net.parameters()]
    ropt = torch.optim.Adam(ref_params, lr=5e-4, eps=1e-8)
    rsched = torch.optim.lr_scheduler.ExponentialLR(ropt, 0.9)
    x = torch.rand(256, 3, device="cuda") * 4 - 2
    d = torch.rand(256, 3, device="cuda") * 2 - 1
    with torch.no_grad():
        before = mine[0].forward_fused(x, d)[1].clone()
    for s in range(4):
        _fake_backward(mine, 10 + s)
        _fake_backward(ref, 10 + s)
        if s == 2:                                    # torch skips parameters that have no gradient
            mine[1].fc_3.weight.grad = None
            ref[1].fc_3.weight.grad = None
        opt.step(); sched.step()
        ropt.step(); rsched.step()
    for (name, p), q in zip([(k, v) for net in mine for k, v in net.named_parameters()], ref_params):
        # 4 steps of ~lr each; the two formulations differ by rounding only
        torch.testing.assert_close(p, q, rtol=0, atol=5e-8, msg=name)
    # the networks read the optimizer's blob directly, and notice that it changed
    for i, net in enumerate(mine):
        _, flat, _ = net._stream()
        assert flat.data_ptr() == opt._arenas[0].P.data_ptr() + 4 * i * 595844
        assert flat.numel() == 595844
    with torch.no_grad():
        after = mine[0].forward_fused(x, d)[1]
        want = ref[0].forward_fused(x, d)[1]
    assert not torch.equal(after, before)
    torch.testing.assert_close(after, want, rtol=0, atol=2e-5)
    # torch.optim.Adam can resume from the fused optimizer's checkpoint (runner_utils.py:758-775 saves it)
    sd = opt.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == 4.0
    skipped = [i for i, p in enumerate(params) if p is mine[1].fc_3.weight][0]
    assert float(sd["state"][skipped]["step"]) == 3.0
    ropt2 = torch.optim.Adam(ref_params, lr=1.0)
    ropt2.load_state_dict(sd)
    torch.testing.assert_close(ropt2.state[ref_params[0]]["exp_avg"], ropt.state[ref_params[0]]["exp_avg"],
                               rtol=1e-5, atol=1e-9)
    # and the fused optimizer resumes from torch's
    opt2 = FusedAdam(params, lr=1.0)
    opt2.load_state_dict(ropt.state_dict())
    _fake_backward(mine, 99); _fake_backward(ref, 99)
    opt2.step(); ropt.step()
    for p, q in zip(params, ref_params):
        torch.testing.assert_close(p, q, rtol=0, atol=5e-8)


def test_checkpoint_round_trip_between_fused_and_torch_adam(tmp_path):
    """A checkpoint written while training with FusedAdam (reference layout, runner_utils.py:737-775) resumes
    under torch.optim.Adam and keeps stepping identically to the fused optimizer."""
    from torch_nerf.amd import checkpoint
    from torch_nerf.amd.optim import FusedAdam
    from torch_nerf.src.scene import PrimitiveCube
    from torch_nerf.src.signal_encoder import PositionalEncoder
    enc = {"coord_enc": PositionalEncoder(3, 10, True), "dir_enc": PositionalEncoder(3, 4, True)}
    nets = _nets(11)
    scenes = [PrimitiveCube(n, enc) for n in nets]
    params = [p for n in nets for p in n.parameters()]
    opt = FusedAdam(params, lr=5e-4)
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, 0.99)
    for s in range(2):
        _fake_backward(nets, 50 + s)
        opt.step(); sched.step()
    checkpoint.save_checkpoint(tmp_path, 2, scenes[0], scenes[1], opt, sched)
    other = _nets(12)
    oscenes = [PrimitiveCube(n, enc) for n in other]
    oparams = [p for n in other for p in n.parameters()]
    topt = torch.optim.Adam(oparams, lr=1.0)
    tsched = torch.optim.lr_scheduler.ExponentialLR(topt, 0.5)
    assert checkpoint.load_checkpoint(tmp_path, oscenes[0], oscenes[1], topt, tsched) == 2
    for p, q in zip(params, oparams):
        assert q.is_cuda and torch.equal(p, q)
    _fake_backward(nets, 60); _fake_backward(other, 60)
    opt.step(); topt.step()
    for p, q in zip(params, oparams):
        torch.testing.assert_close(p, q, rtol=0, atol=5e-8)


def _views(size=48, n_views=6):
    from torch_nerf.amd import procedural
    return procedural.make_views(n_views, size, size, torch.device("cuda"))


def _camera(pose, size, focal):
    from torch_nerf.src.renderer.cameras import PerspectiveCamera
    return PerspectiveCamera({"f_x": focal, "f_y": focal, "img_width": size, "img_height": size}, pose, 2.0, 6.0)


def test_sharded_step_equals_full_batch_step():
    """Two half-batch shards (what ranks 0 and 1 of a 2-GPU job compute) add up to the full-batch
    gradient: same draws per global ray, loss scaled by world / (3 n)."""
    from torch_nerf.amd import train, shard
    from torch_nerf.src.renderer.ray_samplers import StratifiedSampler
    images, poses, focal = _views()
    cam = _camera(poses[1], 48, focal)
    nets = _nets(5)
    with torch.no_grad():                            # make the density field non-trivial
        for net in nets:
            net.fc_8.bias[0] += 0.5
    gen = torch.Generator(device="cuda").manual_seed(1)
    pix = train.choose_pixels(48, 48, 301, gen)     # odd count: unequal shards
    assert pix.unique().numel() == 301
    params = [p for net in nets for p in net.parameters()]

    def grads_of(lo, hi, scale):
        for p in params:
            p.grad = None
        draws = shard.ray_draws(7 * 1000003 + 4, lo, hi - lo, 64, 128, "cuda")
        c, f = train._render_pair(cam, nets[0], nets[1], pix[lo:hi], 64, 128, False, draws, StratifiedSampler())
        gt = images[1].index_select(0, pix[lo:hi])
        ((torch.sum((c - gt) ** 2) + torch.sum((f - gt) ** 2)) * scale).backward()
        return torch.cat([p.grad.reshape(-1) for p in params]).clone()

    full = grads_of(0, 301, 1.0 / (3 * 301))
    (a0, a1), (b0, b1) = shard.shard_range(301, 0, 2), shard.shard_range(301, 1, 2)
    halves = (grads_of(a0, a1, 2.0 / (3 * 301)) + grads_of(b0, b1, 2.0 / (3 * 301))) / 2
    rel = (torch.linalg.vector_norm(halves - full) / torch.linalg.vector_norm(full)).item()
    assert rel < 2e-5, rel


def test_full_size_batch_gradient_is_additive_over_ray_chunks():
    """BASELINE size (4096 rays x (64 + 192) samples, 800x800 camera): the gradient of the whole batch equals the
    sum of the gradients of its four 1024-ray chunks -- a size-independent property of the whole
    forward/backward chain (ray generation, both samplers, both MLPs, both integrals)."""
    from torch_nerf.amd import train, shard, synth
    from torch_nerf.src.renderer.ray_samplers import StratifiedSampler
    size, n = 800, 4096
    focal = float(synth.blender_focal(size))
    cam = _camera(torch.from_numpy(synth.pose_spherical(37.0, -30.0, 4.0)), size, focal)
    nets = _nets(7)
    with torch.no_grad():
        for net in nets:
            net.fc_8.bias[0] += 0.5
    gen = torch.Generator(device="cuda").manual_seed(3)
    pix = train.choose_pixels(size, size, n, gen)
    gt = torch.rand((n, 3), device="cuda", generator=gen)
    params = [p for net in nets for p in net.parameters()]

    def grads_of(lo, hi):
        for p in params:
            p.grad = None
        draws = shard.ray_draws(11, lo, hi - lo, 64, 128, "cuda")
        c, f = train._render_pair(cam, nets[0], nets[1], pix[lo:hi], 64, 128, False, draws, StratifiedSampler())
        ((torch.sum((c - gt[lo:hi]) ** 2) + torch.sum((f - gt[lo:hi]) ** 2)) / (3 * n)).backward()
        return torch.cat([p.grad.reshape(-1) for p in params]).double()

    full = grads_of(0, n)
    parts = sum(grads_of(lo, lo + 1024) for lo in range(0, n, 1024))
    rel = (torch.linalg.vector_norm(parts - full) / torch.linalg.vector_norm(full)).item()
    assert torch.isfinite(full).all() and torch.linalg.vector_norm(full) > 0
    assert rel < 1e-5, rel


def test_choose_pixels_centre_crop_and_determinism():
    from torch_nerf.amd import train
    g1 = torch.Generator(device="cuda").manual_seed(11)
    g2 = torch.Generator(device="cuda").manual_seed(11)
    a = train.choose_pixels(100, 80, 512, g1, centre_crop=True)
    b = train.choose_pixels(100, 80, 512, g2, centre_crop=True)
    assert torch.equal(a, b) and a.unique().numel() == 512
    rows, cols = a // 80, a % 80
    ci, cj = 99 // 2, 79 // 2                       # train.py:146-147
    assert rows.min() >= ci - ci // 2 and rows.max() < ci + ci // 2
    assert cols.min() >= cj - cj // 2 and cols.max() < cj + cj // 2
    c = train.choose_pixels(100, 80, 512, g1)
    assert c.unique().numel() == 512 and c.max() < 8000


def test_training_converges_on_procedural_scene():
    """600 steps (about 5 s) of the device-resident training step: PSNR on the training batches must rise
    clearly.  Initial weights are nn.Linear defaults plus +0.3 on the density bias: with the plain default
    some seeds start with relu(fc_8[0]) == 0 on most samples and sit at the all-black image for a long time
    (in the reference as well), which would make the test a coin flip."""
    from torch_nerf.amd import train
    from torch_nerf.amd.optim import FusedAdam
    size = 48
    images, poses, focal = _views(size, 6)
    data = train.DeviceImages(images, poses, size, size, focal)
    nets = _nets(2)
    with torch.no_grad():
        for net in nets:
            net.fc_8.bias[0] += 0.3
    opt = FusedAdam([p for net in nets for p in net.parameters()], lr=5e-4, eps=1e-8)
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, pow(0.1, 1 / 300000))
    gen = torch.Generator(device="cuda").manual_seed(0)
    rng = np.random.RandomState(0)
    history = []
    for step in range(600):
        view = int(rng.randint(len(data)))
        cam = _camera(data.poses[view], size, focal)
        pix = train.choose_pixels(size, size, 1024, gen)
        c_sse, f_sse = train.train_step(cam, nets[0], nets[1], opt, data.images[view], pix, 64, 128, False,
                                        seed=3, step=step, scheduler=sched)
        history.append(f_sse / (3 * 1024))
    mse = torch.stack(history).cpu().numpy()
    first, last = mse[:10].mean(), mse[-50:].mean()
    psnr_first, psnr_last = -10 * np.log10(first), -10 * np.log10(last)
    assert np.isfinite(mse).all()
    print("convergence test: psnr_first", psnr_first, "psnr_last", psnr_last)
    # measured with this tree's kernels: 7.5 dB -> 23.1 dB in 600 steps; the floor is that minus 1 dB (was 17 dB, set with
    # round 1's kernels).  The long run -- 6000 steps, 37-41 dB held out -- is profiles/r05_train_procedural_6000.json
    assert psnr_last > psnr_first + 6.0 and psnr_last > 22.0, (psnr_first, psnr_last)


def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from torch_nerf.amd import train
    from torch_nerf.amd.optim import FusedAdam
    torch.cuda.set_device(0)                         # both ranks share the box's single GPU
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank, _two_steps(world).cpu().numpy()))
    finally:
        dist.destroy_process_group()


def _two_steps(world):
    from torch_nerf.amd import train
    from torch_nerf.amd.optim import FusedAdam
    images, poses, focal = _views(32, 2)
    nets = _nets(2)
    with torch.no_grad():
        for net in nets:
            net.fc_8.bias[0] += 0.5
    opt = FusedAdam([p for net in nets for p in net.parameters()], lr=5e-4)
    gen = torch.Generator(device="cuda").manual_seed(4)
    for step in range(2):
        pix = train.choose_pixels(32, 32, 257, gen)
        train.train_step(_camera(poses[step], 32, focal), nets[0], nets[1], opt, images[step], pix, 64, 128,
                         False, seed=1, step=step)
    return torch.cat([p.detach().reshape(-1) for net in nets for p in net.parameters()])


def test_fused_adam_keeps_step_counts_when_a_parameter_is_rehomed():
    """Reassigning p.data (an EMA swap, a .to(), a manual re-initialisation) takes the parameter out of the
    optimizer's blob; the arena is rebuilt on the next step and must carry the step counts over (they live as
    host integers between checkpoints) -- otherwise Adam's bias correction restarts from step 1."""
    from torch_nerf.amd.optim import FusedAdam
    mine, ref = _nets(3), _nets(3)
    for a, b in zip(mine, ref):
        b.load_state_dict(a.state_dict())
    params = [p for net in mine for p in net.parameters()]
    ref_params = [p for net in ref for p in net.parameters()]
    opt = FusedAdam(params, lr=5e-4, eps=1e-8)
    ropt = torch.optim.Adam(ref_params, lr=5e-4, eps=1e-8)
    for s in range(6):
        _fake_backward(mine, 40 + s)
        _fake_backward(ref, 40 + s)
        if s == 3:      # mid-training: fresh storage for one tensor of each network, same values
            for net in mine:
                net.fc_2.weight.data = net.fc_2.weight.data.clone()
            assert not opt._arenas[0].intact()
        opt.step()
        ropt.step()
    assert opt._arenas[0].intact() and opt._arenas[0].steps == [6] * len(params)
    assert float(opt.state_dict()["state"][0]["step"]) == 6.0
    for (name, p), q in zip([(k, v) for net in mine for k, v in net.named_parameters()], ref_params):
        torch.testing.assert_close(p, q, rtol=0, atol=8e-8, msg=name)


def _uneven_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from torch_nerf.amd.optim import FusedAdam
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        nets = _nets(5)
        opt = FusedAdam([p for net in nets for p in net.parameters()], lr=5e-4)
        if rank == 0:
            _fake_backward(nets, 7)          # rank 1 back-propagated nothing this step (p.grad is None everywhere)
        opt.step()
        q.put((rank, torch.cat([p.detach().reshape(-1) for net in nets for p in net.parameters()]).cpu().numpy()))
    finally:
        dist.destroy_process_group()


def test_data_parallel_step_with_a_rank_without_gradients_does_not_hang():
    """Every rank joins the gradient all-reduce, contributing zeros if it has no gradient: the step is the
    average over ranks and the replicas stay identical (a rank that skipped the collective would hang the job)."""
    import torch.multiprocessing as mp
    from torch_nerf.amd.optim import FusedAdam
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_uneven_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert np.array_equal(got[0], got[1])
    nets = _nets(5)
    _fake_backward(nets, 7)
    for net in nets:
        for p in net.parameters():
            p.grad = p.grad * 0.5           # mean of (g, 0)
    opt = FusedAdam([p for net in nets for p in net.parameters()], lr=5e-4)
    opt.step()
    want = torch.cat([p.detach().reshape(-1) for net in nets for p in net.parameters()]).cpu().numpy()
    np.testing.assert_allclose(got[0], want, rtol=0, atol=1e-7)


def _ragged_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from torch_nerf.amd.optim import FusedAdam
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        nets = _nets(5)
        params = [p for net in nets for p in net.parameters()]
        opt = FusedAdam(params, lr=5e-4)
        for s in range(3):
            _fake_backward(nets, 11 + s)
            # ragged shards: rank 1 has holes (fc_3 of both networks), NO rank has a gradient for fc_7.bias
            for net in nets:
                net.fc_7.bias.grad = None
                if rank == 1:
                    net.fc_3.weight.grad = None
                    net.fc_3.bias.grad = None
            opt.step()
        steps = list(opt._arenas[0].steps)
        q.put((rank, torch.cat([p.detach().reshape(-1) for p in params]).cpu().numpy(), steps))
    finally:
        dist.destroy_process_group()


def test_data_parallel_step_with_ragged_gradient_holes():
    """ranks whose gradient sets differ take different host paths through FusedAdam.step (a rank with
    every gradient never reads the reduced has-gradient flags back, a rank with a hole does).  Both must end with
    identical parameters and identical per-parameter step counts: a parameter steps iff SOME rank holds a gradient
    for it (what DDP + torch.optim.Adam do), with the mean over ranks of what was contributed."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ragged_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {r: (v, st) for r, v, st in (q.get(timeout=600) for _ in procs)}
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert np.array_equal(got[0][0], got[1][0]) and got[0][1] == got[1][1]
    # reference: one process, torch.optim.Adam on the rank-averaged gradients (both ranks ran the same _fake_backward)
    nets = _nets(5)
    params = [p for net in nets for p in net.parameters()]
    ropt = torch.optim.Adam(params, lr=5e-4, eps=1e-8)
    names = [k for net in nets for k, _ in net.named_parameters()]
    for s in range(3):
        _fake_backward(nets, 11 + s)
        for net in nets:
            net.fc_7.bias.grad = None
            net.fc_3.weight.grad = net.fc_3.weight.grad * 0.5        # mean of (g, 0)
            net.fc_3.bias.grad = net.fc_3.bias.grad * 0.5
        ropt.step()
    want = torch.cat([p.detach().reshape(-1) for p in params]).cpu().numpy()
    np.testing.assert_allclose(got[0][0], want, rtol=0, atol=2e-7)
    assert [st for st, n in zip(got[0][1], names) if n == "fc_7.bias"] == [0, 0]      # never stepped
    assert all(st == 3 for st, n in zip(got[0][1], names) if n != "fc_7.bias")


def test_training_step_is_deterministic():
    """No atomics anywhere on the path: the same two steps twice give bit-identical parameters."""
    assert torch.equal(_two_steps(1), _two_steps(1))


def test_data_parallel_training_matches_single_process():
    """world_size 2 (two processes on the one GPU, gloo carrying the gradient all-reduce) ends two
    optimisation steps with the same parameters as one process on the whole batch."""
    import torch.multiprocessing as mp
    single = _two_steps(1).cpu().numpy()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert np.array_equal(got[0], got[1])            # replicas stay bit-identical
    # Adam normalises the step (|update| ~ lr = 5e-4), so a parameter whose gradient is within rounding of
    # zero may move differently; everything else agrees to a small fraction of one step
    diff = np.abs(got[0] - single)
    assert np.mean(diff > 0.02 * 5e-4) < 1e-3 and diff.max() <= 2.2 * 5e-4, (np.mean(diff > 1e-5), diff.max())
