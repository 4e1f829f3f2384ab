#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING THE REFERENCE.

Runs only in the build container, where /root/reference exists; the .npz files it
writes are committed and are what travels to the GPU box.  Inputs are produced by
the build's own deterministic generators (torch_nerf.amd.synth); outputs are
whatever the reference's CPU path (torch 2.10.0 CPU) returns for them.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

Fixture plan = SURVEY.md section 8(c) F1..F7.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = os.environ.get("NERF_REFERENCE", "/root/reference")

# the build's generators (numpy only) -- loaded by file path so that the build's own
# `torch_nerf` package does not shadow the reference's `torch_nerf`
import importlib.util

_spec = importlib.util.spec_from_file_location(
    "nerf_synth", os.path.join(ROOT, "torch-nerf_amd", "torch_nerf", "amd", "synth.py"))
synth = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(synth)

sys.path.insert(0, REFERENCE)
import torch_nerf.src.renderer.cameras as ref_cameras  # noqa: E402
import torch_nerf.src.renderer.integrators.quadrature_integrator as ref_integrators  # noqa: E402
import torch_nerf.src.renderer.ray_samplers as ref_samplers  # noqa: E402
import torch_nerf.src.renderer.volume_renderer as ref_vr  # noqa: E402
import torch_nerf.src.network.nerf as ref_nerf  # noqa: E402
import torch_nerf.src.scene as ref_scene  # noqa: E402
from torch_nerf.src.renderer.ray_samplers.utils import sample_pdf as ref_sample_pdf  # noqa: E402
from torch_nerf.src.signal_encoder.positional_encoder import PositionalEncoder as RefPE  # noqa: E402

assert ref_vr.__file__.startswith(REFERENCE), ref_vr.__file__


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"wrote {path}: {os.path.getsize(path)/1024:.1f} KiB")


def load_ref_net(flat):
    net = ref_nerf.NeRF(63, 27)
    sd = {k: torch.from_numpy(v.copy()) for k, v in synth.split_flat_params(flat).items()}
    net.load_state_dict(sd)
    return net


def camera(H, W, focal, pose, near, far):
    return ref_cameras.PerspectiveCamera(
        {"f_x": focal, "f_y": focal, "img_width": W, "img_height": H},
        torch.from_numpy(np.asarray(pose, np.float32).copy()), near, far)


# ---------------------------------------------------------------- F1 raygen
def f1_raygen():
    out = {}
    sampler = ref_samplers.StratifiedSampler()
    integ = ref_integrators.QuadratureIntegrator()
    cases = [
        ("blender", 800, 800, float(synth.blender_focal(800)), synth.pose_spherical(37.0, -30.0, 4.0),
         2.0, 6.0, False),
        ("blender400", 400, 400, float(synth.blender_focal(400)),
         synth.pose_spherical(-153.0, -30.0, 4.0), 2.0, 6.0, False),
        ("llff_ndc0", 756, 1008, 815.0, synth.llff_like_pose(), 0.0, 1.0, True),
        ("llff_ndc1", 756, 1008, 815.0, synth.llff_like_pose(), 1.0, 1.0 + 1.0, True),
    ]
    for name, H, W, focal, pose, near, far, ndc in cases:
        cam = camera(H, W, focal, pose, near, far)
        vr = ref_vr.VolumeRenderer(integ, sampler, cam)
        corners = np.array([0, W - 1, (H - 1) * W, H * W - 1, (H // 2) * W + W // 2], np.int64)
        pix = np.concatenate([corners, synth.pixel_batch(11, H, W, 251)])
        coords = vr.screen_coords.clone()[torch.from_numpy(pix), :]
        bundle = sampler.generate_rays(coords, cam, project_to_ndc=ndc)
        out[name + "_meta"] = np.array([H, W, focal, near, far, float(ndc)], np.float64)
        out[name + "_pose"] = np.asarray(pose, np.float32)
        out[name + "_pix"] = pix
        out[name + "_coords"] = coords.numpy()
        out[name + "_o"] = bundle.ray_origin.numpy()
        out[name + "_d"] = bundle.ray_dir.numpy()
        out[name + "_intrinsic"] = cam.intrinsic.numpy()
    # full screen-coordinate table checksum for one small camera
    cam = camera(6, 5, 10.0, synth.pose_spherical(0.0, -30.0, 4.0), 2.0, 6.0)
    vr = ref_vr.VolumeRenderer(integ, sampler, cam)
    out["small_coords_6x5"] = vr.screen_coords.numpy()
    # analytic pose generator parity (load_blender.pose_spherical needs imageio: optional)
    save("f1_raygen", **out)


def rays_for(n, seed):
    """Blender rays from a seeded pixel batch via the reference."""
    H = W = 800
    cam = camera(H, W, float(synth.blender_focal(W)), synth.pose_spherical(37.0, -30.0, 4.0), 2.0, 6.0)
    sampler = ref_samplers.StratifiedSampler()
    vr = ref_vr.VolumeRenderer(ref_integrators.QuadratureIntegrator(), sampler, cam)
    pix = synth.pixel_batch(seed, H, W, n)
    bundle = sampler.generate_rays(vr.screen_coords.clone()[torch.from_numpy(pix), :], cam, False)
    return cam, vr, pix, bundle


# ---------------------------------------------------------------- F2 coarse sampling
def f2_coarse():
    out = {}
    sampler = ref_samplers.StratifiedSampler()
    for name, near, far, S in (("b", 2.0, 6.0, 64), ("ndc", 0.0, 1.0, 64), ("odd", 0.5, 3.25, 40)):
        _, _, _, bundle = rays_for(48, 5)
        bundle = ref_samplers.RayBundle(bundle.ray_origin, bundle.ray_dir, near, far, False)
        torch.manual_seed(1234)
        u1 = torch.rand((48, S))
        torch.manual_seed(1234)
        pts, dirs, delta = sampler.sample_along_rays(bundle, S, device="cpu")
        t_bins, ps = sampler._create_t_bins(near, far, S, "cpu")
        out[name + "_meta"] = np.array([near, far, S], np.float64)
        out[name + "_o"] = bundle.ray_origin.numpy()
        out[name + "_d"] = bundle.ray_dir.numpy()
        out[name + "_u1"] = u1.numpy()
        out[name + "_t_bins"] = t_bins.numpy()
        out[name + "_ps"] = np.array([ps], np.float64)
        out[name + "_pts"] = pts.numpy()
        out[name + "_dirs"] = dirs.numpy()
        out[name + "_delta"] = delta.numpy()
    save("f2_coarse", **out)


# ---------------------------------------------------------------- F3 fine sampling
def adversarial_weights(n, S, seed):
    rng = np.random.RandomState(seed)
    w = rng.rand(n, S).astype(np.float32)
    w[0] = 0.0                                   # all-zero row -> uniform pdf from the 1e-5 floor
    w[1] = 0.0; w[1, 17 % S] = 1.0               # one-hot
    w[2] = 0.0; w[2, S - 1] = 1.0                # all mass in the last bin
    w[3] = 0.0; w[3, 0] = 1.0                    # all mass in the first bin
    w[4] = (10.0 ** rng.uniform(-12, 0, S)).astype(np.float32)   # huge dynamic range
    w[5] = 1e-12                                 # far below the floor
    w[6] = np.float32(1.0)                       # exactly uniform
    w[7] = np.exp(-0.5 * ((np.arange(S) - S / 2) / 2.0) ** 2).astype(np.float32)  # a narrow peak
    w[8:16] = (rng.rand(8, S) ** 8).astype(np.float32)
    # rows 16.. look like real compositing weights: T*alpha with decaying transmittance
    sig = rng.gamma(0.5, 4.0, size=(n - 16, S)).astype(np.float32)
    tau = sig * np.float32(4.0 / S)
    T = np.exp(-np.concatenate([np.zeros((n - 16, 1), np.float32), np.cumsum(tau, 1)[:, :-1]], 1))
    w[16:] = (T * (1 - np.exp(-tau))).astype(np.float32)
    return w


def f3_fine():
    out = {}
    sampler = ref_samplers.StratifiedSampler()
    for name, near, far, Sc, Sf, n in (("b", 2.0, 6.0, 64, 128, 96), ("ndc", 0.0, 1.0, 64, 128, 32),
                                       ("s128", 2.0, 6.0, 128, 64, 32), ("s40", 2.0, 6.0, 40, 24, 32),
                                       ("s1000", 2.0, 6.0, 1000, 16, 24)):
        _, _, _, bundle = rays_for(n, 7)
        bundle = ref_samplers.RayBundle(bundle.ray_origin, bundle.ray_dir, near, far, False)
        w_in = adversarial_weights(n, Sc, 99)
        torch.manual_seed(4321)
        u1 = torch.rand((n, Sc)); u2 = torch.rand((n, Sf)); u3 = torch.rand((n, Sf))
        # (a) the whole branch through the public API
        w_t = torch.from_numpy(w_in.copy())
        torch.manual_seed(4321)
        pts, dirs, delta = sampler.sample_along_rays(bundle, (Sc, Sf), device="cpu", weights=w_t)
        # (b) the bin indices, by replaying sample_pdf's arithmetic with ATen ops on the same draws
        t_bins, ps = sampler._create_t_bins(near, far, Sc, "cpu")
        w2 = torch.from_numpy(w_in.copy())
        w2 += 1e-5
        pdf = w2 / torch.sum(w2, dim=-1, keepdim=True)
        cdf = torch.cumsum(pdf, dim=-1)
        cdf = torch.cat([torch.zeros((n, 1)), cdf[..., :-1]], dim=-1)
        idx = torch.searchsorted(cdf, u2.contiguous(), right=True) - 1
        t_f = torch.gather(t_bins.unsqueeze(0).repeat(n, 1), 1, idx) + ps * u3
        t_c = t_bins.unsqueeze(0).repeat(n, 1) + ps * u1
        t_sorted, _ = torch.sort(torch.cat([t_c, t_f], -1), -1)
        # the replay must reproduce the API call bit for bit, or the replay is wrong
        d_chk = torch.diff(torch.cat([t_sorted, 1e8 * torch.ones((n, 1))], -1), n=1, dim=-1)
        assert torch.equal(d_chk, delta), "replayed draws do not match the reference API call"
        assert torch.equal(w2, w_t), "in-place weight floor mismatch"
        out[name + "_meta"] = np.array([near, far, Sc, Sf], np.float64)
        out[name + "_o"] = bundle.ray_origin.numpy(); out[name + "_d"] = bundle.ray_dir.numpy()
        out[name + "_w_in"] = w_in; out[name + "_w_after"] = w_t.numpy()
        out[name + "_u1"] = u1.numpy(); out[name + "_u2"] = u2.numpy(); out[name + "_u3"] = u3.numpy()
        out[name + "_t_bins"] = t_bins.numpy(); out[name + "_ps"] = np.array([ps], np.float64)
        out[name + "_norm"] = torch.sum(w2, dim=-1).numpy()
        out[name + "_idx"] = idx.numpy().astype(np.int16)
        out[name + "_t"] = t_sorted.numpy()
        out[name + "_delta"] = delta.numpy()
        out[name + "_pts"] = pts.numpy()
    save("f3_fine", **out)


# ---------------------------------------------------------------- F4 positional encoding
def f4_posenc():
    rng = np.random.RandomState(3)
    x = np.concatenate([rng.uniform(-6, 6, (192, 3)), rng.uniform(-1.5, 1.5, (60, 3)),
                        np.array([[0.0, -0.0, 1.0], [6.0, -6.0, 3.14159265], [1e-8, 100.0, -250.0],
                                  [0.5, 0.25, -0.125]])]).astype(np.float32)
    pe10 = RefPE(3, 10, True).encode(torch.from_numpy(x)).numpy()
    pe4 = RefPE(3, 4, True).encode(torch.from_numpy(x)).numpy()
    pe4n = RefPE(3, 4, False).encode(torch.from_numpy(x)).numpy()
    save("f4_posenc", x=x, pe10=pe10, pe4=pe4, pe4_noinput=pe4n)


GRAD_SLICE = 192


def grad_digest(net):
    """Per-tensor norm + sum + leading slice + strided sample of the parameter grads."""
    d = {}
    for k, p in net.named_parameters():
        g = p.grad.detach().reshape(-1).numpy()
        d[k + ".norm"] = np.array([np.sqrt(np.sum(g.astype(np.float64) ** 2))])
        d[k + ".sum"] = np.array([np.sum(g.astype(np.float64))])
        d[k + ".head"] = g[:GRAD_SLICE].copy()
        d[k + ".stride"] = g[:: max(1, g.size // GRAD_SLICE)][:GRAD_SLICE].copy()
        # every element takes part in two more numbers (round 5): the sums along both axes of a weight gradient (a wrong
        # element anywhere moves one row sum and one column sum), and bias gradients whole
        if p.grad.ndim == 2:
            g2 = p.grad.detach().numpy().astype(np.float64)
            d[k + ".rowsum"] = g2.sum(axis=1)
            d[k + ".colsum"] = g2.sum(axis=0)
        else:
            d[k + ".full"] = g.copy()
    return d


class ReluSigns:
    """The reference's own ReLU decisions, captured with a forward hook on its ONE nn.ReLU module (nerf.py:102-118 calls
    it ten times: h0..h4, h5..h7, sigma, h9).  Stored as packed bits in the oracle's layout (M, 8 F + F/2 + 1) =
    [h0 .. h7 | h9 | sigma]: the tests require the kernels' decoded masks to agree with them and
    then compare gradients on mask-identical fixtures at summation-order tolerance instead of a flip-blind one."""

    def __init__(self, net):
        self.calls = []
        self.handle = net.relu_actvn.register_forward_hook(lambda m, i, o: self.calls.append((o.detach() > 0).numpy()))

    def packed(self):
        self.handle.remove()
        assert len(self.calls) == 10, len(self.calls)
        planes = self.calls[:8] + [self.calls[9], self.calls[8][:, None]]
        bits = np.concatenate(planes, axis=1).astype(np.uint8)
        return np.packbits(bits.reshape(-1)), np.array(bits.shape, np.int64)

    def row_hashes(self):
        """One 64-bit digest per sample of its 8 F + F/2 + 1 decisions (blake2b over the row's packed bits): what a
        fixture with 18 432 samples keeps instead of 5 MB of bits.  Equal digests <=> the sample's decisions agree."""
        import hashlib
        bits, shape = self.packed()
        rows = np.packbits(np.unpackbits(bits)[:shape[0] * shape[1]].reshape(tuple(shape)), axis=1)
        return np.array([int.from_bytes(hashlib.blake2b(r.tobytes(), digest_size=8).digest(), "little") for r in rows],
                        np.uint64)


# ---------------------------------------------------------------- F5 MLP
def f5_mlp():
    rng = np.random.RandomState(17)
    M = 256
    pts = rng.uniform(-4, 4, (M, 3)).astype(np.float32)
    dirs = rng.uniform(-1, 1, (M, 3)).astype(np.float32)
    out = dict(pts=pts, dirs=dirs)
    pe = RefPE(3, 10, True).encode(torch.from_numpy(pts))
    de = RefPE(3, 4, True).encode(torch.from_numpy(dirs))
    for tag, kw in (("default", dict(seed=1)), ("dense", dict(seed=2, sigma_bias=1.0, sigma_gain=30.0))):
        flat = synth.nerf_flat_params(**kw)
        net = load_ref_net(flat)
        signs = ReluSigns(net)
        sigma, rgb = net(pe, de)
        out[tag + "_relu_bits"], out[tag + "_relu_shape"] = signs.packed()
        g_sigma = torch.from_numpy(rng.standard_normal(M).astype(np.float32))
        g_rgb = torch.from_numpy(rng.standard_normal((M, 3)).astype(np.float32))
        (sigma * g_sigma).sum().add((rgb * g_rgb).sum()).backward()
        out[tag + "_sigma"] = sigma.detach().numpy(); out[tag + "_rgb"] = rgb.detach().numpy()
        out[tag + "_g_sigma"] = g_sigma.numpy(); out[tag + "_g_rgb"] = g_rgb.numpy()
        for k, v in grad_digest(net).items():
            out[tag + "_grad_" + k] = v
    save("f5_mlp", **out)


# ---------------------------------------------------------------- F6 compositing
def f6_composite():
    out = {}
    rng = np.random.RandomState(23)
    integ = ref_integrators.QuadratureIntegrator()
    for S in (64, 192, 7):
        n = 40
        sigma = rng.gamma(0.7, 3.0, (n, S)).astype(np.float32)
        sigma[0] = 0.0
        sigma[1] = 1e3
        sigma[2, ::2] = 0.0
        sigma[3] = 1e-6
        c = rng.rand(n, S, 3).astype(np.float32)
        t = np.sort(rng.uniform(2, 6, (n, S)).astype(np.float32), axis=1)
        delta = np.diff(np.concatenate([t, np.full((n, 1), 1e8, np.float32)], 1), axis=1).astype(np.float32)
        g_rgb = rng.standard_normal((n, 3)).astype(np.float32)
        g_w = rng.standard_normal((n, S)).astype(np.float32)
        st = torch.from_numpy(sigma).requires_grad_(True)
        ct = torch.from_numpy(c).requires_grad_(True)
        rgb, w = integ.integrate_along_rays(st, ct, torch.from_numpy(delta))
        (rgb * torch.from_numpy(g_rgb)).sum().backward()
        gs1, gc1 = st.grad.clone(), ct.grad.clone()
        st.grad = None; ct.grad = None
        rgb2, w2 = integ.integrate_along_rays(st, ct, torch.from_numpy(delta))
        ((rgb2 * torch.from_numpy(g_rgb)).sum() + (w2 * torch.from_numpy(g_w)).sum()).backward()
        p = f"S{S}_"
        out.update({p + "sigma": sigma, p + "c": c, p + "delta": delta, p + "g_rgb": g_rgb,
                    p + "g_w": g_w, p + "rgb": rgb.detach().numpy(), p + "w": w.detach().numpy(),
                    p + "gs": gs1.numpy(), p + "gc": gc1.numpy(), p + "gs_w": st.grad.numpy(),
                    p + "gc_w": ct.grad.numpy()})
    save("f6_composite", **out)


# ---------------------------------------------------------------- F7 end to end
def f7_e2e():
    out = {}
    n = 96
    cam, vr, pix, _ = rays_for(n, 31)
    flat_c = synth.nerf_flat_params(seed=3, sigma_bias=1.0, sigma_gain=30.0)
    flat_f = synth.nerf_flat_params(seed=4, sigma_bias=1.0, sigma_gain=30.0)
    enc = {"coord_enc": RefPE(3, 10, True), "dir_enc": RefPE(3, 4, True)}
    net_c, net_f = load_ref_net(flat_c), load_ref_net(flat_f)
    scene_c = ref_scene.PrimitiveCube(net_c, enc)
    scene_f = ref_scene.PrimitiveCube(net_f, enc)
    gt = synth.counter_uniform(77, 0, n * 3).reshape(n, 3)
    # the draws the two passes will consume, in order (coarse: U1c ; fine: U1, U2, U3)
    torch.manual_seed(2024)
    u1c = torch.rand((n, 64)); u1 = torch.rand((n, 64)); u2 = torch.rand((n, 128)); u3 = torch.rand((n, 128))
    torch.manual_seed(2024)
    pix_t = torch.from_numpy(pix)
    signs_c, signs_f = ReluSigns(net_c), ReluSigns(net_f)      # one query_points call per pass: ten ReLU calls per network
    c_rgb, c_idx, c_w = vr.render_scene(scene_c, n, 64, False, "cpu", pixel_indices=pix_t)
    c_w_before = c_w.detach().clone()
    f_rgb, f_idx, f_w = vr.render_scene(scene_f, n, (64, 128), False, "cpu", pixel_indices=c_idx,
                                        weights=c_w)
    mse = torch.nn.MSELoss()
    loss = mse(torch.from_numpy(gt), c_rgb) + mse(torch.from_numpy(gt), f_rgb)
    loss.backward()
    out.update(dict(pix=pix, pose=cam.extrinsic.numpy(), meta=np.array([800, 800, cam.focal_lengths[0], 2.0, 6.0]),
                    gt=gt, u1c=u1c.numpy(), u1=u1.numpy(), u2=u2.numpy(), u3=u3.numpy(),
                    coarse_rgb=c_rgb.detach().numpy(), coarse_w=c_w_before.numpy(),
                    coarse_w_after=c_w.detach().numpy(), fine_rgb=f_rgb.detach().numpy(),
                    fine_w=f_w.detach().numpy(), loss=np.array([loss.item()]),
                    idx_match=np.array([int(torch.equal(c_idx, pix_t) and torch.equal(f_idx, pix_t))])))
    for tag, net, signs in (("coarse", net_c, signs_c), ("fine", net_f, signs_f)):
        out[tag + "_relu_hash"] = signs.row_hashes()
        for k, v in grad_digest(net).items():
            out[tag + "_grad_" + k] = v
    save("f7_e2e", **out)


# ---------------------------------------------------------------- F8 optimizer step (row f1)
def f8_adam():
    """The optimizer exactly as runners/runner_utils.py:691-711 builds it and runners/train.py:215-218
    steps it.  runner_utils itself is not importable here (hydra/omegaconf absent), so the fixture makes
    the same two constructor calls on torch's own classes with the values of configs/train_params/nerf.yaml
    (num_iter shortened so that the decay is visible in 8 steps)."""
    rng = np.random.RandomState(41)
    init_lr, end_lr, num_iter, eps = 0.0005, 0.00005, 12, 1e-8
    shapes = [(37, 27), (4,)]                      # 1003 values: exercises the n % 4 tail
    n = sum(int(np.prod(s)) for s in shapes)
    p0 = rng.uniform(-0.2, 0.2, n).astype(np.float32)
    steps = 8
    grads = (rng.standard_normal((steps, n)) * np.exp(rng.uniform(-12, 2, (steps, n)))).astype(np.float32)
    grads[:, ::17] = 0.0                            # parameters that never receive gradient
    grads[3] = 0.0                                  # a step with an all-zero gradient
    params, off = [], 0
    for shp in shapes:
        k = int(np.prod(shp))
        params.append(torch.nn.Parameter(torch.from_numpy(p0[off:off + k].copy()).reshape(shp)))
        off += k
    optimizer = torch.optim.Adam(params, lr=init_lr, eps=eps)
    gamma = pow(end_lr / init_lr, 1 / num_iter)
    scheduler = torch.optim.lr_scheduler.ExponentialLR(optimizer, gamma)
    traj, lrs = [], []
    for s in range(steps):
        optimizer.zero_grad()
        off = 0
        for p in params:
            p.grad = torch.from_numpy(grads[s, off:off + p.numel()].copy()).reshape(p.shape)
            off += p.numel()
        lrs.append(optimizer.param_groups[0]["lr"])
        optimizer.step()
        scheduler.step()
        traj.append(np.concatenate([p.detach().numpy().reshape(-1) for p in params]))
    m = np.concatenate([optimizer.state[p]["exp_avg"].numpy().reshape(-1) for p in params])
    v = np.concatenate([optimizer.state[p]["exp_avg_sq"].numpy().reshape(-1) for p in params])
    save("f8_adam", p0=p0, grads=grads, params=np.stack(traj), exp_avg=m, exp_avg_sq=v,
         lrs=np.array(lrs, dtype=np.float64), config=np.array([init_lr, end_lr, num_iter, eps], dtype=np.float64),
         shapes=np.array([int(np.prod(s)) for s in shapes], dtype=np.int64))


# ---------------------------------------------------------------- F9 checkpoint layout (row f3)
def f9_checkpoint():
    """The dictionary runners/runner_utils.py:737-775 (_save_ckpt) writes, built with the same statements on the
    reference's own NeRF module (runner_utils itself needs hydra, which is absent here): only its STRUCTURE is
    recorded -- key names, shapes, dtypes -- which is what a checkpoint-compatible build has to reproduce."""
    enc = {"coord_enc": RefPE(3, 10, True), "dir_enc": RefPE(3, 4, True)}
    default_scene = ref_scene.PrimitiveCube(ref_nerf.NeRF(63, 27), enc)
    fine_scene = ref_scene.PrimitiveCube(ref_nerf.NeRF(63, 27), enc)
    params = list(default_scene.radiance_field.parameters()) + list(fine_scene.radiance_field.parameters())
    optimizer = torch.optim.Adam(params, lr=0.0005, eps=1e-8)
    scheduler = torch.optim.lr_scheduler.ExponentialLR(optimizer, pow(0.00005 / 0.0005, 1 / 300000))
    for p in params:
        p.grad = torch.ones_like(p)
    optimizer.step()
    scheduler.step()
    ckpt = {"epoch": 7, "optimizer_state_dict": optimizer.state_dict()}
    ckpt["scheduler_state_dict"] = scheduler.state_dict()
    ckpt["scene_default"] = default_scene.radiance_field.state_dict()
    ckpt["scene_fine"] = fine_scene.radiance_field.state_dict()
    sd = ckpt["scene_default"]
    opt = ckpt["optimizer_state_dict"]
    save("f9_checkpoint",
         top_keys=np.array(sorted(ckpt.keys())),
         scene_keys=np.array(list(sd.keys())),
         scene_shapes=np.array([list(v.shape) + [0] * (2 - v.ndim) for v in sd.values()], dtype=np.int64),
         scene_dtypes=np.array([str(v.dtype) for v in sd.values()]),
         optimizer_keys=np.array(sorted(opt.keys())),
         param_group_keys=np.array(sorted(opt["param_groups"][0].keys())),
         param_group_size=np.array([len(opt["param_groups"][0]["params"])]),
         state_keys=np.array(sorted(opt["state"][0].keys())),
         state_step=np.array([float(opt["state"][0]["step"])]),
         scheduler_keys=np.array(sorted(ckpt["scheduler_state_dict"].keys())),
         file_name=np.array(["ckpt_" + str(7).zfill(6) + ".pth"]))


# ---------------------------------------------------------------- F10 LLFF render poses (row f3)
def f10_llff_poses():
    """utils/data/load_llff.py:213-376,519-559 on a synthetic forward-facing pose set.  The module imports
    `imageio` (absent here) at its top for the image loader, which is out of scope: an empty stand-in module is
    placed in sys.modules for the duration of THIS import only; none of the functions captured touches it."""
    import types
    stub = "imageio" not in sys.modules
    if stub:
        sys.modules["imageio"] = types.ModuleType("imageio")
    try:
        import torch_nerf.src.utils.data.load_llff as ref_llff
    finally:
        if stub:
            del sys.modules["imageio"]
    assert ref_llff.__file__.startswith(REFERENCE)
    poses, z_bounds = synth.llff_like_pose_set(20, seed=0)
    recentred = ref_llff.recenter_poses(poses)
    avg = ref_llff.poses_avg(recentred)

    def spiral(extr, bds, zflat):   # the statements of load_llff_data between poses_avg and the fp32 cast (:519-559)
        c2w = ref_llff.poses_avg(extr)
        up = ref_llff.normalize(extr[:, :, 1].sum(0))
        close_depth, inf_depth = bds.min() * 0.9, bds.max() * 5.0
        dt = 0.75
        focal = 1.0 / (((1.0 - dt) / close_depth + dt / inf_depth))
        rads = np.percentile(np.abs(extr[:, :, 3]), 90, 0)
        n_key, n_rot = 120, 2
        if zflat:
            c2w[:3, 3] = c2w[:3, 3] + (-close_depth * 0.1) * c2w[:3, 2]
            rads[2] = 0.0
            n_rot, n_key = 1, 60
        return np.array(ref_llff.render_path_spiral(c2w, up, rads, focal, z_rate=0.5, rots=n_rot,
                                                    num_keyframe=n_key)).astype(np.float32)

    save("f10_llff_poses", poses=poses, z_bounds=z_bounds, recentred=recentred, poses_avg=avg,
         spiral=spiral(recentred, z_bounds, False), spiral_zflat=spiral(recentred, z_bounds, True),
         extrinsic_probe=ref_llff.build_extrinsic(np.array([0.1, -0.2, 0.9]), np.array([0.05, 1.0, 0.0]),
                                                  np.array([1.0, 2.0, 3.0])))


# ---------------------------------------------------------------- F11 network / encoder variants
NET_VARIANTS = {   # tag: (coord_encode_level, dir_encode_level, include_input, feat_dim) -- the knobs of
    # configs/signal_encoder/positional_encoding.yaml:2-4 and NeRF's feat_dim (network/nerf.py:27)
    "l6_l2": (6, 2, True, 256),
    "l4_l4": (4, 4, True, 256),
    "l10_l4_noinput": (10, 4, False, 256),
    "l10_l4_f128": (10, 4, True, 128),
    "l12_l6_f64": (12, 6, True, 64),      # pos_dim 75 > 64, view_dir_dim 39 > 32
}


def f11_net_variants():
    """NeRF(coord_enc.out_dim, dir_enc.out_dim[, feat_dim]) as runner_utils.py:584-612 builds it for other yaml
    values: outputs, parameter-gradient digests and the gradients autograd returns for the encoded inputs."""
    rng = np.random.RandomState(23)
    M = 96
    pts = rng.uniform(-3, 3, (M, 3)).astype(np.float32)
    dirs = rng.uniform(-1, 1, (M, 3)).astype(np.float32)
    g_sigma = rng.standard_normal(M).astype(np.float32)
    g_rgb = rng.standard_normal((M, 3)).astype(np.float32)
    out = dict(pts=pts, dirs=dirs, g_sigma=g_sigma, g_rgb=g_rgb)
    for tag, (lp, ld, inc, feat) in NET_VARIANTS.items():
        ce, de_ = RefPE(3, lp, inc), RefPE(3, ld, inc)
        pts_t = torch.from_numpy(pts.copy()).requires_grad_(True)     # leaves: autograd runs back through encode()
        dirs_t = torch.from_numpy(dirs.copy()).requires_grad_(True)
        pe, de = ce.encode(pts_t), de_.encode(dirs_t)
        pe.retain_grad(); de.retain_grad()
        flat = synth.nerf_flat_params(seed=5, pos_dim=ce.out_dim, view_dir_dim=de_.out_dim, feat_dim=feat,
                                      sigma_bias=0.5, sigma_gain=4.0)
        net = ref_nerf.NeRF(ce.out_dim, de_.out_dim, feat)
        net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in
                             synth.split_flat_params(flat, ce.out_dim, de_.out_dim, feat).items()})
        signs = ReluSigns(net)
        sigma, rgb = net(pe, de)
        out[tag + "_relu_bits"], out[tag + "_relu_shape"] = signs.packed()
        (sigma * torch.from_numpy(g_sigma)).sum().add((rgb * torch.from_numpy(g_rgb)).sum()).backward()
        out[tag + "_dims"] = np.array([ce.out_dim, de_.out_dim, feat, lp, ld, int(inc)])
        out[tag + "_pe"] = pe.detach().numpy(); out[tag + "_de"] = de.detach().numpy()
        out[tag + "_sigma"] = sigma.detach().numpy(); out[tag + "_rgb"] = rgb.detach().numpy()
        out[tag + "_g_pe"] = pe.grad.numpy(); out[tag + "_g_de"] = de.grad.numpy()
        out[tag + "_g_pts"] = pts_t.grad.numpy(); out[tag + "_g_dirs"] = dirs_t.grad.numpy()
        for k, v in grad_digest(net).items():
            out[tag + "_grad_" + k] = v
    save("f11_net_variants", **out)


# ---------------------------------------------------------------- F12 SH encoder
class _CpuSignal(torch.Tensor):
    """SHEncoder.encode allocates with device=in_signal.get_device(), which is -1 (an invalid index) for a CPU
    tensor: the reference's encoder only runs on CUDA tensors.  This subclass answers get_device() with the CPU
    device so that the reference's OWN statements run unmodified here."""
    def get_device(self):
        return torch.device("cpu")


def f12_sh_encoder():
    """signal_encoder: sh (configs/signal_encoder/sh.yaml, runner_utils.py:595-604): SHEncoder(3, degree) on points and
    directions, NeRF(degree^2, degree^2) behind it; encodings, outputs, parameter-gradient digests and the gradients
    w.r.t. the raw inputs for degree 4 (the shipped yaml), encodings + input gradients for degrees 1..5."""
    from torch_nerf.src.signal_encoder.spherical_harmonics_encoder import SHEncoder as RefSH
    rng = np.random.RandomState(31)
    M = 96
    pts = rng.uniform(-2, 2, (M, 3)).astype(np.float32)
    dirs = rng.uniform(-1, 1, (M, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    out = dict(pts=pts, dirs=dirs)
    for degree in (1, 2, 3, 4, 5):
        enc = RefSH(3, degree)
        x = torch.from_numpy(pts.copy()).as_subclass(_CpuSignal).requires_grad_(True)
        e = enc.encode(x)
        g_e = torch.from_numpy(rng.standard_normal((M, degree * degree)).astype(np.float32))
        if e.requires_grad:
            (e * g_e).sum().backward()
        out[f"d{degree}_enc"] = e.detach().numpy(); out[f"d{degree}_g_enc"] = g_e.numpy()
        out[f"d{degree}_g_pts"] = np.zeros_like(pts) if x.grad is None else np.asarray(x.grad)   # degree 1: a constant
    enc = RefSH(3, 4)
    x = torch.from_numpy(pts.copy()).as_subclass(_CpuSignal).requires_grad_(True)
    v = torch.from_numpy(dirs.copy()).as_subclass(_CpuSignal).requires_grad_(True)
    flat = synth.nerf_flat_params(seed=6, pos_dim=16, view_dir_dim=16, sigma_bias=0.5, sigma_gain=4.0)
    net = ref_nerf.NeRF(16, 16)
    net.load_state_dict({k: torch.from_numpy(a.copy()) for k, a in synth.split_flat_params(flat, 16, 16, 256).items()})
    signs = ReluSigns(net)
    sigma, rgb = net(enc.encode(x).as_subclass(torch.Tensor), enc.encode(v).as_subclass(torch.Tensor))
    out["net_relu_bits"], out["net_relu_shape"] = signs.packed()
    g_sigma = rng.standard_normal(M).astype(np.float32); g_rgb = rng.standard_normal((M, 3)).astype(np.float32)
    (sigma * torch.from_numpy(g_sigma)).sum().add((rgb * torch.from_numpy(g_rgb)).sum()).backward()
    out.update(net_sigma=sigma.detach().numpy(), net_rgb=rgb.detach().numpy(), net_g_sigma=g_sigma, net_g_rgb=g_rgb,
               net_g_pts=np.asarray(x.grad), net_g_dirs=np.asarray(v.grad))
    for k, a in grad_digest(net).items():
        out["net_grad_" + k] = a
    save("f12_sh_encoder", **out)


# ---------------------------------------------------------------- F13 Instant-NGP: does the reference's own module run?
def f13_instant_ngp():
    """network: instant_nerf (configs/network/instant_nerf.yaml, runner_utils.py:617-626).  The module constructs, but
    its forward cannot complete on ANY device: spatial_hash_func builds torch.tensor([[1, 2654435761, 805459861]],
    dtype=torch.int32) (instant_ngp.py:553-557) and 2654435761 does not fit int32.  Recorded as a fixture so that
    the claim 'there is no behaviour to be a drop-in for' is a pinned observation, not prose."""
    import json
    import torch_nerf.src.network.instant_ngp as ref_ngp
    rec = {"reference_file": "torch_nerf/src/network/instant_ngp.py", "torch": torch.__version__}
    net = ref_ngp.InstantNeRF(3, 16, 16, 19, 16, 512, table_feat_dim=2)
    rec["constructs"] = True
    rec["num_parameters"] = sum(p.numel() for p in net.parameters())
    try:
        net(torch.rand(8, 3).as_subclass(_CpuSignal), torch.rand(8, 16).as_subclass(_CpuSignal))
        rec["forward_runs"] = True
    except Exception as exc:  # noqa: BLE001
        rec["forward_runs"] = False
        rec["forward_error"] = f"{type(exc).__name__}: {exc}"
    try:
        ref_ngp.spatial_hash_func(torch.zeros((4, 3), dtype=torch.int32).as_subclass(_CpuSignal), 1 << 19)
        rec["spatial_hash_func_runs"] = True
    except Exception as exc:  # noqa: BLE001
        rec["spatial_hash_func_runs"] = False
        rec["spatial_hash_func_error"] = f"{type(exc).__name__}: {exc}"
    path = os.path.join(HERE, "f13_instant_ngp.json")
    json.dump(rec, open(path, "w"), indent=1)
    print(f"wrote {path}: {rec}")


# ---------------------------------------------------------------- F14 the training loop, 20 iterations
F14 = dict(n=96, steps=20, init_lr=0.0005, end_lr=0.00005, num_iter=40, eps=1e-8, keep=(0, 9, 19))


def f14_inputs(step, n=96):
    """Everything iteration `step` consumes, as pure functions of the step (shared with tests/test_gpu_trajectory.py
    through this module's twin in tests/helpers.py): camera pose, pixel batch, ground-truth colours and the four
    uniform tensors the two passes draw, in the order they draw them (coarse: U1c; fine: U1, U2, U3)."""
    pose = synth.pose_spherical(-180.0 + 18.0 * step, -30.0, 4.0)
    pix = synth.pixel_batch(140 + step, 800, 800, n)
    gt = synth.counter_uniform(78, step, n * 3).reshape(n, 3)
    draws = [synth.counter_uniform(500 + step, k, n * s).reshape(n, s) for k, s in enumerate((64, 64, 128, 128))]
    return pose, pix, gt, draws


class _ReplayDraws:
    """Stands in for torch.rand / torch.rand_like while the reference runs one iteration: hands out the prepared
    uniforms in call order and insists on the shapes (stratified_sampler.py:77,109; ray_samplers/utils.py:43,56)."""

    def __init__(self, draws):
        self.draws = [torch.from_numpy(d.copy()) for d in draws]

    def rand(self, *size, **kw):
        shape = tuple(size[0]) if len(size) == 1 and not isinstance(size[0], int) else tuple(size)
        d = self.draws.pop(0)
        assert tuple(d.shape) == shape, (d.shape, shape)
        return d

    def rand_like(self, t, **kw):
        return self.rand(tuple(t.shape))


def param_digest(prefix, now, start):
    """Digest of a parameter vector and of the update it received (now - start, the part 20 Adam steps wrote)."""
    d = {}
    for tag, v in (("p", now), ("dp", now - start)):
        d[f"{prefix}_{tag}.norm"] = np.array([np.sqrt(np.sum(v.astype(np.float64) ** 2))])
        d[f"{prefix}_{tag}.head"] = v[:GRAD_SLICE].copy()
        d[f"{prefix}_{tag}.stride"] = v[:: v.size // (8 * GRAD_SLICE)][: 8 * GRAD_SLICE].copy()
    return d


def f14_train_loop(name="f14_train_loop", levels=(10, 4)):
    """The body of runners/train.py:120-218, statement for statement, for 20 consecutive iterations on the reference's
    classes: new camera, coarse render_scene, MSE, fine render_scene on the coarse pass's (floored in place) weights,
    MSE, backward, Adam.step, ExponentialLR.step -- optimizer and scheduler built as runner_utils.py:691-711 builds
    them (num_iter shortened to 40 so that the decay shows within 20 steps).  What only shows ACROSS steps -- the
    optimizer's moments and step counts, parameters that changed under a cached weight image, the learning-rate
    schedule, the in-place weight floor feeding the next call -- is pinned by the per-step losses, the pixels of three
    iterations and digests of both networks' parameters after the last one.
    `levels` = (coord_encode_level, dir_encode_level) of configs/signal_encoder/positional_encoding.yaml:2-3: the shipped
    (10, 4) and, as f14_train_loop_l12_l5, (12, 5) -- NeRF(75, 33), which the build serves with other kernels (layered
    family: raw-point entry, register-resident forward with three position / two direction blocks, reg_dx_kernel)."""
    cfg = F14
    n, steps = cfg["n"], cfg["steps"]
    enc = {"coord_enc": RefPE(3, levels[0], True), "dir_enc": RefPE(3, levels[1], True)}
    e_p, e_d = enc["coord_enc"].out_dim, enc["dir_enc"].out_dim
    flat_c = synth.nerf_flat_params(seed=3, pos_dim=e_p, view_dir_dim=e_d, sigma_bias=1.0, sigma_gain=30.0)
    flat_f = synth.nerf_flat_params(seed=4, pos_dim=e_p, view_dir_dim=e_d, sigma_bias=1.0, sigma_gain=30.0)

    def load(flat):
        net = ref_nerf.NeRF(e_p, e_d)
        net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.split_flat_params(flat, e_p, e_d, 256).items()})
        return net

    net_c, net_f = load(flat_c), load(flat_f)
    default_scene, fine_scene = ref_scene.PrimitiveCube(net_c, enc), ref_scene.PrimitiveCube(net_f, enc)
    sampler = ref_samplers.StratifiedSampler()
    renderer = ref_vr.VolumeRenderer(ref_integrators.QuadratureIntegrator(), sampler,
                                     camera(800, 800, float(synth.blender_focal(800)), f14_inputs(0)[0], 2.0, 6.0))
    params = list(default_scene.radiance_field.parameters()) + list(fine_scene.radiance_field.parameters())
    optimizer = torch.optim.Adam(params, lr=cfg["init_lr"], eps=cfg["eps"])
    scheduler = torch.optim.lr_scheduler.ExponentialLR(optimizer, pow(cfg["end_lr"] / cfg["init_lr"], 1 / cfg["num_iter"]))
    loss_func = torch.nn.MSELoss()
    rec = dict(coarse_loss=[], fine_loss=[], loss=[], lr=[])
    out = {}
    real_rand, real_rand_like = torch.rand, torch.rand_like
    for step in range(steps):
        pose, pix, gt, draws = f14_inputs(step, n)
        pixel_gt = torch.from_numpy(gt)
        replay = _ReplayDraws(draws)
        torch.rand, torch.rand_like = replay.rand, replay.rand_like
        try:
            loss = 0.0
            optimizer.zero_grad()
            renderer.camera = camera(800, 800, float(synth.blender_focal(800)), pose, 2.0, 6.0)
            coarse_pred, coarse_indices, coarse_weights = renderer.render_scene(
                default_scene, num_pixels=n, num_samples=64, project_to_ndc=False,
                pixel_indices=torch.from_numpy(pix), device="cpu")
            coarse_loss = loss_func(pixel_gt, coarse_pred)
            loss += coarse_loss
            fine_pred, fine_indices, _ = renderer.render_scene(
                fine_scene, num_pixels=n, num_samples=(64, 128), project_to_ndc=False,
                pixel_indices=coarse_indices, weights=coarse_weights, device="cpu")
            fine_loss = loss_func(pixel_gt, fine_pred)
            loss += fine_loss
        finally:
            torch.rand, torch.rand_like = real_rand, real_rand_like
        assert not replay.draws, "the reference consumed fewer uniform tensors than prepared"
        rec["coarse_loss"].append(coarse_loss.item()); rec["fine_loss"].append(fine_loss.item())
        rec["loss"].append(loss.item()); rec["lr"].append(optimizer.param_groups[0]["lr"])
        loss.backward()
        optimizer.step()
        scheduler.step()
        if step in cfg["keep"]:
            out[f"s{step}_coarse_rgb"] = coarse_pred.detach().numpy()
            out[f"s{step}_fine_rgb"] = fine_pred.detach().numpy()
    for k, v in rec.items():
        out[k] = np.array(v, np.float64)
    for tag, net, flat in (("coarse", net_c, flat_c), ("fine", net_f, flat_f)):
        now = np.concatenate([p.detach().numpy().reshape(-1) for p in net.parameters()])
        out.update(param_digest(tag, now, flat))
    out["config"] = np.array([n, steps, cfg["init_lr"], cfg["end_lr"], cfg["num_iter"], cfg["eps"]], np.float64)
    out["keep"] = np.array(cfg["keep"], np.int64)
    out["levels"] = np.array(levels, np.int64)
    save(name, **out)


def f14_train_loop_l12_l5():
    f14_train_loop("f14_train_loop_l12_l5", (12, 5))


# ---------------------------------------------------------------- F15 what the runners touch on the hot-path classes
HOT_MODULES = ("torch_nerf.src.network", "torch_nerf.src.scene", "torch_nerf.src.renderer.cameras",
               "torch_nerf.src.renderer.integrators.quadrature_integrator", "torch_nerf.src.renderer.ray_samplers",
               "torch_nerf.src.renderer.volume_renderer", "torch_nerf.src.signal_encoder")
HOT_CLASS_FILES = {      # class -> the reference file that defines it (public methods / properties / attributes read off it)
    "VolumeRenderer": "torch_nerf/src/renderer/volume_renderer.py",
    "PerspectiveCamera": "torch_nerf/src/renderer/cameras.py",
    "StratifiedSampler": "torch_nerf/src/renderer/ray_samplers/stratified_sampler.py",
    "RaySamplerBase": "torch_nerf/src/renderer/ray_samplers/sampler_base.py",
    "QuadratureIntegrator": "torch_nerf/src/renderer/integrators/quadrature_integrator.py",
    "PrimitiveCube": "torch_nerf/src/scene/primitives/cube.py",
    "PrimitiveBase": "torch_nerf/src/scene/primitives/primitive_base.py",
    "NeRF": "torch_nerf/src/network/nerf.py",
    "PositionalEncoder": "torch_nerf/src/signal_encoder/positional_encoder.py",
}
RUNNER_FILES = ("torch_nerf/runners/runner_utils.py", "torch_nerf/runners/train.py", "torch_nerf/runners/render.py")


def f15_runner_surface():
    """The caller contract, read mechanically: an `ast` walk over the reference's three runner
    files records every attribute they take from the seven hot-path modules (`network.NeRF`, `scene.scene` in
    annotations, ...), the shape of every constructor call (positional count + keyword names), the shape of every call
    of a public method the hot-path classes define (`render_scene(..., pixel_indices=, weights=, num_ray_batch=)`, ...)
    and every member of those classes read or assigned on some object (`.radiance_field`, `.camera =`, ...).
    tests/test_runner_surface.py checks each record against the drop-in with `hasattr` / `inspect.signature`.  Names
    only -- no reference source text is stored."""
    import ast
    import json

    # members of the hot-path classes, from their own definitions
    methods, members = {}, {}
    for cls, rel in HOT_CLASS_FILES.items():
        tree = ast.parse(open(os.path.join(REFERENCE, rel)).read())
        node = next(n for n in ast.walk(tree) if isinstance(n, ast.ClassDef) and n.name == cls)
        for fn in node.body:
            if not isinstance(fn, ast.FunctionDef):
                continue
            is_prop = any((isinstance(d, ast.Name) and d.id == "property") or
                          (isinstance(d, ast.Attribute) and d.attr in ("setter", "getter")) for d in fn.decorator_list)
            if is_prop:
                members.setdefault(fn.name, set()).add(cls)
            elif not fn.name.startswith("_"):
                methods.setdefault(fn.name, set()).add(cls)
            for sub in ast.walk(fn):        # instance attributes: self.x = ...
                if isinstance(sub, ast.Attribute) and isinstance(sub.value, ast.Name) and sub.value.id == "self" and \
                        isinstance(sub.ctx, ast.Store) and not sub.attr.startswith("_"):
                    members.setdefault(sub.attr, set()).add(cls)
    for generic in ("parameters", "state_dict", "load_state_dict", "to", "train", "eval"):    # nn.Module surface used on scenes
        methods.setdefault(generic, set())

    module_attrs, ctor_calls, method_calls, member_uses = {}, [], [], []
    for rel in RUNNER_FILES:
        tree = ast.parse(open(os.path.join(REFERENCE, rel)).read())
        alias_mod, alias_cls = {}, {}
        for n in ast.walk(tree):
            if isinstance(n, ast.Import):
                for a in n.names:
                    if a.name in HOT_MODULES:
                        alias_mod[a.asname or a.name.split(".")[-1]] = a.name
            elif isinstance(n, ast.ImportFrom) and n.module in HOT_MODULES:
                for a in n.names:
                    alias_cls[a.asname or a.name] = (n.module, a.name)
                    module_attrs.setdefault(n.module, set()).add(a.name)
        for n in ast.walk(tree):
            if isinstance(n, ast.Attribute) and isinstance(n.value, ast.Name) and n.value.id in alias_mod:
                module_attrs.setdefault(alias_mod[n.value.id], set()).add(n.attr)
            if isinstance(n, ast.Call):
                shape = {"file": rel, "line": n.lineno, "nargs": len(n.args),
                         "keywords": sorted(k.arg for k in n.keywords if k.arg is not None)}
                f = n.func
                if isinstance(f, ast.Attribute) and isinstance(f.value, ast.Name) and f.value.id in alias_mod:
                    ctor_calls.append(dict(shape, callee=alias_mod[f.value.id] + "." + f.attr))
                elif isinstance(f, ast.Name) and f.id in alias_cls:
                    ctor_calls.append(dict(shape, callee=".".join(alias_cls[f.id])))
                elif isinstance(f, ast.Attribute) and f.attr in methods and \
                        not (isinstance(f.value, ast.Name) and f.value.id in ("torch", "os", "self")):
                    method_calls.append(dict(shape, method=f.attr, defined_by=sorted(methods[f.attr])))
            if isinstance(n, ast.Attribute) and n.attr in members and \
                    not (isinstance(n.value, ast.Name) and n.value.id in alias_mod):
                member_uses.append({"file": rel, "line": n.lineno, "member": n.attr,
                                    "store": isinstance(n.ctx, ast.Store), "defined_by": sorted(members[n.attr])})
    rec = {"runner_files": list(RUNNER_FILES),
           "module_attrs": {m: sorted(v) for m, v in sorted(module_attrs.items())},
           "ctor_calls": sorted(ctor_calls, key=lambda c: (c["file"], c["line"], c["callee"])),
           "method_calls": sorted(method_calls, key=lambda c: (c["file"], c["line"], c["method"])),
           "member_uses": sorted(member_uses, key=lambda c: (c["file"], c["line"], c["member"], c["store"]))}
    path = os.path.join(HERE, "f15_runner_surface.json")
    json.dump(rec, open(path, "w"), indent=1, sort_keys=True)
    print(f"wrote {path}: {sum(len(v) for v in rec['module_attrs'].values())} module attributes, {len(ctor_calls)} constructor "
          f"calls, {len(method_calls)} method calls, {len(member_uses)} member uses")


if __name__ == "__main__":
    torch.set_num_threads(8)
    every = dict(f15=f15_runner_surface, f14=f14_train_loop, f14b=f14_train_loop_l12_l5, f13=f13_instant_ngp, f12=f12_sh_encoder, f11=f11_net_variants, f10=f10_llff_poses, f9=f9_checkpoint, f1=f1_raygen, f2=f2_coarse, f3=f3_fine, f4=f4_posenc, f5=f5_mlp, f6=f6_composite, f7=f7_e2e,
                 f8=f8_adam)
    for name in (sys.argv[1:] or list(every)):      # e.g. `make_golden.py f8` rewrites one fixture
        every[name]()
