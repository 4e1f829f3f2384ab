"""BASELINE.json configurations other than the bench workload, as parity cases on the GPU, plus
size-independent properties at the full 4096 x (64+128) batch size."""
import numpy as np
import pytest
import torch

import torch_nerf.src.network as network
import torch_nerf.src.scene as scene
import torch_nerf.src.renderer.cameras as cameras
import torch_nerf.src.renderer.integrators.quadrature_integrator as integrators
import torch_nerf.src.renderer.ray_samplers as ray_samplers
from torch_nerf.src.renderer.volume_renderer import VolumeRenderer
from torch_nerf.src.signal_encoder import PositionalEncoder
from torch_nerf.amd import ops, shard, synth

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def nets():
    out = []
    for seed in (3, 4):
        flat = synth.nerf_flat_params(seed=seed, sigma_bias=1.0, sigma_gain=30.0)
        net = network.NeRF(63, 27)
        net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.split_flat_params(flat).items()})
        out.append((net.cuda(), flat))
    return out


def oracle_two_pass(oracle, flats, o, d, near, far, draws):
    t_bins = torch.linspace(near, far, 65)[:-1].numpy()
    ps = (far - near) / 64
    c = oracle.render_rays(flats[0], o, d, t_bins, ps, draws[0])
    f = oracle.render_rays(flats[1], o, d, t_bins, ps, draws[1], weights=c["weights"], u2=draws[2], u3=draws[3])
    return c, f


def test_llff_ndc_forward_facing(oracle):
    """configs[3]: LLFF fern geometry 1008x756, NDC rays, t in [0,1] (runner_utils.py:489-491)."""
    H, W, focal, near, far = 756, 1008, 815.0, 0.0, 1.0
    pose = synth.llff_like_pose()
    n = 96
    pix = synth.pixel_batch(5, H, W, n)
    draws = [d.numpy() for d in shard.ray_draws(3, 0, n, 64, 128, "cpu")]
    (net_c, flat_c), (net_f, flat_f) = nets()
    k4 = (np.float32(focal), np.float32(focal), W / 2.0, H / 2.0)
    o, d = ops.generate_rays(H, W, k4, torch.from_numpy(pose), True, focal, near, "cuda", pix=dev(pix))
    oo, do = oracle.raygen(oracle.screen_coords(H, W, pix), *k4, pose)
    oo, do = oracle.map_rays_to_ndc(focal, near, H, W, oo, do)
    assert np.array_equal(o.cpu().numpy(), oo) and np.array_equal(d.cpu().numpy(), do)
    assert np.all(oo[:, 2] == 1.0)  # near = 0 under the reference's LLFF path: origin_z == 1
    c, f = oracle_two_pass(oracle, (flat_c, flat_f), oo, do, near, far, draws)
    t_bins = torch.linspace(near, far, 65)[:-1].cuda()
    ps = (far - near) / 64
    pc, pf = ops.mlp_pack(dev(flat_c)), ops.mlp_pack(dev(flat_f))
    c_rgb, c_w = ops.render_rays(pc, o, d, t_bins, ps, dev(draws[0]))
    np.testing.assert_allclose(c_rgb.cpu().numpy(), c["rgb"], rtol=0, atol=1e-5)
    f_rgb, f_w = ops.render_rays(pf, o, d, t_bins, ps, dev(draws[1]), weights=dev(c["weights"]), u2=dev(draws[2]),
                                 u3=dev(draws[3]))
    np.testing.assert_allclose(f_rgb.cpu().numpy(), f["rgb"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(f_w.cpu().numpy(), f["weights"], rtol=0, atol=1e-5)


def test_blender_400_coarse_only(oracle):
    """configs[0]: 400x400, coarse-only 64 samples (the reference's CPU-runnable case)."""
    H = W = 400
    focal = float(synth.blender_focal(W))
    pose = synth.pose_spherical(-153.0, -30.0, 4.0)
    n = 128
    pix = synth.pixel_batch(8, H, W, n)
    u1 = shard.ray_draws(4, 0, n, 64, 128, "cpu")[0].numpy()
    (net_c, flat_c), _ = nets()
    k4 = (np.float32(focal), np.float32(focal), W / 2.0, H / 2.0)
    oo, do = oracle.raygen(oracle.screen_coords(H, W, pix), *k4, pose)
    t_bins = torch.linspace(2.0, 6.0, 65)[:-1]
    c = oracle.render_rays(flat_c, oo, do, t_bins.numpy(), 4.0 / 64, u1)
    o, d = ops.generate_rays(H, W, k4, torch.from_numpy(pose), False, focal, 2.0, "cuda", pix=dev(pix))
    rgb, w = ops.render_rays(ops.mlp_pack(dev(flat_c)), o, d, t_bins.cuda(), 4.0 / 64, dev(u1))
    np.testing.assert_allclose(rgb.cpu().numpy(), c["rgb"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(w.cpu().numpy(), c["weights"], rtol=0, atol=1e-5)


def test_full_batch_properties():
    """4096 rays x (64+128): properties that do not need the oracle at this size."""
    H = W = 800
    focal = float(synth.blender_focal(W))
    cam = cameras.PerspectiveCamera({"f_x": focal, "f_y": focal, "img_width": W, "img_height": H},
                                    torch.from_numpy(synth.pose_spherical(37.0, -30.0, 4.0)), 2.0, 6.0)
    enc = {"coord_enc": PositionalEncoder(3, 10, True), "dir_enc": PositionalEncoder(3, 4, True)}
    (net_c, _), (net_f, _) = nets()
    vr = VolumeRenderer(integrators.QuadratureIntegrator(), ray_samplers.StratifiedSampler(), cam)
    pix = torch.from_numpy(synth.pixel_batch(0, H, W, 4096))
    di = torch.cuda.current_device()
    with torch.no_grad():
        torch.manual_seed(0)
        c_rgb, idx, c_w = vr.render_scene(scene.PrimitiveCube(net_c, enc), 4096, 64, False, di, pixel_indices=pix)
        w_before = c_w.clone()
        f_rgb, idx2, f_w = vr.render_scene(scene.PrimitiveCube(net_f, enc), 4096, (64, 128), False, di,
                                           pixel_indices=idx, weights=c_w)
        # determinism: same seed, same pixels -> identical bits
        torch.manual_seed(0)
        c_rgb2, _, c_w2 = vr.render_scene(scene.PrimitiveCube(net_c, enc), 4096, 64, False, di, pixel_indices=pix)
    assert torch.equal(idx, pix) and torch.equal(idx2, pix)
    assert torch.equal(c_rgb, c_rgb2) and torch.equal(w_before, c_w2)
    assert torch.equal(c_w, w_before + 1e-5)                      # in-place floor, exactly one fp32 add
    for w in (w_before, f_w):
        assert (w >= 0).all() and (w.sum(-1) <= 1 + 1e-5).all()   # weights are a sub-probability vector
    for rgb in (c_rgb, f_rgb):
        assert torch.isfinite(rgb).all() and (rgb >= 0).all() and (rgb <= 1 + 1e-5).all()
    # the sampler's outputs at full size: sorted t (non-negative delta), last delta ~ 1e8
    s = ray_samplers.StratifiedSampler()
    bundle = s.generate_rays_from_pixels(cam, False, pixel_indices=pix)
    pts, dirs, delta = s.sample_along_rays(bundle, (64, 128), di, weights=w_before.clone())
    assert pts.shape == (4096, 192, 3) and (delta >= 0).all() and (delta[:, -1] > 9e7).all()
    assert torch.equal(dirs[:, 0], bundle.ray_dir) and torch.equal(dirs[:, 191], bundle.ray_dir)
    # linearity of the integrator in the radiance: C(2c) = 2 C(c)
    sigma = torch.rand(4096, 192, device="cuda") * 3
    c = torch.rand(4096, 192, 3, device="cuda")
    r1, _ = ops.composite_forward(sigma, c, delta)
    r2, _ = ops.composite_forward(sigma, 2 * c, delta)
    np.testing.assert_allclose(r2.cpu().numpy(), 2 * r1.cpu().numpy(), rtol=1e-6, atol=1e-6)


def test_sharded_frame_equals_single_launch():
    """shard.render_frame on one GPU == the same frame rendered in one launch with the same draws,
    and does not depend on the launch granularity (hence not on the GPU count)."""
    H, W = 48, 64
    focal = 70.0
    cam = cameras.PerspectiveCamera({"f_x": focal, "f_y": focal, "img_width": W, "img_height": H},
                                    torch.from_numpy(synth.pose_spherical(10.0, -30.0, 4.0)), 2.0, 6.0)
    (net_c, _), (net_f, _) = nets()
    a = shard.render_frame(cam, net_c, net_f, 64, 128, False, seed=5, rays_per_launch=H * W)
    b = shard.render_frame(cam, net_c, net_f, 64, 128, False, seed=5, rays_per_launch=1000)
    assert a.shape == (H * W, 3) and torch.equal(a, b)
    # a two-way split rendered rank by rank and concatenated is the same image
    parts = []
    for r in range(2):
        lo, hi = shard.shard_range(H * W, r, 2)
        s = ray_samplers.StratifiedSampler()
        bundle = s.generate_rays_from_pixels(cam, False, first=lo, count=hi - lo)
        t_bins, ps = s._create_t_bins(2.0, 6.0, 64, "cuda")
        u1c, u1, u2, u3 = shard.ray_draws(5, lo, hi - lo, 64, 128, "cuda")
        _, w = ops.render_rays(net_c._stream()[2], bundle.ray_origin, bundle.ray_dir, t_bins, ps, u1c)
        rgb, _ = ops.render_rays(net_f._stream()[2], bundle.ray_origin, bundle.ray_dir, t_bins, ps, u1, weights=w,
                                 u2=u2, u3=u3)
        parts.append(rgb)
    assert torch.equal(torch.cat(parts), a)


# ---------------------------------------------------------------- frame-size properties
def _frame_setup(kind):
    (net_c, _), (net_f, _) = nets()
    if kind == "llff":          # configs[3]: 1008 x 756, NDC rays, t in [0, 1]
        H, W, focal, ndc, near, far = 756, 1008, 815.0, True, 0.0, 1.0
        pose = synth.llff_like_pose()
    else:                       # configs[0]: 400 x 400 Blender geometry
        H = W = 400
        focal, ndc, near, far = float(synth.blender_focal(W)), False, 2.0, 6.0
        pose = synth.pose_spherical(37.0, -30.0, 4.0)
    cam = cameras.PerspectiveCamera({"f_x": focal, "f_y": focal, "img_width": W, "img_height": H},
                                    torch.from_numpy(pose), near, far)
    return net_c, net_f, cam, ndc, H, W


@pytest.mark.parametrize("kind,n_fine", [("llff", 128), ("coarse400", 0)])
def test_full_frame_properties(kind, n_fine):
    """Whole frames of BASELINE configs[3] (LLFF fern geometry, NDC, 64+128) and configs[0] (400x400, coarse-only 64)
    through shard.render_frame: every pixel finite and inside [0, 1] (weights are a sub-probability, colours a
    sigmoid), and the image does not depend on how the frame is cut into launches -- bit for bit (the draws are a
    function of the global ray index, and the kernels have no cross-ray state)."""
    net_c, net_f, cam, ndc, H, W = _frame_setup(kind)
    img = shard.render_frame(cam, net_c, net_f, 64, n_fine, ndc, seed=11, single_rank=True)
    assert img.shape == (H * W, 3) and bool(torch.isfinite(img).all())
    assert float(img.min()) >= 0.0 and float(img.max()) <= 1.0 + 1e-6
    assert float(img.std()) > 1e-3                                    # not a constant image
    for rays_per_launch in (50000, 4096 * 7 + 4):
        again = shard.render_frame(cam, net_c, net_f, 64, n_fine, ndc, seed=11, single_rank=True,
                                   rays_per_launch=rays_per_launch)
        assert torch.equal(img, again), rays_per_launch


@pytest.mark.parametrize("kind,fine", [("llff", True), ("coarse400", False)])
def test_frame_rows_sorted_and_subprobability(kind, fine):
    """Per-ray properties at frame scale, through the fused pass with its optional outputs: sorted sample positions
    (t ascending, inside [t_near, t_far + one bin]), weights in [0, 1] with sum <= 1 (transmittance never grows),
    bin indices inside [0, 63]; pixel colour = sum_i w_i c_i is therefore inside [0, 1].  One image row band of
    65 536 rays per configuration."""
    net_c, net_f, cam, ndc, H, W = _frame_setup(kind)
    n, Sc, Sf = 65536, 64, 128
    first = (H * W - n) // 2
    sampler = ray_samplers.StratifiedSampler()
    bundle = sampler.generate_rays_from_pixels(cam, ndc, first=first, count=n, device=torch.device("cuda"))
    t_bins, ps = sampler._create_t_bins(cam.t_near, cam.t_far, Sc, torch.device("cuda"))
    u1c, u1, u2, u3 = shard.ray_draws(5, first, n, Sc, Sf, torch.device("cuda"))
    _, _, packed_c = net_c._stream()
    _, _, packed_f = net_f._stream()
    rgb, w, t = ops.render_rays(packed_c, bundle.ray_origin, bundle.ray_dir, t_bins, ps, u1c, want_t=True)
    if fine:
        rgb, w, idx, t = ops.render_rays(packed_f, bundle.ray_origin, bundle.ray_dir, t_bins, ps, u1, weights=w, u2=u2,
                                         u3=u3, want_idx=True, want_t=True)
        assert int(idx.min()) >= 0 and int(idx.max()) <= Sc - 1
    assert bool((t[:, 1:] >= t[:, :-1]).all()), "sample positions must be sorted along every ray"
    assert float(t.min()) >= cam.t_near and float(t.max()) <= cam.t_far
    assert bool(torch.isfinite(w).all()) and float(w.min()) >= 0.0 and float(w.max()) <= 1.0
    assert float(w.sum(dim=1).max()) <= 1.0 + 1e-5
    assert float(rgb.min()) >= 0.0 and float(rgb.max()) <= 1.0 + 1e-5
