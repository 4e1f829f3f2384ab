"""Parity of the HIP kernels (through the C ABI) against the CPU oracle and the golden
vectors captured from the reference.  All of these need a real MI355X.

Bars (north_star): bit-exact for pixel / sample indices and for everything built from
IEEE + - * / only; <= 1e-5 abs fp32 for sigma, rgb and pixel colours.
"""
import numpy as np
import pytest
import torch

from torch_nerf.amd import synth

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


@pytest.fixture(scope="module")
def ops():
    from torch_nerf.amd import ops as _ops
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return _ops


def test_rays_against_golden(golden, oracle, ops):
    g = golden("f1_raygen")
    for name in ("blender", "blender400", "llff_ndc0", "llff_ndc1"):
        H, W, focal, near, far, ndc = g[name + "_meta"]
        H, W, ndc = int(H), int(W), bool(ndc)
        K = g[name + "_intrinsic"]
        pix = g[name + "_pix"]
        k4 = (K[0, 0], K[1, 1], K[0, 2], K[1, 2])
        pose = torch.from_numpy(g[name + "_pose"])
        coords = ops.screen_coords(H, W, "cuda", pix=dev(pix))
        assert np.array_equal(coords.cpu().numpy(), g[name + "_coords"])
        o1, d1 = ops.generate_rays(H, W, k4, pose, ndc, focal, near, "cuda", coords=coords)
        o2, d2 = ops.generate_rays(H, W, k4, pose, ndc, focal, near, "cuda", pix=dev(pix))
        assert torch.equal(o1, o2) and torch.equal(d1, d2)
        # bit-exact against the oracle (same op order), tolerance against ATen's sgemm
        co = oracle.screen_coords(H, W, pix)
        oo, do = oracle.raygen(co, *k4, g[name + "_pose"])
        if ndc:
            oo, do = oracle.map_rays_to_ndc(float(focal), float(near), H, W, oo, do)
        assert np.array_equal(_bits(o1.cpu().numpy()), _bits(oo)), name
        assert np.array_equal(_bits(d1.cpu().numpy()), _bits(do)), name
        np.testing.assert_allclose(d1.cpu().numpy(), g[name + "_d"], rtol=3e-6, atol=1e-6)
        np.testing.assert_allclose(o1.cpu().numpy(), g[name + "_o"], rtol=3e-6, atol=1e-6)
    # whole-frame mode (no index tensor): first + i
    full = ops.screen_coords(6, 5, "cuda")
    assert np.array_equal(full.cpu().numpy(), g["small_coords_6x5"])


@pytest.mark.parametrize("name", ["b", "ndc", "odd"])
def test_stratified_bit_exact(golden, ops, name):
    g = golden("f2_coarse")
    pts, dirs, delta = ops.sample_stratified(dev(g[name + "_o"]), dev(g[name + "_d"]), dev(g[name + "_t_bins"]),
                                             float(g[name + "_ps"][0]), dev(g[name + "_u1"]))
    assert np.array_equal(_bits(delta.cpu().numpy()), _bits(g[name + "_delta"]))
    assert np.array_equal(_bits(pts.cpu().numpy()), _bits(g[name + "_pts"]))
    assert np.array_equal(_bits(dirs.cpu().numpy()), _bits(g[name + "_dirs"]))


@pytest.mark.parametrize("name", ["b", "ndc", "s128", "s40", "s1000"])
def test_hierarchical_bit_exact(golden, ops, name):
    g = golden("f3_fine")
    w = dev(g[name + "_w_in"])
    pts, dirs, delta, idx, t = ops.sample_hierarchical(
        dev(g[name + "_o"]), dev(g[name + "_d"]), dev(g[name + "_t_bins"]), float(g[name + "_ps"][0]), w,
        dev(g[name + "_u1"]), dev(g[name + "_u2"]), dev(g[name + "_u3"]), want_idx=True, want_t=True)
    assert np.array_equal(_bits(w.cpu().numpy()), _bits(g[name + "_w_after"])), "in-place +1e-5"
    assert np.array_equal(idx.cpu().numpy(), g[name + "_idx"].astype(np.int64)), "bin indices"
    assert np.array_equal(_bits(t.cpu().numpy()), _bits(g[name + "_t"])), "sorted t"
    assert np.array_equal(_bits(delta.cpu().numpy()), _bits(g[name + "_delta"]))
    assert np.array_equal(_bits(pts.cpu().numpy()), _bits(g[name + "_pts"]))


def test_hierarchical_random_vs_oracle(oracle, ops):
    """Seeded random rays/weights at the bench shape (subset): indices bit-exact vs the oracle."""
    rng = np.random.RandomState(5)
    n, Sc, Sf = 512, 64, 128
    o = rng.uniform(-4, 4, (n, 3)).astype(np.float32)
    d = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    w = (rng.rand(n, Sc) ** 6).astype(np.float32)
    u1, u2, u3 = (rng.rand(n, s).astype(np.float32) for s in (Sc, Sf, Sf))
    t_bins = torch.linspace(2.0, 6.0, Sc + 1)[:-1].numpy()
    ps = 4.0 / Sc
    idx_o, t_o, pts_o, _, delta_o, w_o = oracle.hierarchical_sample(o, d, t_bins, ps, w, u1, u2, u3)
    wt = dev(w)
    pts, dirs, delta, idx, t = ops.sample_hierarchical(dev(o), dev(d), dev(t_bins), ps, wt, dev(u1), dev(u2),
                                                       dev(u3), want_idx=True, want_t=True)
    assert np.array_equal(idx.cpu().numpy(), idx_o)
    assert np.array_equal(_bits(t.cpu().numpy()), _bits(t_o))
    assert np.array_equal(_bits(pts.cpu().numpy()), _bits(pts_o))
    assert np.array_equal(_bits(delta.cpu().numpy()), _bits(delta_o))
    assert np.array_equal(_bits(wt.cpu().numpy()), _bits(w_o))
    assert np.all(np.diff(t.cpu().numpy(), axis=1) >= 0), "sortedness"


def test_posenc(golden, ops):
    g = golden("f4_posenc")
    x = dev(g["x"])
    # sin/cos implementations differ by a few ulp; values are in [-1, 1]
    np.testing.assert_allclose(ops.posenc(x, 10, True).cpu().numpy(), g["pe10"], rtol=0, atol=5e-7)
    np.testing.assert_allclose(ops.posenc(x, 4, True).cpu().numpy(), g["pe4"], rtol=0, atol=5e-7)
    np.testing.assert_allclose(ops.posenc(x, 4, False).cpu().numpy(), g["pe4_noinput"], rtol=0, atol=5e-7)


@pytest.mark.parametrize("S", [64, 192, 7])
def test_composite(golden, ops, S):
    g = golden("f6_composite")
    p = f"S{S}_"
    sigma, c, delta = dev(g[p + "sigma"]), dev(g[p + "c"]), dev(g[p + "delta"])
    rgb, w = ops.composite_forward(sigma, c, delta)
    np.testing.assert_allclose(w.cpu().numpy(), g[p + "w"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(rgb.cpu().numpy(), g[p + "rgb"], rtol=0, atol=1e-5)
    gs, gc = ops.composite_backward(sigma, c, delta, dev(g[p + "g_rgb"]))
    np.testing.assert_allclose(gc.cpu().numpy(), g[p + "gc"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(gs.cpu().numpy(), g[p + "gs"], rtol=1e-4, atol=1e-4)
    gs, gc = ops.composite_backward(sigma, c, delta, dev(g[p + "g_rgb"]), dev(g[p + "g_w"]))
    np.testing.assert_allclose(gc.cpu().numpy(), g[p + "gc_w"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(gs.cpu().numpy(), g[p + "gs_w"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("S", [64, 192, 7])
def test_composite_backward_vs_fp64_oracle(golden, ops, oracle, S):
    """The golden d sigma above is what the reference's fp32 autograd produced -- itself only good to ~1e-4 (its
    cumsum / exp chain cancels in fp32), hence the loose bound there.  The yardstick for the KERNEL is the oracle's
    reverse pass evaluated in double (nerf_oracle.c:orc_composite_backward): rtol 1e-5, plus 2e-6 of the ray's largest
    |d sigma| for the elements where T_{i+1} G_i - sum_{k>i} w_k G_k cancels (the kernel's exp() are fp32 like the
    reference's; its prefix / suffix sums are double)."""
    g = golden("f6_composite")
    p = f"S{S}_"
    sigma, c, delta = g[p + "sigma"], g[p + "c"], g[p + "delta"]
    for g_w in (None, g[p + "g_w"]):
        want_s, want_c = oracle.composite_backward(sigma, c, delta, g[p + "g_rgb"], g_w)
        gs, gc = ops.composite_backward(dev(sigma), dev(c), dev(delta), dev(g[p + "g_rgb"]), None if g_w is None else dev(g_w))
        scale = np.abs(want_s).max(axis=1, keepdims=True)
        err = np.abs(gs.cpu().numpy() - want_s)
        # (+ 1e-8 absolute: rays whose transmittance has underflowed in fp32 -- sigma = 1e3 in the first bins -- carry
        # gradients of 1e-13 .. 1e-9 that the double-precision oracle still resolves)
        bound = 1e-5 * np.abs(want_s) + 2e-6 * scale + 1e-8
        assert np.all(err <= bound), float((err / bound).max())
        np.testing.assert_allclose(gc.cpu().numpy(), want_c, rtol=1e-5, atol=1e-7)


def test_composite_autograd_function(golden, ops):
    g = golden("f6_composite")
    p = "S64_"
    sigma = dev(g[p + "sigma"]).requires_grad_(True)
    c = dev(g[p + "c"]).requires_grad_(True)
    rgb, w = ops.CompositeFunction.apply(sigma, c, dev(g[p + "delta"]))
    ((rgb * dev(g[p + "g_rgb"])).sum() + (w * dev(g[p + "g_w"])).sum()).backward()
    np.testing.assert_allclose(sigma.grad.cpu().numpy(), g[p + "gs_w"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(c.grad.cpu().numpy(), g[p + "gc_w"], rtol=1e-5, atol=1e-6)


MLP_KW = {"default": dict(seed=1), "dense": dict(seed=2, sigma_bias=1.0, sigma_gain=30.0)}


@pytest.mark.parametrize("tag", ["default", "dense"])
def test_mlp_forward_golden(golden, ops, tag):
    g = golden("f5_mlp")
    flat = synth.nerf_flat_params(**MLP_KW[tag])
    packed = ops.mlp_pack(dev(flat))
    # raw points: encoding happens inside the kernel
    sigma, rgb = ops.mlp_forward(packed, dev(g["pts"]), dev(g["dirs"]), encoded=False)
    np.testing.assert_allclose(sigma.cpu().numpy(), g[tag + "_sigma"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(rgb.cpu().numpy(), g[tag + "_rgb"], rtol=0, atol=1e-5)
    # pre-encoded inputs: plain NeRF.forward
    pe = ops.posenc(dev(g["pts"]), 10, True)
    de = ops.posenc(dev(g["dirs"]), 4, True)
    sigma2, rgb2 = ops.mlp_forward(packed, pe, de, encoded=True)
    np.testing.assert_allclose(sigma2.cpu().numpy(), g[tag + "_sigma"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(rgb2.cpu().numpy(), g[tag + "_rgb"], rtol=0, atol=1e-5)


@pytest.mark.parametrize("M", [1, 31, 128, 129, 1000, 40000])
def test_mlp_forward_ragged_vs_oracle(oracle, ops, M):
    """Tile tails, single sample, and more tiles than CUs; checked against the C oracle."""
    rng = np.random.RandomState(M)
    pts = rng.uniform(-4, 4, (M, 3)).astype(np.float32)
    dirs = rng.uniform(-1, 1, (M, 3)).astype(np.float32)
    flat = synth.nerf_flat_params(seed=9, sigma_bias=0.5, sigma_gain=20.0)
    packed = ops.mlp_pack(dev(flat))
    sigma, rgb = ops.mlp_forward(packed, dev(pts), dev(dirs), encoded=False)
    sub = slice(None) if M <= 1000 else rng.permutation(M)[:1500]
    so, ro = oracle.mlp_forward(flat, oracle.posenc(pts[sub], 10), oracle.posenc(dirs[sub], 4))
    np.testing.assert_allclose(sigma.cpu().numpy()[sub], so, rtol=0, atol=1e-5)
    np.testing.assert_allclose(rgb.cpu().numpy()[sub], ro, rtol=0, atol=1e-5)


def test_mlp_forward_huge_coordinates_take_the_exact_path(oracle, ops):
    """|2^9 x| beyond the Cody-Waite range: the kernel switches the whole tile to library sin/cos."""
    rng = np.random.RandomState(4)
    M = 300
    pts = rng.uniform(-2000, 2000, (M, 3)).astype(np.float32)
    pts[:150] = rng.uniform(-4, 4, (150, 3))          # tiles mixing ordinary and huge samples
    dirs = rng.uniform(-1, 1, (M, 3)).astype(np.float32)
    flat = synth.nerf_flat_params(seed=9, sigma_bias=0.5, sigma_gain=20.0)
    sigma, rgb = ops.mlp_forward(ops.mlp_pack(dev(flat)), dev(pts), dev(dirs), encoded=False)
    so, ro = oracle.mlp_forward(flat, oracle.posenc(pts, 10), oracle.posenc(dirs, 4))
    s, c = sigma.cpu().numpy(), rgb.cpu().numpy()
    # ordinary samples sharing a wavefront with huge ones went through the exact path: usual bound
    np.testing.assert_allclose(s[:150], so[:150], rtol=0, atol=1e-5)
    np.testing.assert_allclose(c[:150], ro[:150], rtol=0, atol=1e-5)
    # raw coordinates of 2e3 enter fc_in / fc_5 directly: activations are ~1e3x larger and so is the
    # fp32 summation-order noise; a wrong range reduction would be off by O(1), not O(1e-4)
    np.testing.assert_allclose(s[150:], so[150:], rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(c[150:], ro[150:], rtol=0, atol=2e-3)


def test_render_rays_end_to_end_golden(golden, ops):
    """Coarse + fine pass on the reference's own draws: pixel colours within 1e-5."""
    g = golden("f7_e2e")
    H, W, focal, near, far = g["meta"]
    H, W = int(H), int(W)
    pose = torch.from_numpy(g["pose"])
    o, d = ops.generate_rays(H, W, (np.float32(focal), np.float32(focal), W / 2.0, H / 2.0), pose, False,
                             focal, near, "cuda", pix=dev(g["pix"]))
    t_bins = torch.linspace(float(near), float(far), 65)[:-1].cuda()
    ps = (float(far) - float(near)) / 64
    pc = ops.mlp_pack(dev(synth.nerf_flat_params(seed=3, sigma_bias=1.0, sigma_gain=30.0)))
    pf = ops.mlp_pack(dev(synth.nerf_flat_params(seed=4, sigma_bias=1.0, sigma_gain=30.0)))
    c_rgb, c_w = ops.render_rays(pc, o, d, t_bins, ps, dev(g["u1c"]))
    np.testing.assert_allclose(c_rgb.cpu().numpy(), g["coarse_rgb"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(c_w.cpu().numpy(), g["coarse_w"], rtol=0, atol=1e-5)
    # fine pass seeded with the REFERENCE's coarse weights (bins then match it bit for bit)
    w_in = dev(g["coarse_w"])
    f_rgb, f_w = ops.render_rays(pf, o, d, t_bins, ps, dev(g["u1"]), weights=w_in, u2=dev(g["u2"]),
                                 u3=dev(g["u3"]))
    np.testing.assert_allclose(f_rgb.cpu().numpy(), g["fine_rgb"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(f_w.cpu().numpy(), g["fine_w"], rtol=0, atol=1e-5)
    assert np.array_equal(_bits(w_in.cpu().numpy()), _bits(g["coarse_w_after"]))


def test_fine_bin_flip_rate_with_the_gpus_own_coarse_weights(oracle, ops):
    """End to end the fine pass starts from the GPU's OWN coarse weights, which differ from the reference's by
    fp32 summation-order noise (~1e-7); a cdf ordinate that falls within that distance of a cdf knot then picks
    the neighbouring bin.  Count it at the full batch: bins chosen by the GPU chain vs bins chosen by the
    oracle chain (oracle coarse pass -> oracle sample_pdf), ray_samplers/utils.py:31-56.  Bound: <= 1e-5 of all
    bins; rays without a flipped bin keep the 1e-5 pixel bound (checked on the first 512 rays)."""
    from torch_nerf.amd import shard
    H = W = 800
    n, Sc, Sf = 4096, 64, 128
    focal = float(synth.blender_focal(W))
    pose = synth.pose_spherical(37.0, -30.0, 4.0)
    pix = synth.pixel_batch(11, H, W, n)
    u1c, u1, u2, u3 = (x.numpy() for x in shard.ray_draws(5, 0, n, Sc, Sf, "cpu"))
    t_bins = torch.linspace(2.0, 6.0, Sc + 1)[:-1]
    ps = 4.0 / Sc
    flat_c = synth.nerf_flat_params(seed=3, sigma_bias=1.0, sigma_gain=30.0)
    flat_f = synth.nerf_flat_params(seed=4, sigma_bias=1.0, sigma_gain=30.0)
    k4 = (np.float32(focal), np.float32(focal), W / 2.0, H / 2.0)
    o, d = ops.generate_rays(H, W, k4, torch.from_numpy(pose), False, focal, 2.0, "cuda", pix=dev(pix))
    # GPU chain
    pc, pf = ops.mlp_pack(dev(flat_c)), ops.mlp_pack(dev(flat_f))
    g_rgb, g_w = ops.render_rays(pc, o, d, t_bins.cuda(), ps, dev(u1c))
    g_idx = ops.sample_hierarchical(o, d, t_bins.cuda(), ps, g_w.clone(), dev(u1), dev(u2), dev(u3),
                                    want_idx=True)[3].cpu().numpy()
    # oracle chain
    oo, do = oracle.raygen(oracle.screen_coords(H, W, pix), *k4, pose)
    c = oracle.render_rays(flat_c, oo, do, t_bins.numpy(), ps, u1c)
    o_idx = oracle.hierarchical_sample(oo, do, t_bins.numpy(), ps, c["weights"], u1, u2, u3)[0]
    np.testing.assert_allclose(g_w.cpu().numpy(), c["weights"], rtol=0, atol=2e-6)
    flipped = g_idx != o_idx
    rate = flipped.mean()
    print(f"fine-bin flips: {int(flipped.sum())} of {flipped.size} ({rate:.2e}); "
          f"max |coarse w - oracle| = {np.abs(g_w.cpu().numpy() - c['weights']).max():.2e}")
    assert rate <= 1e-5, f"{int(flipped.sum())} flipped bins"
    assert np.all(np.abs(g_idx[flipped] - o_idx[flipped]) == 1)        # and a flip is always to the neighbour
    # pixels of the fine pass, each chain on its own coarse weights
    m = 512
    f_rgb, _ = ops.render_rays(pf, o[:m], d[:m], t_bins.cuda(), ps, dev(u1[:m]), weights=g_w[:m].clone(),
                               u2=dev(u2[:m]), u3=dev(u3[:m]))
    f = oracle.render_rays(flat_f, oo[:m], do[:m], t_bins.numpy(), ps, u1[:m], weights=c["weights"][:m],
                           u2=u2[:m], u3=u3[:m])
    clean = ~flipped[:m].any(axis=1)
    np.testing.assert_allclose(f_rgb.cpu().numpy()[clean], f["rgb"][clean], rtol=0, atol=1e-5)
