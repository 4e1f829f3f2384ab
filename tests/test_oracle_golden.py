"""The CPU oracle against the golden vectors captured from the imported reference.

This is what pins the oracle: integer quantities (screen
coordinates, fine-sample bin indices) and everything built only from IEEE +,-,*,/
must match the reference bit for bit; transcendental / GEMM-order quantities match
to the tolerances written next to each assert.
"""
import numpy as np
import pytest

from torch_nerf.amd import synth
from helpers import NET_VARIANTS, check_grad_digest, variant_params


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_screen_coords_and_rays(golden, oracle):
    g = golden("f1_raygen")
    assert np.array_equal(oracle.screen_coords(6, 5), g["small_coords_6x5"])
    for name in ("blender", "blender400", "llff_ndc0", "llff_ndc1"):
        H, W, focal, near, far, ndc = g[name + "_meta"]
        H, W = int(H), int(W)
        pix = g[name + "_pix"]
        coords = oracle.screen_coords(H, W, pix)
        assert np.array_equal(coords, g[name + "_coords"]), name
        K = g[name + "_intrinsic"]
        o, d = oracle.raygen(coords, K[0, 0], K[1, 1], K[0, 2], K[1, 2], g[name + "_pose"])
        if ndc:
            o, d = oracle.map_rays_to_ndc(float(focal), float(near), H, W, o, d)
        # sgemm order / FMA use inside ATen's (N,3)@(3,3) is not restated: 2 ulp-ish
        np.testing.assert_allclose(o, g[name + "_o"], rtol=3e-6, atol=1e-6, err_msg=name)
        np.testing.assert_allclose(d, g[name + "_d"], rtol=3e-6, atol=1e-6, err_msg=name)


def test_pose_generator_matches_fixture_pose(golden):
    g = golden("f1_raygen")
    assert np.array_equal(synth.pose_spherical(37.0, -30.0, 4.0), g["blender_pose"])


@pytest.mark.parametrize("name", ["b", "ndc", "odd"])
def test_coarse_sampling_bit_exact(golden, oracle, name):
    g = golden("f2_coarse")
    t, pts, dirs, delta = oracle.stratified_sample(g[name + "_o"], g[name + "_d"], g[name + "_t_bins"],
                                                   float(g[name + "_ps"][0]), g[name + "_u1"])
    assert np.array_equal(_bits(delta), _bits(g[name + "_delta"]))
    assert np.array_equal(_bits(pts), _bits(g[name + "_pts"]))
    assert np.array_equal(_bits(dirs), _bits(g[name + "_dirs"]))


@pytest.mark.parametrize("name", ["b", "ndc", "s128", "s40", "s1000"])
def test_fine_sampling_bit_exact(golden, oracle, name):
    g = golden("f3_fine")
    # ATen's sum order, restated
    w_after = g[name + "_w_after"]
    norm = np.array([oracle.aten_sum_lastdim(r) for r in w_after], np.float32)
    assert np.array_equal(_bits(norm), _bits(g[name + "_norm"])), "sum order"
    idx, t, pts, dirs, delta, w = oracle.hierarchical_sample(
        g[name + "_o"], g[name + "_d"], g[name + "_t_bins"], float(g[name + "_ps"][0]),
        g[name + "_w_in"], g[name + "_u1"], g[name + "_u2"], g[name + "_u3"])
    assert np.array_equal(_bits(w), _bits(w_after)), "in-place floor"
    assert np.array_equal(idx, g[name + "_idx"].astype(np.int64)), "bin indices"
    assert np.array_equal(_bits(t), _bits(g[name + "_t"])), "sorted t"
    assert np.array_equal(_bits(delta), _bits(g[name + "_delta"]))
    assert np.array_equal(_bits(pts), _bits(g[name + "_pts"]))


def test_posenc(golden, oracle):
    g = golden("f4_posenc")
    x = g["x"]
    # glibc sinf/cosf vs ATen's SLEEF: <= 2 ulp of values in [-1,1]
    np.testing.assert_allclose(oracle.posenc(x, 10), g["pe10"], rtol=0, atol=3e-7)
    np.testing.assert_allclose(oracle.posenc(x, 4), g["pe4"], rtol=0, atol=3e-7)
    np.testing.assert_allclose(oracle.posenc(x, 4, include_input=False), g["pe4_noinput"], rtol=0, atol=3e-7)
    # layout: raw input first, then per-frequency [sin(xyz), cos(xyz)]
    assert np.array_equal(oracle.posenc(x, 10)[:, :3], x)


GRAD_KW = {"default": dict(seed=1), "dense": dict(seed=2, sigma_bias=1.0, sigma_gain=30.0)}


@pytest.mark.parametrize("tag", ["default", "dense"])
def test_mlp_forward_backward(golden, oracle, tag):
    g = golden("f5_mlp")
    flat = synth.nerf_flat_params(**GRAD_KW[tag])
    pe = oracle.posenc(g["pts"], 10)
    de = oracle.posenc(g["dirs"], 4)
    sigma, rgb = oracle.mlp_forward(flat, pe, de)
    # north-star tolerance: 1e-5 abs on sigma / rgb
    np.testing.assert_allclose(sigma, g[tag + "_sigma"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(rgb, g[tag + "_rgb"], rtol=0, atol=1e-5)
    grad = oracle.mlp_backward(flat, pe, de, g[tag + "_g_sigma"], g[tag + "_g_rgb"])
    check_grad_digest(grad, g, tag + "_grad_", rtol=2e-4, atol_scale=2e-3)


@pytest.mark.parametrize("tag", sorted(NET_VARIANTS))
def test_mlp_variants(golden, oracle, tag):
    """F11: the network built from other yaml values (encode levels, include_input) and other feat_dim
    (runner_utils.py:584-612, nerf.py:24-63): outputs, parameter gradients, and the gradients w.r.t. the encoded inputs."""
    g = golden("f11_net_variants")
    lp, ld, inc, feat = NET_VARIANTS[tag]
    flat, dims = variant_params(g, tag)
    pe = oracle.posenc(g["pts"], lp, include_input=inc)
    de = oracle.posenc(g["dirs"], ld, include_input=inc)
    assert pe.shape[1] == dims[0] and de.shape[1] == dims[1]
    np.testing.assert_allclose(pe, g[tag + "_pe"], rtol=0, atol=3e-7)
    sigma, rgb = oracle.mlp_forward(flat, g[tag + "_pe"], g[tag + "_de"], F=feat)
    np.testing.assert_allclose(sigma, g[tag + "_sigma"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(rgb, g[tag + "_rgb"], rtol=0, atol=1e-5)
    grad, g_pe, g_de, _ = oracle.mlp_backward_ex(flat, g[tag + "_pe"], g[tag + "_de"], g["g_sigma"], g["g_rgb"], F=feat)
    check_grad_digest(grad, g, tag + "_grad_", rtol=2e-4, atol_scale=2e-3, dims=dims)
    for got, want in ((g_pe, g[tag + "_g_pe"]), (g_de, g[tag + "_g_de"])):
        np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-5 * np.abs(want).max())


@pytest.mark.parametrize("degree", [1, 2, 3, 4, 5])
def test_sh_encoder(golden, oracle, degree):
    """F12: SHEncoder.encode (spherical_harmonics_encoder.py:86-139) -- products in the reference's order, so the
    fp32 values are BIT-identical; the reverse pass against the reference's autograd."""
    g = golden("f12_sh_encoder")
    e = oracle.shenc(g["pts"], degree)
    assert np.array_equal(e.view(np.uint32), g[f"d{degree}_enc"].view(np.uint32))
    want = g[f"d{degree}_g_pts"]
    np.testing.assert_allclose(oracle.shenc_backward(g["pts"], g[f"d{degree}_g_enc"], degree), want, rtol=1e-5,
                               atol=2e-6 * max(1.0, np.abs(want).max()))


def test_sh_network(golden, oracle):
    """F12: NeRF(16, 16) behind two SHEncoder(3, 4) (signal_encoder: sh, runner_utils.py:595-604)."""
    g = golden("f12_sh_encoder")
    flat = synth.nerf_flat_params(seed=6, pos_dim=16, view_dir_dim=16, sigma_bias=0.5, sigma_gain=4.0)
    pe, de = oracle.shenc(g["pts"], 4), oracle.shenc(g["dirs"], 4)
    sigma, rgb = oracle.mlp_forward(flat, pe, de)
    np.testing.assert_allclose(sigma, g["net_sigma"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(rgb, g["net_rgb"], rtol=0, atol=1e-5)
    grad, g_pe, g_de, _ = oracle.mlp_backward_ex(flat, pe, de, g["net_g_sigma"], g["net_g_rgb"])
    check_grad_digest(grad, g, "net_grad_", rtol=2e-4, atol_scale=2e-3, dims=(16, 16, 256))
    for got, want in ((oracle.shenc_backward(g["pts"], g_pe, 4), g["net_g_pts"]),
                      (oracle.shenc_backward(g["dirs"], g_de, 4), g["net_g_dirs"])):
        np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-5 * np.abs(want).max())


@pytest.mark.parametrize("S", [64, 192, 7])
def test_composite(golden, oracle, S):
    g = golden("f6_composite")
    p = f"S{S}_"
    rgb, w = oracle.composite_forward(g[p + "sigma"], g[p + "c"], g[p + "delta"])
    np.testing.assert_allclose(w, g[p + "w"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(rgb, g[p + "rgb"], rtol=0, atol=1e-5)
    gs, gc = oracle.composite_backward(g[p + "sigma"], g[p + "c"], g[p + "delta"], g[p + "g_rgb"])
    np.testing.assert_allclose(gc, g[p + "gc"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(gs, g[p + "gs"], rtol=1e-4, atol=1e-4)
    gs, gc = oracle.composite_backward(g[p + "sigma"], g[p + "c"], g[p + "delta"], g[p + "g_rgb"],
                                       g[p + "g_w"])
    np.testing.assert_allclose(gc, g[p + "gc_w"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(gs, g[p + "gs_w"], rtol=1e-4, atol=1e-4)


def test_end_to_end(golden, oracle):
    g = golden("f7_e2e")
    H, W, focal, near, far = g["meta"]
    H, W = int(H), int(W)
    assert int(g["idx_match"][0]) == 1
    coords = oracle.screen_coords(H, W, g["pix"])
    o, d = oracle.raygen(coords, np.float32(focal), np.float32(focal), W / 2.0, H / 2.0, g["pose"])
    import torch
    t_bins = torch.linspace(float(near), float(far), 65)[:-1].numpy()
    ps = (float(far) - float(near)) / 64
    pc = synth.nerf_flat_params(seed=3, sigma_bias=1.0, sigma_gain=30.0)
    pf = synth.nerf_flat_params(seed=4, sigma_bias=1.0, sigma_gain=30.0)
    c = oracle.render_rays(pc, o, d, t_bins, ps, g["u1c"])
    np.testing.assert_allclose(c["rgb"], g["coarse_rgb"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(c["weights"], g["coarse_w"], rtol=0, atol=1e-5)
    # fine pass driven by the REFERENCE's coarse weights so that bins are comparable bit for bit
    f = oracle.render_rays(pf, o, d, t_bins, ps, g["u1"], weights=g["coarse_w"], u2=g["u2"], u3=g["u3"])
    np.testing.assert_allclose(f["rgb"], g["fine_rgb"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(f["weights"], g["fine_w"], rtol=0, atol=1e-5)


def test_end_to_end_gradients_with_the_references_relu_decisions(golden, oracle):
    """loss = MSE(coarse) + MSE(fine) differentiated through integrator + MLP of both networks (train.py:172-215)
    against the gradients the reference's autograd returned (F7 digests).  F7 keeps a 64-bit digest per sample of the
    ReLU decisions autograd differentiated through; the oracle's own differ from them in ~20 of 24 576 samples
    (one unit each, pre-activation within rounding of zero).  They are rebuilt exactly (helpers.
    reference_relu_decisions: toggle marginal units until the digest matches), forced into the oracle's backward, and
    the result is compared at summation-order tolerance -- 30x tighter than a flip-blind comparison allows."""
    from helpers import F7_TIGHT, check_grad_digest, f7_oracle_chain, f7_oracle_grad, reference_relu_decisions
    g = golden("f7_e2e")
    chain = f7_oracle_chain(oracle, g)
    for tag, c in chain.items():
        ref_masks, differing, unresolved = reference_relu_decisions(oracle, c["params"], c["pe"], c["de"], c["masks"],
                                                                    g[tag + "_relu_hash"])
        assert unresolved.size == 0, f"{tag}: decisions of samples {unresolved} could not be rebuilt"
        toggled = int((ref_masks != c["masks"]).sum())
        assert toggled < 1e-5 * ref_masks.size, (tag, toggled)          # measured: 5 / 15 units of 13.4 M / 40.1 M
        check_grad_digest(f7_oracle_grad(oracle, c, ref_masks), g, tag + "_grad_", **F7_TIGHT)


def test_adam_and_exponential_lr(golden, oracle):
    """Row f1: the optimizer step (runner_utils.py:691-711, train.py:215-218) against torch.optim.Adam +
    ExponentialLR driven the reference's way.  The learning-rate sequence is exact (double arithmetic);
    parameters match to one ulp of the parameter plus 2e-6 of the step taken (ATen's vectorised
    lerp / sqrt / div roundings are not restated bit for bit)."""
    g = golden("f8_adam")
    init_lr, end_lr, num_iter, eps = g["config"]
    p = g["p0"].copy()
    m, v = np.zeros_like(p), np.zeros_like(p)
    for s in range(g["grads"].shape[0]):
        lr = oracle.exponential_lr(init_lr, end_lr, int(num_iter), s)
        assert lr == g["lrs"][s]
        before = p.copy()
        oracle.adam_step(p, np.ascontiguousarray(g["grads"][s]), m, v, s + 1, lr, eps=eps)
        want = g["params"][s]
        step_size = np.abs(want - before).max()
        tol = np.spacing(np.abs(want)) + 2e-6 * step_size
        assert np.all(np.abs(p - want) <= tol), f"step {s}: {np.abs(p - want).max()}"
    np.testing.assert_allclose(m, g["exp_avg"], rtol=2e-6, atol=1e-12)
    np.testing.assert_allclose(v, g["exp_avg_sq"], rtol=2e-6, atol=1e-30)
