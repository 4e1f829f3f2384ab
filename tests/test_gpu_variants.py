"""The reference's constructor / yaml space beyond the shipped 63 / 27 / 256: NeRF(pos_dim,
view_dir_dim, feat_dim) for other encode levels, include_input = False and other feat_dim
(R/network/nerf.py:24-63, R/../runners/runner_utils.py:584-612) -- forward AND backward, through every entry that
can serve them, against fixture F11 (captured from the imported reference) and the CPU oracle.

  fused family   (feat_dim 256, widths <= 64 / 32): pre-encoded NeRF.forward, the raw-point fused query
                 (PrimitiveCube.query_points with run-time encode levels), the single-kernel render pass
  layered family (anything else, and any call that wants input gradients): one GEMM launch per layer
Tolerances: sigma / rgb / pixels 1e-5 abs (north star); gradients as for the shipped network (test_gpu_backward.py).
"""
import numpy as np
import pytest
import torch

import torch_nerf.src.network as network
import torch_nerf.src.scene as scene
from torch_nerf.src.signal_encoder import PositionalEncoder
from torch_nerf.amd import ops, synth
from helpers import NET_VARIANTS, check_grad_digest, variant_params

pytestmark = pytest.mark.gpu

FUSED = [t for t, (lp, ld, inc, feat) in NET_VARIANTS.items() if feat == 256 and lp <= 10 and ld <= 4]
LAYERED = [t for t in NET_VARIANTS if t not in FUSED]


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def make_net(flat, dims):
    net = network.NeRF(*dims)
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.split_flat_params(flat, *dims).items()})
    return net.cuda()


def flat_grad(net):
    return torch.cat([p.grad.reshape(-1) for p in net.parameters()]).cpu().numpy()


def upstream(g, sigma, rgb):
    return (sigma * dev(g["g_sigma"])).sum() + (rgb * dev(g["g_rgb"])).sum()


@pytest.mark.parametrize("tag", sorted(NET_VARIANTS))
def test_forward_backward_preencoded(golden, tag):
    """NeRF.forward(pos_enc, dir_enc) as the reference's cube.py:63-72 calls it, parameters' gradients included."""
    g = golden("f11_net_variants")
    flat, dims = variant_params(g, tag)
    net = make_net(flat, dims)
    assert net._net.fused == (tag in FUSED)
    sigma, rgb = net(dev(g[tag + "_pe"]), dev(g[tag + "_de"]))
    np.testing.assert_allclose(sigma.detach().cpu().numpy(), g[tag + "_sigma"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(rgb.detach().cpu().numpy(), g[tag + "_rgb"], rtol=0, atol=1e-5)
    upstream(g, sigma, rgb).backward()
    check_grad_digest(flat_grad(net), g, tag + "_grad_", rtol=2e-4, atol_scale=2e-3, dims=dims)
    with torch.no_grad():                                   # inference entry (no record)
        s2, c2 = net(dev(g[tag + "_pe"]), dev(g[tag + "_de"]))
    assert torch.equal(s2, sigma.detach()) and torch.equal(c2, rgb.detach())


@pytest.mark.parametrize("tag", sorted(NET_VARIANTS))
def test_input_gradients(golden, tag):
    """Gradients w.r.t. the encoded inputs and, through PositionalEncoder.encode, w.r.t. the raw points and
    directions -- what the reference's autograd returns (VERDICT r02 'missing' item 4): out of the fused dX chain for
    the fused family (round 4), out of the layered kernels otherwise."""
    g = golden("f11_net_variants")
    lp, ld, inc, feat = NET_VARIANTS[tag]
    flat, dims = variant_params(g, tag)
    net = make_net(flat, dims)
    pts, dirs = dev(g["pts"]).requires_grad_(True), dev(g["dirs"]).requires_grad_(True)
    pe = PositionalEncoder(3, lp, inc).encode(pts)
    de = PositionalEncoder(3, ld, inc).encode(dirs)
    pe.retain_grad(); de.retain_grad()
    np.testing.assert_allclose(pe.detach().cpu().numpy(), g[tag + "_pe"], rtol=0, atol=5e-7)
    sigma, rgb = net(pe, de)
    np.testing.assert_allclose(sigma.detach().cpu().numpy(), g[tag + "_sigma"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(rgb.detach().cpu().numpy(), g[tag + "_rgb"], rtol=0, atol=1e-5)
    upstream(g, sigma, rgb).backward()
    check_grad_digest(flat_grad(net), g, tag + "_grad_", rtol=2e-4, atol_scale=2e-3, dims=dims)
    for got, want in ((pe.grad, g[tag + "_g_pe"]), (de.grad, g[tag + "_g_de"]), (pts.grad, g[tag + "_g_pts"]),
                      (dirs.grad, g[tag + "_g_dirs"])):
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=2e-4, atol=2e-5 * np.abs(want).max())


@pytest.mark.parametrize("tag", FUSED)
def test_fused_query_with_runtime_levels(golden, tag):
    """PrimitiveCube.query_points on RAW points: the encodings of any (level, include_input) are built in registers
    by the fused kernel (csrc/mlp_tile.h IN_LEVELS); forward, record-mode forward and backward."""
    g = golden("f11_net_variants")
    lp, ld, inc, feat = NET_VARIANTS[tag]
    flat, dims = variant_params(g, tag)
    net = make_net(flat, dims)
    cube = scene.PrimitiveCube(net, {"coord_enc": PositionalEncoder(3, lp, inc), "dir_enc": PositionalEncoder(3, ld, inc)})
    assert cube.raw_net() is not None and cube.fused_net().key == (*dims, lp, int(inc), ld, int(inc))
    M = g["pts"].shape[0]
    pts, dirs = dev(g["pts"]).view(M // 4, 4, 3), dev(g["dirs"]).view(M // 4, 4, 3)
    with torch.no_grad():
        s0, c0 = cube.query_points(pts, dirs)
    sigma, rgb = cube.query_points(pts, dirs)
    np.testing.assert_allclose(sigma.detach().cpu().numpy().reshape(-1), g[tag + "_sigma"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(rgb.detach().cpu().numpy().reshape(-1, 3), g[tag + "_rgb"], rtol=0, atol=1e-5)
    assert torch.equal(s0, sigma.detach()) and torch.equal(c0, rgb.detach())
    upstream(g, sigma.reshape(-1), rgb.reshape(-1, 3)).backward()
    check_grad_digest(flat_grad(net), g, tag + "_grad_", rtol=2e-4, atol_scale=2e-3, dims=dims)


def test_raw_and_preencoded_entries_agree():
    """Raw points through the in-register encodings (compile-time table for the shipped levels, enc_feature for
    run-time levels) against the same network fed with nerf_posenc's rows: the encodings differ by <= 1.5e-7
    (Cody-Waite sincos vs the library), the outputs by far less than the 1e-5 bound."""
    rng = np.random.RandomState(5)
    M = 1000
    pts = dev(rng.uniform(-4, 4, (M, 3)).astype(np.float32))
    dirs = dev(rng.uniform(-1, 1, (M, 3)).astype(np.float32))
    for levels in ((10, 4, True), (6, 2, True), (10, 4, False), (1, 1, True), (3, 5, False)):
        lp, ld, inc = levels
        e_p, e_d = 6 * lp + 3 * inc, 6 * ld + 3 * inc
        spec = ops.Net(e_p, e_d, 256, lp, inc, ld, inc)
        assert spec.fused
        fp = dev(synth.nerf_flat_params(seed=9, pos_dim=e_p, view_dir_dim=e_d, sigma_bias=0.5, sigma_gain=5.0))
        packed = ops.mlp_pack(fp, spec)
        s_raw, c_raw = ops.mlp_forward(packed, pts, dirs, encoded=False, net=spec)
        s_pre, c_pre = ops.mlp_forward(packed, ops.posenc(pts, lp, inc), ops.posenc(dirs, ld, inc), encoded=True, net=spec)
        np.testing.assert_allclose(s_raw.cpu().numpy(), s_pre.cpu().numpy(), rtol=0, atol=1e-5, err_msg=str(levels))
        np.testing.assert_allclose(c_raw.cpu().numpy(), c_pre.cpu().numpy(), rtol=0, atol=1e-5, err_msg=str(levels))


@pytest.mark.parametrize("tag,M", [("l6_l2", 5000), ("l10_l4_noinput", 777), ("l10_l4_f128", 4097), ("l12_l6_f64", 1000),
                                   ("l4_l4", 1)])
def test_full_tensor_vs_oracle(oracle, golden, tag, M):
    """Ragged sizes, every element of every gradient tensor against the CPU oracle (both families).  A pre-activation
    within an ulp of zero may take the other ReLU branch under a different fp32 summation order; the kernel's own
    decisions are read back from its record, must differ from the oracle's in < 1e-5 of all units, and are handed to
    the oracle's backward (force_masks): both then differentiate the same piecewise-linear function, and every
    element must agree to summation-order rounding (2e-5 rel + 2e-5 rms, helpers.assert_grads_match_given_masks)."""
    from helpers import assert_grads_match_given_masks, fused_masks, layered_masks
    g = golden("f11_net_variants")
    lp, ld, inc, feat = NET_VARIANTS[tag]
    flat, dims = variant_params(g, tag)
    rng = np.random.RandomState(M)
    pts = rng.uniform(-3, 3, (M, 3)).astype(np.float32)
    dirs = rng.uniform(-1, 1, (M, 3)).astype(np.float32)
    gs, gc = rng.standard_normal(M).astype(np.float32), rng.standard_normal((M, 3)).astype(np.float32)
    pe, de = oracle.posenc(pts, lp, include_input=inc), oracle.posenc(dirs, ld, include_input=inc)
    want_s, want_c = oracle.mlp_forward(flat, pe, de, F=feat)
    _, _, _, own = oracle.mlp_backward_ex(flat, pe, de, gs, gc, F=feat, want_inputs=False, want_masks=True)
    spec = ops.Net.dims_only(*dims)
    fp = dev(flat)
    split = lambda v: synth.split_flat_params(v, *dims)

    # ---- layered family (serves every network; the only one for feat_dim != 256 / wide inputs)
    sigma, rgb, rec = ops.mlp_layered_forward(fp, dev(pe), dev(de), spec, record=True)
    np.testing.assert_allclose(sigma.cpu().numpy(), want_s, rtol=0, atol=1e-5)
    np.testing.assert_allclose(rgb.cpu().numpy(), want_c, rtol=0, atol=1e-5)
    masks = layered_masks(rec, sigma, M, spec)
    assert (masks != own).mean() < 1e-5, f"{(masks != own).sum()} ReLU decisions differ from the oracle's"
    want_g, want_gp, want_gd, _ = oracle.mlp_backward_ex(flat, pe, de, gs, gc, F=feat, force_masks=masks)
    got, g_pos, g_dir = ops.mlp_layered_backward(fp, dev(pe), dev(de), spec, sigma, rgb, rec, dev(gs), dev(gc),
                                                 want_pos=True, want_dir=True)
    assert_grads_match_given_masks(got.cpu().numpy(), want_g, split, "layered ")
    for a, b in ((g_pos.cpu().numpy(), want_gp), (g_dir.cpu().numpy(), want_gd)):
        np.testing.assert_allclose(a, b, rtol=2e-5, atol=2e-6 * np.abs(b).max())

    # ---- fused family on the same inputs (pre-encoded entry; no input gradients there)
    if spec.fused:
        packed = ops.mlp_pack(fp, spec)
        s2, c2, saved = ops.mlp_forward(packed, dev(pe), dev(de), True, save=True, net=spec)
        np.testing.assert_allclose(s2.cpu().numpy(), want_s, rtol=0, atol=1e-5)
        np.testing.assert_allclose(c2.cpu().numpy(), want_c, rtol=0, atol=1e-5)
        masks = fused_masks(saved, s2, M)
        assert (masks != own).mean() < 1e-5, f"{(masks != own).sum()} ReLU decisions differ from the oracle's"
        want_g = oracle.mlp_backward_ex(flat, pe, de, gs, gc, F=feat, want_inputs=False, force_masks=masks)[0]
        got = ops.mlp_backward(packed, fp, dev(pe), dev(de), True, s2, c2, saved, dev(gs), dev(gc), net=spec)
        assert_grads_match_given_masks(got.cpu().numpy(), want_g, split, "fused ")


@pytest.mark.parametrize("dims,M", [((21, 15, 128), 700), ((63, 27, 128), 2049), ((75, 27, 100), 513), ((96, 32, 128), 256),
                                    ((75, 27, 256), 515), ((93, 27, 256), 130), ((27, 15, 230), 257),
                                    ((63, 27, 512), 300), ((40, 40, 160), 333), ((63, 27, 96), 129),
                                    ((63, 27, 64), 700), ((75, 27, 40), 257), ((21, 15, 50), 130), ((75, 39, 64), 300),
                                    ((63, 27, 64), 1), ((63, 27, 128), 31), ((63, 27, 512), 33), ((75, 27, 256), 2),
                                    ((99, 27, 256), 300), ((99, 27, 64), 515), ((99, 27, 128), 129),
                                    # two direction blocks in the register-resident forward (round 5: dir_encode_level 5..10)
                                    ((63, 33, 256), 1000), ((75, 39, 256), 513), ((99, 63, 256), 300), ((27, 64, 256), 129)])
def test_layered_family_any_widths_vs_oracle(oracle, dims, M):
    """Widths the fixtures do not hold, every element against the CPU oracle under the kernel's own ReLU decisions:
    the register-resident kernels (feat_dim 33..64 and 97..128 with two sample blocks per wavefront, feat_dim 225..256 with
    pos_dim 65..128 = coord_encode_level 12..16 -- one to four position blocks, ragged feat_dim 100 / 230) and
    the general plane-parked kernel (feat_dim 512: two passes per layer; 160: a ragged
    pass; view_dir_dim 40: two direction blocks), forward, parameter gradients and input gradients.  (63, 33, 256) =
    dir_encode_level 5 and the other view_dir_dim 33..64 cases with feat_dim 256 take reg_forward_kernel<1, 8, PB, RECORD, 2>
    (two direction blocks, every PB)."""
    from helpers import assert_grads_match_given_masks, layered_masks
    e_p, e_d, feat = dims
    rng = np.random.RandomState(M + feat)
    pe = rng.uniform(-1, 1, (M, e_p)).astype(np.float32)
    de = rng.uniform(-1, 1, (M, e_d)).astype(np.float32)
    gs, gc = rng.standard_normal(M).astype(np.float32), rng.standard_normal((M, 3)).astype(np.float32)
    flat = synth.nerf_flat_params(seed=21, pos_dim=e_p, view_dir_dim=e_d, feat_dim=feat, sigma_bias=0.3, sigma_gain=6.0)
    spec = ops.Net.dims_only(*dims)
    assert not spec.fused
    fp = dev(flat)
    want_s, want_c = oracle.mlp_forward(flat, pe, de, F=feat)
    s0, c0 = ops.mlp_layered_forward(fp, dev(pe), dev(de), spec)                      # inference entry: nothing recorded
    sigma, rgb, rec = ops.mlp_layered_forward(fp, dev(pe), dev(de), spec, record=True)
    assert torch.equal(s0, sigma) and torch.equal(c0, rgb)
    np.testing.assert_allclose(sigma.cpu().numpy(), want_s, rtol=0, atol=1e-5)
    np.testing.assert_allclose(rgb.cpu().numpy(), want_c, rtol=0, atol=1e-5)
    _, _, _, own = oracle.mlp_backward_ex(flat, pe, de, gs, gc, F=feat, want_inputs=False, want_masks=True)
    masks = layered_masks(rec, sigma, M, spec)
    assert (masks != own).mean() < 1e-5, f"{(masks != own).sum()} ReLU decisions differ from the oracle's"
    want_g, want_gp, want_gd, _ = oracle.mlp_backward_ex(flat, pe, de, gs, gc, F=feat, force_masks=masks)
    got, g_pos, g_dir = ops.mlp_layered_backward(fp, dev(pe), dev(de), spec, sigma, rgb, rec, dev(gs), dev(gc),
                                                 want_pos=True, want_dir=True)
    plain, _, _ = ops.mlp_layered_backward(fp, dev(pe), dev(de), spec, sigma, rgb, rec, dev(gs), dev(gc))
    assert torch.equal(plain, got)
    assert_grads_match_given_masks(got.cpu().numpy(), want_g, lambda v: synth.split_flat_params(v, *dims), f"{dims} ")
    for a, b in ((g_pos.cpu().numpy(), want_gp), (g_dir.cpu().numpy(), want_gd)):
        np.testing.assert_allclose(a, b, rtol=2e-5, atol=2e-6 * np.abs(b).max())


@pytest.mark.parametrize("dims", [(63, 27, 64), (63, 27, 128), (75, 27, 256), (63, 27, 512), (75, 39, 64), (63, 33, 256)])
def test_layered_inference_walks_the_batch_in_chunks(monkeypatch, dims):
    """An inference call (nothing recorded) walks the batch through a scratch of LAYERED_INFERENCE_ROWS rows -- networks
    whose activations stay in registers take longer chunks out of the same bytes (only the two input planes are
    touched).  Whatever the chunking, every sample's outputs are the bits the one-launch recorded forward computes;
    ragged last chunk included."""
    M = 5000
    rng = np.random.RandomState(7)
    pe, de = dev(rng.uniform(-1, 1, (M, dims[0])).astype(np.float32)), dev(rng.uniform(-1, 1, (M, dims[1])).astype(np.float32))
    fp = dev(synth.nerf_flat_params(seed=4, pos_dim=dims[0], view_dir_dim=dims[1], feat_dim=dims[2], sigma_bias=0.3, sigma_gain=6.0))
    spec = ops.Net.dims_only(*dims)
    sigma, rgb, _ = ops.mlp_layered_forward(fp, pe, de, spec, record=True)
    for rows in (256, 300, 1024):
        monkeypatch.setattr(ops, "LAYERED_INFERENCE_ROWS", rows)
        s, c = ops.mlp_layered_forward(fp, pe, de, spec)
        assert torch.equal(s, sigma) and torch.equal(c, rgb), (dims, rows)


def test_layered_gradients_are_deterministic():
    """No atomics: the sliced sample-axis reductions add up in a fixed order."""
    dims = (63, 27, 128)
    flat = synth.nerf_flat_params(seed=2, feat_dim=128)
    rng = np.random.RandomState(0)
    M = 30000
    pe, de = dev(rng.standard_normal((M, 63)).astype(np.float32)), dev(rng.standard_normal((M, 27)).astype(np.float32))
    gs, gc = dev(rng.standard_normal(M).astype(np.float32)), dev(rng.standard_normal((M, 3)).astype(np.float32))
    outs = []
    for _ in range(2):
        net = make_net(flat, dims)
        sigma, rgb = net(pe, de)
        ((sigma * gs).sum() + (rgb * gc).sum()).backward()
        outs.append(flat_grad(net))
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("tag", ["l6_l2", "l10_l4_noinput"])
def test_render_pass_other_levels(oracle, golden, tag):
    """The single-kernel render pass (sampling + in-register encodings of other levels + MLP + integral) and the
    class API around it, coarse and fine, against the oracle chain: bins bit-exact, pixels 1e-5."""
    import torch_nerf.src.renderer.cameras as cameras
    import torch_nerf.src.renderer.integrators.quadrature_integrator as integrators
    import torch_nerf.src.renderer.ray_samplers as ray_samplers
    from torch_nerf.src.renderer.volume_renderer import VolumeRenderer
    g = golden("f11_net_variants")
    lp, ld, inc, feat = NET_VARIANTS[tag]
    flat, dims = variant_params(g, tag)
    n, Sc, Sf = 130, 64, 128
    H = W = 200
    focal = float(synth.blender_focal(W))
    pose = synth.pose_spherical(50.0, -30.0, 4.0)
    pix = synth.pixel_batch(2, H, W, n)
    rng = np.random.RandomState(1)
    u1c, u1 = rng.rand(n, Sc).astype(np.float32), rng.rand(n, Sc).astype(np.float32)
    u2, u3 = rng.rand(n, Sf).astype(np.float32), rng.rand(n, Sf).astype(np.float32)
    t_bins = torch.linspace(2.0, 6.0, Sc + 1)[:-1]
    ps = 4.0 / Sc
    k4 = (np.float32(focal), np.float32(focal), W / 2.0, H / 2.0)
    o, d = ops.generate_rays(H, W, k4, torch.from_numpy(pose), False, focal, 2.0, "cuda:0", pix=dev(pix))
    oo, do = oracle.raygen(oracle.screen_coords(H, W, pix), *k4, pose)
    spec = ops.Net(*dims, lp, inc, ld, inc)
    packed = ops.mlp_pack(dev(flat), spec)

    def ref_pass(weights=None, **kw):
        if weights is None:
            t, pts, dirs, delta = oracle.stratified_sample(oo, do, t_bins.numpy(), ps, kw["u1"])
            idx = None
        else:
            idx, t, pts, dirs, delta, _ = oracle.hierarchical_sample(oo, do, t_bins.numpy(), ps, weights, kw["u1"], kw["u2"], kw["u3"])
        S = delta.shape[1]
        s, c = oracle.mlp_forward(flat, oracle.posenc(pts.reshape(-1, 3), lp, include_input=inc),
                                  oracle.posenc(dirs.reshape(-1, 3), ld, include_input=inc), F=feat)
        rgb, w = oracle.composite_forward(s.reshape(n, S), c.reshape(n, S, 3), delta)
        return rgb, w, idx

    c_rgb, c_w = ops.render_rays(packed, o, d, t_bins.cuda(), ps, dev(u1c), net=spec)
    want_rgb, want_w, _ = ref_pass(u1=u1c)
    np.testing.assert_allclose(c_rgb.cpu().numpy(), want_rgb, rtol=0, atol=1e-5)
    np.testing.assert_allclose(c_w.cpu().numpy(), want_w, rtol=0, atol=1e-5)
    w_in = dev(want_w)
    f_rgb, f_w, idx = ops.render_rays(packed, o, d, t_bins.cuda(), ps, dev(u1), weights=w_in, u2=dev(u2), u3=dev(u3),
                                      want_idx=True, net=spec)
    want_rgb, want_fw, want_idx = ref_pass(weights=want_w, u1=u1, u2=u2, u3=u3)
    assert np.array_equal(idx.cpu().numpy(), want_idx)
    np.testing.assert_allclose(f_rgb.cpu().numpy(), want_rgb, rtol=0, atol=1e-5)
    np.testing.assert_allclose(f_w.cpu().numpy(), want_fw, rtol=0, atol=1e-5)
    # the same through VolumeRenderer.render_scene (its own torch.rand draws: check against the step-by-step path)
    cam = cameras.PerspectiveCamera({"f_x": focal, "f_y": focal, "img_width": W, "img_height": H},
                                    torch.from_numpy(pose), 2.0, 6.0)
    vr = VolumeRenderer(integrators.QuadratureIntegrator(), ray_samplers.StratifiedSampler(), cam)
    cube = scene.PrimitiveCube(make_net(flat, dims), {"coord_enc": PositionalEncoder(3, lp, inc),
                                                     "dir_enc": PositionalEncoder(3, ld, inc)})
    with torch.no_grad():
        torch.manual_seed(3)
        a_rgb, a_idx, a_w = vr.render_scene(cube, n, Sc, False, 0, pixel_indices=torch.from_numpy(pix))
    torch.manual_seed(3)
    b_rgb, b_idx, b_w = vr.render_scene(cube, n, Sc, False, 0, pixel_indices=torch.from_numpy(pix))   # record mode: 3 kernels
    assert b_rgb.requires_grad and not a_rgb.requires_grad
    np.testing.assert_allclose(a_rgb.cpu().numpy(), b_rgb.detach().cpu().numpy(), rtol=0, atol=2e-6)


def test_sh_width_inputs_run_the_fused_family_preencoded(oracle):
    """Widths no PositionalEncoder produces (e.g. 16 / 16, what SHEncoder(3, 4) feeds the network under
    signal_encoder: sh, runner_utils.py:595-604): the fused family through the pre-encoded entry."""
    dims = (16, 16, 256)
    flat = synth.nerf_flat_params(seed=4, pos_dim=16, view_dir_dim=16, sigma_bias=0.5)
    rng = np.random.RandomState(2)
    M = 700
    pe, de = rng.standard_normal((M, 16)).astype(np.float32), rng.standard_normal((M, 16)).astype(np.float32)
    gs, gc = rng.standard_normal(M).astype(np.float32), rng.standard_normal((M, 3)).astype(np.float32)
    net = make_net(flat, dims)
    assert net._net.fused and net.inferred_net() is None
    sigma, rgb = net(dev(pe), dev(de))
    want_s, want_c = oracle.mlp_forward(flat, pe, de, F=256)
    np.testing.assert_allclose(sigma.detach().cpu().numpy(), want_s, rtol=0, atol=1e-5)
    np.testing.assert_allclose(rgb.detach().cpu().numpy(), want_c, rtol=0, atol=1e-5)
    ((sigma * dev(gs)).sum() + (rgb * dev(gc)).sum()).backward()
    want_g = oracle.mlp_backward(flat, pe, de, gs, gc, F=256)
    got = flat_grad(net)
    assert np.linalg.norm(got - want_g) / np.linalg.norm(want_g) < 2e-4
