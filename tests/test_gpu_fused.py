"""The single-kernel render pass (csrc/render_fused.hip, nerf_render_pass) against the three-kernel chain it
replaces, the golden vectors and the oracle.  Volume_renderer.py:136-169 / :192-261 is the reference path."""
import numpy as np
import pytest
import torch

from torch_nerf.amd import ops, shard, synth

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _rays(n, seed=3, H=800, W=800):
    focal = float(synth.blender_focal(W))
    pose = torch.from_numpy(synth.pose_spherical(37.0, -30.0, 4.0))
    pix = dev(synth.pixel_batch(seed, H, W, n))
    k4 = (np.float32(focal), np.float32(focal), W / 2.0, H / 2.0)
    return ops.generate_rays(H, W, k4, pose, False, focal, 2.0, "cuda", pix=pix)


def _chain(packed, o, d, t_bins, ps, u1, weights=None, u2=None, u3=None):
    """The pass as separate kernels: sample -> fused encode+MLP -> integrate (what round 1 enqueued)."""
    if weights is None:
        pts, dirs, delta, t = ops.sample_stratified(o, d, t_bins, ps, u1, want_t=True)
        idx = None
    else:
        pts, dirs, delta, idx, t = ops.sample_hierarchical(o, d, t_bins, ps, weights, u1, u2, u3, want_idx=True,
                                                           want_t=True)
    n, S = delta.shape
    sigma, rad = ops.mlp_forward(packed, pts.reshape(-1, 3), dirs.reshape(-1, 3), encoded=False)
    rgb, w = ops.composite_forward(sigma.view(n, S), rad.view(n, S, 3), delta)
    return rgb, w, idx, t


@pytest.mark.parametrize("n", [1, 2, 3, 257, 4096])
def test_fused_pass_is_bit_identical_to_the_kernel_chain(n):
    Sc, Sf = 64, 128
    assert ops.render_is_fused(Sc, Sf, False) and ops.render_is_fused(Sc, Sf, True)
    o, d = _rays(n)
    t_bins = torch.linspace(2.0, 6.0, Sc + 1, device="cuda")[:-1]
    ps = 4.0 / Sc
    u1c, u1, u2, u3 = shard.ray_draws(21, 0, n, Sc, Sf, "cuda")
    pc = ops.mlp_pack(dev(synth.nerf_flat_params(seed=3, sigma_bias=1.0, sigma_gain=30.0)))
    pf = ops.mlp_pack(dev(synth.nerf_flat_params(seed=4, sigma_bias=1.0, sigma_gain=30.0)))
    rgb_c, w_c, t_c = ops.render_rays(pc, o, d, t_bins, ps, u1c, want_t=True)
    ref_rgb, ref_w, _, ref_t = _chain(pc, o, d, t_bins, ps, u1c)
    assert torch.equal(t_c, ref_t) and torch.equal(w_c, ref_w) and torch.equal(rgb_c, ref_rgb)
    w_a, w_b = w_c.clone(), w_c.clone()
    rgb_f, w_f, idx, t_f = ops.render_rays(pf, o, d, t_bins, ps, u1, weights=w_a, u2=u2, u3=u3, want_idx=True,
                                           want_t=True)
    ref_rgb, ref_w, ref_idx, ref_t = _chain(pf, o, d, t_bins, ps, u1, weights=w_b, u2=u2, u3=u3)
    assert torch.equal(idx, ref_idx) and torch.equal(t_f, ref_t)
    assert torch.equal(w_a, w_b) and torch.equal(w_a, w_c + 1e-5)          # the in-place floor (utils.py:31)
    assert torch.equal(w_f, ref_w) and torch.equal(rgb_f, ref_rgb)


def test_fused_pass_against_golden_f7_and_f3_bins(golden):
    g = golden("f7_e2e")
    H, W, focal, near, far = g["meta"]
    H, W = int(H), int(W)
    o, d = ops.generate_rays(H, W, (np.float32(focal), np.float32(focal), W / 2.0, H / 2.0),
                             torch.from_numpy(g["pose"]), False, focal, near, "cuda", pix=dev(g["pix"]))
    t_bins = torch.linspace(float(near), float(far), 65)[:-1].cuda()
    ps = (float(far) - float(near)) / 64
    pc = ops.mlp_pack(dev(synth.nerf_flat_params(seed=3, sigma_bias=1.0, sigma_gain=30.0)))
    pf = ops.mlp_pack(dev(synth.nerf_flat_params(seed=4, sigma_bias=1.0, sigma_gain=30.0)))
    c_rgb, c_w = ops.render_rays(pc, o, d, t_bins, ps, dev(g["u1c"]))
    np.testing.assert_allclose(c_rgb.cpu().numpy(), g["coarse_rgb"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(c_w.cpu().numpy(), g["coarse_w"], rtol=0, atol=1e-5)
    w_in = dev(g["coarse_w"])       # the REFERENCE's coarse weights: bins then match it bit for bit
    f_rgb, f_w = ops.render_rays(pf, o, d, t_bins, ps, dev(g["u1"]), weights=w_in, u2=dev(g["u2"]), u3=dev(g["u3"]))
    np.testing.assert_allclose(f_rgb.cpu().numpy(), g["fine_rgb"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(f_w.cpu().numpy(), g["fine_w"], rtol=0, atol=1e-5)
    assert np.array_equal(w_in.cpu().numpy().view(np.uint32), g["coarse_w_after"].view(np.uint32))
    # bin indices and sorted positions of golden F3 (adversarial weights) out of the fused kernel
    g3 = golden("f3_fine")
    checked = 0
    for case in sorted({k.split("_")[0] for k in g3.files}):
        w = dev(g3[case + "_w_in"])
        n, Sc = w.shape
        Sf = g3[case + "_u2"].shape[1]
        if not ops.render_is_fused(Sc, Sf, True):
            continue
        out = ops.render_rays(pf, dev(g3[case + "_o"]), dev(g3[case + "_d"]), dev(g3[case + "_t_bins"]),
                              float(g3[case + "_ps"][0]), dev(g3[case + "_u1"]), weights=w, u2=dev(g3[case + "_u2"]),
                              u3=dev(g3[case + "_u3"]), want_idx=True, want_t=True)
        assert np.array_equal(out[2].cpu().numpy(), g3[case + "_idx"]), case
        assert np.array_equal(out[3].cpu().numpy().view(np.uint32), g3[case + "_t"].view(np.uint32)), case
        assert np.array_equal(w.cpu().numpy().view(np.uint32), g3[case + "_w_after"].view(np.uint32)), case
        checked += 1
    assert checked >= 2


@pytest.mark.parametrize("Sc,Sf", [(32, 0), (128, 0), (64, 64), (32, 64), (64, 192)])
def test_other_tiling_sample_counts(Sc, Sf):
    """Every (Sc, Sf) whose rays tile the 128-sample pass runs fused and equals the chain; others fall back."""
    n = 37
    o, d = _rays(n, seed=5)
    t_bins = torch.linspace(2.0, 6.0, Sc + 1, device="cuda")[:-1]
    ps = 4.0 / Sc
    u1c, u1, u2, u3 = shard.ray_draws(2, 0, n, Sc, max(Sf, 1), "cuda")
    pf = ops.mlp_pack(dev(synth.nerf_flat_params(seed=4, sigma_bias=1.0, sigma_gain=30.0)))
    assert ops.render_is_fused(Sc, Sf, False)
    rgb, w = ops.render_rays(pf, o, d, t_bins, ps, u1c)
    ref = _chain(pf, o, d, t_bins, ps, u1c)
    assert torch.equal(rgb, ref[0]) and torch.equal(w, ref[1])
    if Sf:
        assert ops.render_is_fused(Sc, Sf, True)
        wa, wb = w.clone(), w.clone()
        rgb2, w2 = ops.render_rays(pf, o, d, t_bins, ps, u1, weights=wa, u2=u2, u3=u3)
        ref2 = _chain(pf, o, d, t_bins, ps, u1, weights=wb, u2=u2, u3=u3)
        assert torch.equal(rgb2, ref2[0]) and torch.equal(w2, ref2[1])


def test_non_tiling_sample_count_falls_back_to_the_chain():
    Sc, n = 40, 19
    assert not ops.render_is_fused(Sc, 0, False)
    o, d = _rays(n, seed=6)
    t_bins = torch.linspace(2.0, 6.0, Sc + 1, device="cuda")[:-1]
    u1 = torch.rand((n, Sc), device="cuda")
    pf = ops.mlp_pack(dev(synth.nerf_flat_params(seed=4, sigma_bias=1.0, sigma_gain=30.0)))
    rgb, w = ops.render_rays(pf, o, d, t_bins, 4.0 / Sc, u1)
    ref = _chain(pf, o, d, t_bins, 4.0 / Sc, u1)
    assert torch.equal(rgb, ref[0]) and torch.equal(w, ref[1])


def test_render_scene_takes_the_fused_pass_for_inference_and_the_chain_for_training():
    """VolumeRenderer.render_scene: same draws (torch.rand stream), same results, with or without grad."""
    import torch_nerf.src.network as network
    import torch_nerf.src.scene as scene
    import torch_nerf.src.renderer.cameras as cameras
    import torch_nerf.src.renderer.integrators.quadrature_integrator as integrators
    import torch_nerf.src.renderer.ray_samplers as ray_samplers
    from torch_nerf.src.renderer.volume_renderer import VolumeRenderer
    from torch_nerf.src.signal_encoder import PositionalEncoder
    H = W = 200
    focal = float(synth.blender_focal(W))
    cam = cameras.PerspectiveCamera({"f_x": focal, "f_y": focal, "img_width": W, "img_height": H},
                                    torch.from_numpy(synth.pose_spherical(10.0, -30.0, 4.0)), 2.0, 6.0)
    enc = {"coord_enc": PositionalEncoder(3, 10, True), "dir_enc": PositionalEncoder(3, 4, True)}
    nets = []
    for seed in (3, 4):
        flat = synth.nerf_flat_params(seed=seed, sigma_bias=1.0, sigma_gain=30.0)
        net = network.NeRF(63, 27)
        net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.split_flat_params(flat).items()})
        nets.append(net.cuda())
    renderer = VolumeRenderer(integrators.QuadratureIntegrator(), ray_samplers.StratifiedSampler(), cam)
    sc, sf = scene.PrimitiveCube(nets[0], enc), scene.PrimitiveCube(nets[1], enc)
    pix = torch.from_numpy(synth.pixel_batch(1, H, W, 300))

    def both_passes():
        torch.manual_seed(7)
        c_rgb, c_idx, c_w = renderer.render_scene(sc, 300, 64, False, 0, pixel_indices=pix)
        w_before = c_w.detach().clone()
        f_rgb, f_idx, f_w = renderer.render_scene(sf, 300, (64, 128), False, 0, pixel_indices=c_idx, weights=c_w)
        assert torch.equal(c_w.detach(), w_before + 1e-5)          # floored in place, visible to the caller
        return c_rgb, f_rgb, f_w, c_idx

    ops.KERNEL_EVENTS = []
    with torch.no_grad():
        a = both_passes()
    tags_nograd = [e[0] for e in ops.KERNEL_EVENTS]
    ops.KERNEL_EVENTS = []
    b = both_passes()                 # parameters require grad: the differentiable three-kernel path
    tags_grad = [e[0] for e in ops.KERNEL_EVENTS]
    ops.KERNEL_EVENTS = None
    assert tags_nograd == ["render_pass", "render_pass"] and tags_grad == ["mlp_forward", "mlp_forward"]
    assert b[1].requires_grad and not a[1].requires_grad
    for x, y in zip(a[:3], b[:3]):
        assert torch.equal(x, y.detach())
    assert a[3].device.type == "cpu" and a[3].dtype == torch.int64 and torch.equal(a[3], pix)
