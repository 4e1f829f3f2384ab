"""`signal_encoder: sh` (R/../configs/signal_encoder/sh.yaml, runner_utils.py:595-604) on the GPU: SHEncoder.encode and
its reverse against fixture F12 (captured from the imported reference), the 16 / 16-wide network behind it, and a
VolumeRenderer pass over such a scene against the oracle chain."""
import numpy as np
import pytest
import torch

import torch_nerf.src.network as network
import torch_nerf.src.scene as scene
import torch_nerf.src.renderer.cameras as cameras
import torch_nerf.src.renderer.integrators.quadrature_integrator as integrators
import torch_nerf.src.renderer.ray_samplers as ray_samplers
from torch_nerf.src.renderer.volume_renderer import VolumeRenderer
from torch_nerf.src.signal_encoder import SHEncoder
from torch_nerf.amd import ops, synth
from helpers import check_grad_digest

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def sh_net():
    flat = synth.nerf_flat_params(seed=6, pos_dim=16, view_dir_dim=16, sigma_bias=0.5, sigma_gain=4.0)
    net = network.NeRF(16, 16)
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.split_flat_params(flat, 16, 16).items()})
    return net.cuda(), flat


@pytest.mark.parametrize("degree", [1, 2, 3, 4, 5])
def test_sh_encode_bit_exact_and_backward(golden, degree):
    g = golden("f12_sh_encoder")
    enc = SHEncoder(3, degree)
    x = dev(g["pts"]).requires_grad_(True)
    e = enc.encode(x)
    assert np.array_equal(e.detach().cpu().numpy().view(np.uint32), g[f"d{degree}_enc"].view(np.uint32))
    with torch.no_grad():
        assert torch.equal(enc.encode(dev(g["pts"])), e.detach())
    (e * dev(g[f"d{degree}_g_enc"])).sum().backward()
    want = g[f"d{degree}_g_pts"]
    np.testing.assert_allclose(x.grad.cpu().numpy(), want, rtol=1e-5, atol=2e-6 * max(1.0, np.abs(want).max()))


def test_sh_scene_forward_backward(golden):
    """PrimitiveCube(NeRF(16, 16), two SHEncoder(3, 4)).query_points as the runners call it: outputs, parameter
    gradients, and (inputs requiring grad) the gradients w.r.t. points and directions."""
    g = golden("f12_sh_encoder")
    net, _ = sh_net()
    enc = SHEncoder(3, 4)
    cube = scene.PrimitiveCube(net, {"coord_enc": enc, "dir_enc": enc})
    M = g["pts"].shape[0]
    pts, dirs = dev(g["pts"]).view(M // 8, 8, 3), dev(g["dirs"]).view(M // 8, 8, 3)
    sigma, rgb = cube.query_points(pts, dirs)                      # fused family, pre-encoded entry
    np.testing.assert_allclose(sigma.detach().cpu().numpy().reshape(-1), g["net_sigma"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(rgb.detach().cpu().numpy().reshape(-1, 3), g["net_rgb"], rtol=0, atol=1e-5)
    ((sigma.reshape(-1) * dev(g["net_g_sigma"])).sum() + (rgb.reshape(-1, 3) * dev(g["net_g_rgb"])).sum()).backward()
    grad = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).cpu().numpy()
    check_grad_digest(grad, g, "net_grad_", rtol=2e-4, atol_scale=2e-3, dims=(16, 16, 256))
    net.zero_grad()
    pts.requires_grad_(True); dirs.requires_grad_(True)             # layered family + encoder reverse passes
    sigma, rgb = cube.query_points(pts, dirs)
    ((sigma.reshape(-1) * dev(g["net_g_sigma"])).sum() + (rgb.reshape(-1, 3) * dev(g["net_g_rgb"])).sum()).backward()
    for got, want in ((pts.grad.reshape(-1, 3), g["net_g_pts"]), (dirs.grad.reshape(-1, 3), g["net_g_dirs"])):
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=2e-4, atol=2e-5 * np.abs(want).max())


def test_render_scene_with_sh_encoders(oracle):
    """One coarse + one fine render_scene pass over an SH scene through the class API (sampling kernel -> SH kernels ->
    pre-encoded fused MLP -> integral), replaying the draws, against the oracle chain: bins bit-exact, pixels 1e-5."""
    from torch_nerf.amd import ops
    net_c, flat_c = sh_net()
    enc = SHEncoder(3, 4)
    cube = scene.PrimitiveCube(net_c, {"coord_enc": enc, "dir_enc": enc})
    H = W = 100
    focal = float(synth.blender_focal(W))
    pose = synth.pose_spherical(20.0, -30.0, 4.0)
    cam = cameras.PerspectiveCamera({"f_x": focal, "f_y": focal, "img_width": W, "img_height": H}, torch.from_numpy(pose), 2.0, 6.0)
    vr = VolumeRenderer(integrators.QuadratureIntegrator(), ray_samplers.StratifiedSampler(), cam)
    n, Sc, Sf = 70, 64, 128
    pix = synth.pixel_batch(3, H, W, n)
    with torch.no_grad():
        torch.manual_seed(11)
        c_rgb, c_idx, c_w = vr.render_scene(cube, n, Sc, False, 0, pixel_indices=torch.from_numpy(pix))
        w_before = c_w.clone()
        f_rgb, _, f_w = vr.render_scene(cube, n, (Sc, Sf), False, 0, pixel_indices=c_idx, weights=c_w)
    # replay: the sampler draws u1 (coarse), then u1, u2, u3 (fine) from torch's CUDA generator in that order
    torch.manual_seed(11)
    u1c = torch.rand((n, Sc), device="cuda")
    u1, u2, u3 = torch.rand((n, Sc), device="cuda"), torch.rand((n, Sf), device="cuda"), torch.rand((n, Sf), device="cuda")
    k4 = (np.float32(focal), np.float32(focal), W / 2.0, H / 2.0)
    oo, do = oracle.raygen(oracle.screen_coords(H, W, pix), *k4, pose)
    t_bins = torch.linspace(2.0, 6.0, Sc + 1)[:-1].numpy()
    ps = 4.0 / Sc

    def ref(weights, a, b=None, c=None):
        if weights is None:
            t, pts, dirs, delta = oracle.stratified_sample(oo, do, t_bins, ps, a)
        else:
            _, t, pts, dirs, delta, _ = oracle.hierarchical_sample(oo, do, t_bins, ps, weights, a, b, c)
        S = delta.shape[1]
        s, col = oracle.mlp_forward(flat_c, oracle.shenc(pts.reshape(-1, 3), 4), oracle.shenc(dirs.reshape(-1, 3), 4))
        return oracle.composite_forward(s.reshape(n, S), col.reshape(n, S, 3), delta)

    want_rgb, want_w = ref(None, u1c.cpu().numpy())
    np.testing.assert_allclose(c_rgb.cpu().numpy(), want_rgb, rtol=0, atol=1e-5)
    np.testing.assert_allclose(w_before.cpu().numpy(), want_w, rtol=0, atol=1e-5)
    want_rgb, want_fw = ref(w_before.cpu().numpy(), u1.cpu().numpy(), u2.cpu().numpy(), u3.cpu().numpy())
    np.testing.assert_allclose(f_rgb.cpu().numpy(), want_rgb, rtol=0, atol=1e-5)
    np.testing.assert_allclose(f_w.cpu().numpy(), want_fw, rtol=0, atol=1e-5)


def test_sh_encoder_backward_with_strided_input(oracle):
    """a column slice / transposed view that requires grad -- the reverse kernel must see the same rows the
    forward encoded (its contiguous fp32 copy), and the gradient must come back in the input's own layout."""
    rng = np.random.RandomState(9)
    wide = torch.from_numpy(rng.uniform(-1, 1, (300, 6)).astype(np.float32)).cuda().requires_grad_(True)
    x = wide[:, 3:6]                                   # strides (6, 1): not contiguous
    assert not x.is_contiguous()
    g_out = torch.from_numpy(rng.standard_normal((300, 16)).astype(np.float32)).cuda()
    enc = ops.ShencFunction.apply(x, 4)
    (enc * g_out).sum().backward()
    want = oracle.shenc_backward(wide.detach().cpu().numpy()[:, 3:6].copy(), g_out.cpu().numpy(), 4)
    got = wide.grad.cpu().numpy()
    np.testing.assert_allclose(got[:, 3:6], want, rtol=1e-5, atol=1e-6)
    assert np.all(got[:, :3] == 0)
    xt = torch.from_numpy(rng.uniform(-1, 1, (3, 200)).astype(np.float32)).cuda().requires_grad_(True)
    enc = ops.ShencFunction.apply(xt.t(), 4)           # transposed view
    (enc * g_out[:200]).sum().backward()
    want = oracle.shenc_backward(xt.detach().cpu().numpy().T.copy(), g_out[:200].cpu().numpy(), 4)
    np.testing.assert_allclose(xt.grad.cpu().numpy().T, want, rtol=1e-5, atol=1e-6)
