"""The drop-in class surface on a real GPU: VolumeRenderer.render_scene driven exactly as
runners/train.py drives it, with the reference's own random draws replayed, compared with the
golden vectors captured from the reference (tests/golden/f7_e2e.npz)."""
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
from torch_nerf.amd import synth

pytestmark = pytest.mark.gpu


def _net(seed):
    net = network.NeRF(63, 27)
    flat = synth.nerf_flat_params(seed=seed, sigma_bias=1.0, sigma_gain=30.0)
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.split_flat_params(flat).items()})
    return net.cuda()


class _Replay:
    """Stands in for torch.rand: hands out the recorded draws in the recorded order."""

    def __init__(self, draws):
        self.draws = [torch.from_numpy(d).cuda() for d in draws]

    def __call__(self, shape, device=None, **kw):
        d = self.draws.pop(0)
        assert tuple(d.shape) == tuple(shape), (d.shape, shape)
        return d


def _setup(g):
    H, W, focal, near, far = g["meta"]
    cam = cameras.PerspectiveCamera({"f_x": focal, "f_y": focal, "img_width": W, "img_height": H},
                                    torch.from_numpy(g["pose"]), float(near), float(far))
    enc = {"coord_enc": PositionalEncoder(3, 10, True), "dir_enc": PositionalEncoder(3, 4, True)}
    net_c, net_f = _net(3), _net(4)
    vr = VolumeRenderer(integrators.QuadratureIntegrator(), ray_samplers.StratifiedSampler())
    vr.camera = cam
    return vr, scene.PrimitiveCube(net_c, enc), scene.PrimitiveCube(net_f, enc), net_c, net_f


def test_render_scene_matches_reference(golden, monkeypatch):
    g = golden("f7_e2e")
    vr, scene_c, scene_f, _, _ = _setup(g)
    pix = torch.from_numpy(g["pix"])
    dev = torch.cuda.current_device()
    monkeypatch.setattr(torch, "rand", _Replay([g["u1c"], g["u1"], g["u2"], g["u3"]]))
    with torch.no_grad():
        c_rgb, c_idx, c_w = vr.render_scene(scene_c, len(pix), 64, False, dev, pixel_indices=pix)
        assert c_idx.device.type == "cpu" and c_idx.dtype == torch.int64 and torch.equal(c_idx, pix)
        assert c_rgb.is_cuda and c_w.shape == (len(pix), 64)
        np.testing.assert_allclose(c_rgb.cpu().numpy(), g["coarse_rgb"], rtol=0, atol=1e-5)
        np.testing.assert_allclose(c_w.cpu().numpy(), g["coarse_w"], rtol=0, atol=1e-5)
        # feed the reference's coarse weights so the fine bins are comparable bit for bit
        w_ref = torch.from_numpy(g["coarse_w"]).cuda()
        f_rgb, f_idx, f_w = vr.render_scene(scene_f, len(pix), (64, 128), False, dev, pixel_indices=c_idx,
                                            weights=w_ref)
    assert torch.equal(f_idx, pix) and f_w.shape == (len(pix), 192)
    np.testing.assert_allclose(f_rgb.cpu().numpy(), g["fine_rgb"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(f_w.cpu().numpy(), g["fine_w"], rtol=0, atol=1e-5)
    # the in-place +1e-5 on the caller's weights tensor (sample_pdf side effect) is preserved
    assert np.array_equal(w_ref.cpu().numpy().view(np.uint32), g["coarse_w_after"].view(np.uint32))


def test_generic_path_equals_fused_path(golden):
    """encode -> NeRF.forward (three kernels) == fused query (one kernel)."""
    g = golden("f5_mlp")
    net = _net(3)
    pe, de = PositionalEncoder(3, 10, True), PositionalEncoder(3, 4, True)
    pts, dirs = torch.from_numpy(g["pts"]).cuda(), torch.from_numpy(g["dirs"]).cuda()
    with torch.no_grad():
        s1, r1 = net(pe.encode(pts), de.encode(dirs))
        s2, r2 = net.forward_fused(pts, dirs)
    np.testing.assert_allclose(s1.cpu().numpy(), s2.cpu().numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(r1.cpu().numpy(), r2.cpu().numpy(), rtol=0, atol=2e-6)


def test_whole_frame_and_random_pixel_modes():
    cam = cameras.PerspectiveCamera({"f_x": 60.0, "f_y": 60.0, "img_width": 40, "img_height": 30},
                                    torch.from_numpy(synth.pose_spherical(20.0, -30.0, 4.0)), 2.0, 6.0)
    enc = {"coord_enc": PositionalEncoder(3, 10, True), "dir_enc": PositionalEncoder(3, 4, True)}
    cube = scene.PrimitiveCube(_net(3), enc)
    vr = VolumeRenderer(integrators.QuadratureIntegrator(), ray_samplers.StratifiedSampler(), cam)
    dev = torch.cuda.current_device()
    with torch.no_grad():
        torch.manual_seed(0)
        rgb, idx, w = vr.render_scene(cube, 1200, 64, False, dev, num_ray_batch=3)
        assert torch.equal(idx, torch.arange(1200)) and rgb.shape == (1200, 3) and w.shape == (1200, 64)
        torch.manual_seed(0)
        rgb1, _, _ = vr.render_scene(cube, 1200, 64, False, dev)  # batching does not change results
        assert torch.equal(rgb, rgb1)
        np.random.seed(3)
        rgb2, idx2, _ = vr.render_scene(cube, 100, 64, False, dev)
        np.random.seed(3)
        assert np.array_equal(idx2.numpy(), np.random.choice(1200, size=[100], replace=False))
        assert rgb2.shape == (100, 3) and torch.isfinite(rgb2).all()
