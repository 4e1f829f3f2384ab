"""Host-side logic of the drop-in classes (no GPU): construction, validation, error
behaviour and checkpoint layout mirror the reference; compute entry points refuse CPU
tensors instead of silently falling back."""
import numpy as np
import pytest
import torch

import torch_nerf.src.network as network
import torch_nerf.src.scene as scene
import torch_nerf.src.renderer.cameras as cameras
import torch_nerf.src.renderer.integrators.quadrature_integrator as integrators
import torch_nerf.src.renderer.ray_samplers as ray_samplers
from torch_nerf.src.renderer.volume_renderer import VolumeRenderer
from torch_nerf.src.signal_encoder import PositionalEncoder, SHEncoder  # noqa: F401  (runner import line)
from torch_nerf.amd import synth


def _camera(h=6, w=5):
    return cameras.PerspectiveCamera({"f_x": 10.0, "f_y": 10.0, "img_width": w, "img_height": h},
                                     torch.eye(4), 2.0, 6.0)


def test_camera_from_dict_and_tensor(golden):
    g = golden("f1_raygen")
    H, W, focal = g["blender_meta"][:3]
    cam = cameras.PerspectiveCamera({"f_x": focal, "f_y": focal, "img_width": W, "img_height": H},
                                    torch.from_numpy(g["blender_pose"]), 2.0, 6.0)
    assert np.array_equal(cam.intrinsic.numpy(), g["blender_intrinsic"])
    assert (cam.img_height, cam.img_width) == (800, 800)
    assert cam.focal_lengths == (float(focal), float(focal))
    cam2 = cameras.PerspectiveCamera(cam.intrinsic, cam.extrinsic, 2.0, 6.0)
    f32 = float(np.float32(focal))  # the tensor path reads the focal length back from fp32
    assert (cam2.img_height, cam2.img_width, cam2.focal_lengths) == (800, 800, (f32, f32))
    with pytest.raises(ValueError):
        cameras.PerspectiveCamera([1, 2, 3], torch.eye(4), 2.0, 6.0)
    with pytest.raises(ValueError):
        cameras.PerspectiveCamera(torch.eye(3), torch.eye(4), 2.0, 6.0)
    with pytest.raises(ValueError):
        cam.extrinsic = torch.eye(3)
    with pytest.raises(TypeError):  # the reference's setter calls isinstance(x, int, float)
        cam.t_near = 1.0


def test_screen_coords_table_matches_reference(golden):
    g = golden("f1_raygen")
    vr = VolumeRenderer(integrators.QuadratureIntegrator(), ray_samplers.StratifiedSampler(), _camera())
    assert np.array_equal(vr.screen_coords.numpy(), g["small_coords_6x5"])
    assert vr.screen_coords.dtype == torch.int64
    vr.camera = _camera(4, 3)  # the setter invalidates the table
    assert vr.screen_coords.shape == (12, 2)
    empty = VolumeRenderer(integrators.QuadratureIntegrator(), ray_samplers.StratifiedSampler())
    with pytest.raises(AssertionError):
        empty.screen_coords


def test_render_scene_argument_validation():
    vr = VolumeRenderer(integrators.QuadratureIntegrator(), ray_samplers.StratifiedSampler(), _camera())
    with pytest.raises(ValueError):
        vr.render_scene(None, 4.0, 8, False, 0)
    with pytest.raises(ValueError):
        vr.render_scene(None, 4, (8, 8, 8), False, 0, pixel_indices=torch.arange(4))
    with pytest.raises(ValueError):
        vr.render_scene(None, 4, (8, 8), False, 0)


def test_sampler_argument_validation():
    s = ray_samplers.StratifiedSampler()
    bundle = ray_samplers.RayBundle(torch.zeros(2, 3), torch.ones(2, 3), 2.0, 6.0, False)
    with pytest.raises(ValueError):
        s.sample_along_rays(bundle, (4, 4), "cpu")               # tuple without weights
    with pytest.raises(ValueError):
        s.sample_along_rays(bundle, 4, "cpu", weights=torch.ones(2, 4))  # int with weights
    with pytest.raises(ValueError):
        s.sample_along_rays(bundle, (4, 4), "cpu", weights=[1.0])
    t_bins, ps = s._create_t_bins(2.0, 6.0, 64, "cpu")
    assert t_bins.shape == (64,) and ps == 4.0 / 64
    assert torch.equal(t_bins, torch.linspace(2.0, 6.0, 65)[:-1])
    cam = cameras.PerspectiveCamera({"f_x": 10.0, "f_y": 11.0, "img_width": 4, "img_height": 4},
                                    torch.eye(4), 0.0, 1.0)
    with pytest.raises(ValueError):  # ambiguous focal length under NDC
        s.generate_rays(torch.zeros(1, 2, dtype=torch.long), cam, True)


def test_no_cpu_fallback():
    s = ray_samplers.StratifiedSampler()
    bundle = ray_samplers.RayBundle(torch.zeros(2, 3), torch.ones(2, 3), 2.0, 6.0, False)
    with pytest.raises(RuntimeError, match="GPU"):
        s.sample_along_rays(bundle, 4, "cpu")
    with pytest.raises(RuntimeError, match="GPU"):
        integrators.QuadratureIntegrator().integrate_along_rays(torch.ones(2, 4), torch.ones(2, 4, 3),
                                                                torch.ones(2, 4))
    with pytest.raises(RuntimeError, match="GPU"):
        PositionalEncoder(3, 10, True).encode(torch.ones(4, 3))
    net = network.NeRF(63, 27)
    with pytest.raises(RuntimeError, match="GPU"):
        net(torch.ones(4, 63), torch.ones(4, 27))


def test_nerf_module_layout_and_errors():
    net = network.NeRF(63, 27)
    sd = net.state_dict()
    expect = synth.split_flat_params(synth.nerf_flat_params(seed=0))
    assert list(sd.keys()) == list(expect.keys())
    assert all(tuple(sd[k].shape) == expect[k].shape for k in sd)
    assert sum(p.numel() for p in net.parameters()) == 595844
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in expect.items()})
    assert (net.pos_dim, net.view_dir_dim, net.feat_dim) == (63, 27, 256)
    with pytest.raises(ValueError):
        net(torch.ones(4, 63, 1), torch.ones(4, 27))
    with pytest.raises(ValueError):
        net(torch.ones(4, 63), torch.ones(5, 27))
    with pytest.raises(ValueError):
        net(torch.ones(4, 60), torch.ones(4, 27))
    with pytest.raises(ValueError):
        net(torch.ones(4, 63), torch.ones(4, 24))
    # every NeRF(pos_dim, view_dir_dim, feat_dim) the reference's constructor accepts (nerf.py:24-63) is built and
    # mapped to a kernel family; on a box without a GPU the call itself refuses (no CPU fallback)
    other = network.NeRF(60, 24, 128)
    assert sum(p.numel() for p in other.parameters()) == synth.param_count(60, 24, 128)
    assert not other._net.fused and network.NeRF(39, 15)._net.fused and not network.NeRF(75, 27)._net.fused
    with pytest.raises(RuntimeError, match="GPU"):
        other(torch.ones(2, 60), torch.ones(2, 24))


def test_instant_ngp_is_a_name_only_because_the_reference_cannot_run_it():
    """Fixture F13 (captured by importing the reference): InstantNeRF constructs but its forward dies in
    spatial_hash_func (int32 overflow of the hash prime) on every device -- there is no behaviour to reproduce, so
    the drop-in keeps the import name and refuses to construct."""
    import json
    import os
    rec = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "f13_instant_ngp.json")))
    assert rec["constructs"] and not rec["forward_runs"] and not rec["spatial_hash_func_runs"]
    assert "int32" in rec["spatial_hash_func_error"] and "overflow" in rec["spatial_hash_func_error"]
    with pytest.raises(NotImplementedError):
        network.InstantNeRF(3, 16, 16, 19, 16, 512)


def test_sh_encoder_contract():
    """SHEncoder(in_dim, degree): spherical_harmonics_encoder.py:21-84; the runners build it for both inputs and size
    the network from its out_dim (runner_utils.py:595-612)."""
    enc = SHEncoder(3, 4)
    assert (enc.in_dim, enc.degree, enc.out_dim) == (3, 4, 16)
    net = network.NeRF(enc.out_dim, enc.out_dim)
    cube = scene.PrimitiveCube(net, {"coord_enc": enc, "dir_enc": enc})
    assert not cube.fused_query and net._net.fused          # encoders run as kernels, the network in the fused family
    with pytest.raises(RuntimeError, match="GPU"):
        enc.encode(torch.ones(4, 3))


def test_positional_encoder_dims():
    assert PositionalEncoder(3, 10, True).out_dim == 63
    assert PositionalEncoder(3, 4, True).out_dim == 27
    assert PositionalEncoder(3, 4, False).out_dim == 24
    assert PositionalEncoder(3, 10, True).in_dim == 3


def test_primitive_cube_contract():
    enc = {"coord_enc": PositionalEncoder(3, 10, True), "dir_enc": PositionalEncoder(3, 4, True)}
    net = network.NeRF(63, 27)
    cube = scene.PrimitiveCube(net, enc)
    assert cube.radiance_field is net and cube.encoders is enc
    with torch.no_grad():
        assert cube.fused_query                      # inference: one kernel, nothing per sample in HBM -> num_ray_batch is moot
    assert not cube.fused_query                      # a recording call keeps activation planes: the caller's batching holds
    assert cube.fused_net().is_shipped
    # encoders whose widths do not match the network: no fused query (the step-by-step path then raises like the reference)
    assert not scene.PrimitiveCube(net, {"coord_enc": PositionalEncoder(3, 8, True),
                                         "dir_enc": PositionalEncoder(3, 4, True)}).fused_query
    # other yaml values (runner_utils.py:584-612): fused as long as the widths fit two / one 32-wide blocks and feat is 256
    e62 = {"coord_enc": PositionalEncoder(3, 6, True), "dir_enc": PositionalEncoder(3, 2, True)}
    spec = scene.PrimitiveCube(network.NeRF(39, 15), e62).fused_net()
    assert spec is not None and spec.key == (39, 15, 256, 6, 1, 2, 1) and not spec.is_shipped
    eno = {"coord_enc": PositionalEncoder(3, 10, False), "dir_enc": PositionalEncoder(3, 4, False)}
    assert scene.PrimitiveCube(network.NeRF(60, 24), eno).fused_net().key == (60, 24, 256, 10, 0, 4, 0)
    assert network.NeRF(60, 24).inferred_net().key == (60, 24, 256, 10, 0, 4, 0)
    # the layered family behind PositionalEncoders: no single-kernel render pass (fused_net), but query_points still hands
    # RAW points to ONE network kernel -- the encodings go straight into its input planes (raw_net / fused_query)
    narrow = scene.PrimitiveCube(network.NeRF(63, 27, 128), enc)
    assert narrow.fused_net() is None and narrow.raw_net().key == (63, 27, 128, 10, 1, 4, 1)
    with torch.no_grad():
        assert narrow.fused_query
    e126 = {"coord_enc": PositionalEncoder(3, 12, True), "dir_enc": PositionalEncoder(3, 6, True)}
    wide = scene.PrimitiveCube(network.NeRF(75, 39), e126)                              # too wide for the fused kernels
    assert wide.fused_net() is None and wide.raw_net().key == (75, 39, 256, 12, 1, 6, 1)
    assert scene.PrimitiveCube(network.NeRF(75, 39), enc).raw_net() is None              # widths do not match the encoders
    with pytest.raises(ValueError):
        scene.PrimitiveCube("not a module", enc)
    with pytest.raises(ValueError):
        scene.PrimitiveCube(net, ["not", "a", "dict"])
    with pytest.warns(UserWarning):
        scene.PrimitiveCube(net, {"coord_enc": enc["coord_enc"]})
    with pytest.raises(ValueError):
        cube.encoders = {"coord_enc": enc["coord_enc"]}
    with pytest.raises(ValueError):
        cube.query_points(torch.ones(2, 4, 3), torch.ones(2, 5, 3))
    assert isinstance(scene.Scene(cube), scene.Scene)


def test_synthetic_generators_are_deterministic():
    a = synth.nerf_flat_params(seed=5)
    b = synth.nerf_flat_params(seed=5)
    assert a.dtype == np.float32 and a.size == 595844 and np.array_equal(a, b)
    assert not np.array_equal(a, synth.nerf_flat_params(seed=6))
    assert abs(synth.blender_focal(800) - 1111.111) < 1e-2
    assert len(synth.blender_orbit_poses()) == 40
    p = synth.pixel_batch(0, 800, 800, 4096)
    assert p.dtype == np.int64 and len(set(p.tolist())) == 4096


def test_sampler_draws_follow_the_reference_order():
    """U1 (N,Sc) -> U2 (N,Sf) -> U3 (N,Sf) with torch.rand, as stratified_sampler.py:77 / utils.py:43,56 draw them:
    a seed reproduces the reference's stream on the same device, for the step-by-step and the fused path alike."""
    s = ray_samplers.StratifiedSampler()
    torch.manual_seed(11)
    u1, u2, u3 = s.draw_uniforms(5, 64, 128, "cpu")
    torch.manual_seed(11)
    want = (torch.rand((5, 64)), torch.rand((5, 128)), torch.rand((5, 128)))
    assert all(torch.equal(a, b) for a, b in zip((u1, u2, u3), want))
    torch.manual_seed(11)
    c1, c2, c3 = s.draw_uniforms(5, 64, 0, "cpu")
    assert torch.equal(c1, want[0]) and c2 is None and c3 is None
    assert s.check_sample_counts(64, None) == (64, 0) and s.check_sample_counts((64, 128), torch.ones(1, 64)) == (64, 128)


def test_nerf_parameters_share_one_blob_without_changing_the_module_surface():
    """NeRF._rehome (round 6): the 22 parameters are views of one flat storage from the constructor on (the kernels read
    them in place, so writes through p.data can never go stale) -- invisible through the nn.Module surface the
    runners use: state_dict keys / shapes / values, load_state_dict, deepcopy, optimizer steps, torch.save round trip."""
    import copy
    import io
    torch.manual_seed(0)
    net = network.NeRF(63, 27)
    params = net._ordered_params()
    store = params[0].untyped_storage().data_ptr()
    off = params[0].data_ptr()
    for p in params:
        assert p.untyped_storage().data_ptr() == store and p.data_ptr() == off and p.is_contiguous() and p.is_leaf
        off += 4 * p.numel()
    # default nn.Linear initialisation survives the move (same generator draws as eleven nn.Linear in a row)
    torch.manual_seed(0)
    ref = [torch.nn.Linear(i, o) for o, i in synth.layer_shapes()]
    for (name, p), q in zip(net.named_parameters(), [t for l in ref for t in (l.weight, l.bias)]):
        assert torch.equal(p, q), name
    # load_state_dict copies INTO the blob; an alias taken before stays an alias
    alias = net.fc_4.weight.data
    expect = synth.split_flat_params(synth.nerf_flat_params(seed=2))
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in expect.items()})
    assert net._ordered_params()[0].untyped_storage().data_ptr() == store
    assert np.array_equal(alias.numpy(), expect["fc_4.weight"])
    # deepcopy: an independent module with its own blob
    twin = copy.deepcopy(net)
    twin.fc_1.bias.data.add_(1.0)
    assert not torch.equal(twin.fc_1.bias, net.fc_1.bias)
    assert twin._ordered_params()[0].untyped_storage().data_ptr() != store
    # a stock optimizer steps the views in place
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    for p in net.parameters():
        p.grad = torch.ones_like(p)
    opt.step()
    assert np.allclose(net.fc_4.weight.detach().numpy(), expect["fc_4.weight"] - 1e-3, atol=1e-6)
    assert np.array_equal(alias.numpy(), net.fc_4.weight.detach().numpy())
    # torch.save / load round trip of the state_dict (runner_utils.py:758-775)
    buf = io.BytesIO()
    torch.save(net.state_dict(), buf)
    buf.seek(0)
    back = torch.load(buf)
    assert list(back) == list(expect) and all(torch.equal(back[k], net.state_dict()[k]) for k in back)
    # .double() / .float(): still one blob afterwards
    net = net.double().float()
    params = net._ordered_params()
    assert all(p.untyped_storage().data_ptr() == params[0].untyped_storage().data_ptr() for p in params)


def test_f16x2_inference_default_comes_from_the_environment(monkeypatch):
    """The runners construct their networks themselves (runner_utils.py:612, :638) and stay unmodified: the split-f16
    inference kernel is switched on for them through the environment; the default is off (BASELINE names fp32)."""
    monkeypatch.delenv("NERF_AMD_F16X2_INFERENCE", raising=False)
    assert network.NeRF(63, 27).f16x2_inference is False
    monkeypatch.setenv("NERF_AMD_F16X2_INFERENCE", "1")
    net = network.NeRF(63, 27)
    assert net.f16x2_inference is True and net.bf16_inference is False
    net.f16x2_inference = False                       # still an ordinary per-instance attribute
    assert network.NeRF(75, 27).f16x2_inference is True
    monkeypatch.setenv("NERF_AMD_F16X2_INFERENCE", "0")
    assert network.NeRF(63, 27).f16x2_inference is False


def test_f16x2_training_default_comes_from_the_environment(monkeypatch):
    monkeypatch.delenv("NERF_AMD_F16X2_TRAINING", raising=False)
    assert network.NeRF(63, 27).f16x2_training is False
    monkeypatch.setenv("NERF_AMD_F16X2_TRAINING", "1")
    assert network.NeRF(63, 27).f16x2_training is True
