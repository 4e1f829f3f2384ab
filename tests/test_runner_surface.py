"""The caller contract, checked mechanically (golden F15): everything the reference's runners -- runner_utils.py,
train.py, render.py -- take from the seven hot-path modules exists in the drop-in with a compatible shape.  The fixture
(tests/golden/f15_runner_surface.json) is an `ast` walk over the reference's runner files made in the build container
(tests/golden/make_golden.py:f15_runner_surface): module attributes, constructor call shapes, method call shapes
(positional count + keyword names) and members read / assigned.  Here each record is bound against
`inspect.signature` / `hasattr` of the drop-in's classes, on CPU -- no kernel runs."""
import importlib
import inspect
import json
import os

import pytest
import torch

import torch_nerf.src.network as network
import torch_nerf.src.scene as scene
import torch_nerf.src.renderer.cameras as cameras
import torch_nerf.src.renderer.integrators.quadrature_integrator as integrators
import torch_nerf.src.renderer.ray_samplers as ray_samplers
from torch_nerf.src.renderer.volume_renderer import VolumeRenderer
from torch_nerf.src.signal_encoder import PositionalEncoder

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def surface():
    return json.load(open(os.path.join(HERE, "golden", "f15_runner_surface.json")))


@pytest.fixture(scope="module")
def instances():
    """One CPU instance of every hot-path class, built the way runner_utils.py:526-660 builds them."""
    cam = cameras.PerspectiveCamera({"f_x": 50.0, "f_y": 50.0, "img_width": 8, "img_height": 6}, torch.eye(4), 2.0, 6.0)
    sampler, integ = ray_samplers.StratifiedSampler(), integrators.QuadratureIntegrator()
    enc = {"coord_enc": PositionalEncoder(3, 10, True), "dir_enc": PositionalEncoder(3, 4, True)}
    net = network.NeRF(enc["coord_enc"].out_dim, enc["dir_enc"].out_dim)
    cube = scene.PrimitiveCube(net, enc)
    objs = {"PerspectiveCamera": cam, "StratifiedSampler": sampler, "RaySamplerBase": sampler, "QuadratureIntegrator": integ,
            "NeRF": net, "PrimitiveCube": cube, "PrimitiveBase": cube, "PositionalEncoder": enc["coord_enc"],
            "VolumeRenderer": VolumeRenderer(integ, sampler, cam)}
    return objs


def test_fixture_covers_the_three_runner_files(surface):
    assert surface["runner_files"] == ["torch_nerf/runners/runner_utils.py", "torch_nerf/runners/train.py",
                                       "torch_nerf/runners/render.py"]
    assert len(surface["ctor_calls"]) >= 15 and len(surface["method_calls"]) >= 25 and len(surface["member_uses"]) >= 40
    assert {c["method"] for c in surface["method_calls"]} >= {"render_scene", "parameters", "state_dict", "load_state_dict", "to"}


def test_every_module_attribute_the_runners_name_resolves(surface):
    """`network.NeRF`, `network.InstantNeRF`, `scene.scene` / `scene.Scene` (annotations), `scene.PrimitiveCube`, ... at
    the SAME dotted paths (runner_utils.py:15-21)."""
    for mod, names in surface["module_attrs"].items():
        m = importlib.import_module(mod)
        for name in names:
            assert hasattr(m, name), f"{mod}.{name}"


def test_every_constructor_call_shape_binds(surface):
    for call in surface["ctor_calls"]:
        mod, name = call["callee"].rsplit(".", 1)
        cls = getattr(importlib.import_module(mod), name)
        sig = inspect.signature(cls)
        try:
            sig.bind(*[None] * call["nargs"], **{k: None for k in call["keywords"]})
        except TypeError as exc:
            raise AssertionError(f"{call['callee']} at {call['file']}:{call['line']}: {exc}") from None


def test_every_method_call_shape_binds(surface, instances):
    """render_scene(scene, num_pixels=, num_samples=, project_to_ndc=, device=[, pixel_indices=, weights=,
    num_ray_batch=]) and the nn.Module surface used on scenes' networks (.to / .parameters / .state_dict /
    .load_state_dict)."""
    seen = 0
    for call in surface["method_calls"]:
        owners = call["defined_by"] or ["NeRF"]          # generic nn.Module methods: the network is what the runners call them on
        for cls in owners:
            fn = getattr(instances[cls], call["method"], None)
            assert callable(fn), f"{cls}.{call['method']} ({call['file']}:{call['line']})"
            try:
                inspect.signature(fn).bind(*[None] * call["nargs"], **{k: None for k in call["keywords"]})
            except TypeError as exc:
                raise AssertionError(f"{cls}.{call['method']} at {call['file']}:{call['line']}: {exc}") from None
            seen += 1
    assert seen >= 25


def test_every_member_the_runners_touch_exists(surface, instances):
    """`.camera =` (a property WITH a setter on VolumeRenderer), `.radiance_field`, `.t_near`, `.img_height`, `.out_dim`
    ...: read -> the instance has it; assigned -> assignable without an AttributeError."""
    for use in surface["member_uses"]:
        for cls in use["defined_by"]:
            obj = instances[cls]
            assert hasattr(obj, use["member"]), f"{cls}.{use['member']} ({use['file']}:{use['line']})"
            if use["store"]:
                static = inspect.getattr_static(type(obj), use["member"], None)
                if isinstance(static, property):
                    assert static.fset is not None, f"{cls}.{use['member']} is read-only; assigned at {use['file']}:{use['line']}"
                else:
                    setattr(obj, use["member"], getattr(obj, use["member"]))     # plain attribute: assignable
