"""Row f3: checkpoints in the reference's layout (runners/runner_utils.py:737-830), against the structure captured
from the reference's own modules (tests/golden/f9_checkpoint.npz), and a save -> load round trip on CPU."""
import numpy as np
import torch

from torch_nerf.amd import checkpoint
from torch_nerf.amd.optim import FusedAdam
from torch_nerf.src.network import NeRF
from torch_nerf.src.scene import PrimitiveCube
from torch_nerf.src.signal_encoder import PositionalEncoder


def _scenes(seed):
    torch.manual_seed(seed)
    enc = {"coord_enc": PositionalEncoder(3, 10, True), "dir_enc": PositionalEncoder(3, 4, True)}
    return PrimitiveCube(NeRF(63, 27), enc), PrimitiveCube(NeRF(63, 27), enc)


def _trained(seed):
    coarse, fine = _scenes(seed)
    params = list(coarse.radiance_field.parameters()) + list(fine.radiance_field.parameters())
    opt = torch.optim.Adam(params, lr=0.0005, eps=1e-8)
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, pow(0.00005 / 0.0005, 1 / 300000))
    for p in params:
        p.grad = torch.full_like(p, 0.01)
    opt.step(); sched.step()
    return coarse, fine, opt, sched


def test_checkpoint_file_has_the_reference_structure(golden, tmp_path):
    g = golden("f9_checkpoint")
    coarse, fine, opt, sched = _trained(0)
    path = checkpoint.save_checkpoint(tmp_path, 7, coarse, fine, opt, sched)
    assert path.name == str(g["file_name"][0])
    ckpt = torch.load(path, map_location="cpu")
    assert sorted(ckpt.keys()) == list(g["top_keys"])
    for name in ("scene_default", "scene_fine"):
        sd = ckpt[name]
        assert list(sd.keys()) == list(g["scene_keys"])
        for v, shape, dtype in zip(sd.values(), g["scene_shapes"], g["scene_dtypes"]):
            assert list(v.shape) == [int(d) for d in shape[: v.ndim]] and str(v.dtype) == str(dtype)
            assert v.device.type == "cpu"
    osd = ckpt["optimizer_state_dict"]
    assert sorted(osd.keys()) == list(g["optimizer_keys"])
    assert len(osd["param_groups"][0]["params"]) == int(g["param_group_size"][0]) == 44
    assert sorted(osd["state"][0].keys()) == list(g["state_keys"])
    assert float(osd["state"][0]["step"]) == float(g["state_step"][0])
    assert sorted(ckpt["scheduler_state_dict"].keys()) == list(g["scheduler_keys"])


def test_fused_adam_groups_carry_every_key_of_torch_adam(golden):
    g = golden("f9_checkpoint")
    p = torch.nn.Parameter(torch.zeros(4))
    keys = set(FusedAdam([p], lr=1e-3).state_dict()["param_groups"][0].keys())
    assert set(g["param_group_keys"]) - {"initial_lr"} <= keys          # initial_lr is added by the scheduler


def test_round_trip_restores_networks_optimizer_and_schedule(tmp_path):
    coarse, fine, opt, sched = _trained(1)
    checkpoint.save_checkpoint(tmp_path, 3, coarse, fine, opt, sched)
    checkpoint.save_checkpoint(tmp_path, 12, coarse, fine, opt, sched)      # the latest file wins
    c2, f2 = _scenes(99)
    params2 = list(c2.radiance_field.parameters()) + list(f2.radiance_field.parameters())
    opt2 = torch.optim.Adam(params2, lr=1.0)
    sched2 = torch.optim.lr_scheduler.ExponentialLR(opt2, 0.5)
    assert checkpoint.load_checkpoint(tmp_path, c2, f2, opt2, sched2, device="cpu") == 12
    for a, b in zip(coarse.radiance_field.state_dict().values(), c2.radiance_field.state_dict().values()):
        assert torch.equal(a, b)
    for a, b in zip(fine.radiance_field.state_dict().values(), f2.radiance_field.state_dict().values()):
        assert torch.equal(a, b)
    assert opt2.param_groups[0]["lr"] == opt.param_groups[0]["lr"] and sched2.gamma == sched.gamma
    p0, q0 = next(iter(opt.state)), next(iter(opt2.state))
    assert torch.equal(opt.state[p0]["exp_avg"], opt2.state[q0]["exp_avg"])
    assert checkpoint.load_checkpoint(tmp_path / "missing", c2, f2) == 0
    assert checkpoint.load_checkpoint(None, c2, f2) == 0
