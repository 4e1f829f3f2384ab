"""Host-side bookkeeping of the layered family's backward, swept on the CPU.

nerf_mlp_layered_plan_check runs everything nerf_mlp_layered_backward does on the host before it launches -- workspace
layout, the dW window list (mlp_layered.hip:enumerate_dw_items) against dw_item_budget, the list's descriptors and
partial-tile buffer (mlp_backward.hip:plan_dw_items) -- with stand-in base pointers, and checks every destination
rectangle, every read extent and every budget.  The advisor found a latent overflow exactly here in round 5; it was
only ever exercised through GPU launches."""
import ctypes

import pytest

from torch_nerf.amd import _lib

DIMS = sorted(set(range(3, 70)) | set(range(90, 134)) | {159, 160, 161, 191, 192, 193, 223, 224, 225, 255, 256})
DIRS = (3, 16, 27, 31, 32, 33, 63, 64, 65, 96, 97, 128, 129, 255, 256)
ROWS = (1, 255, 256, 257, 4096 * 192)


@pytest.mark.parametrize("feat", [64, 128, 160, 256, 512])
def test_plan_of_every_network_shape(feat):
    lib = _lib.load()
    bad = []
    for pos in DIMS:
        for vd in DIRS:
            net = ctypes.byref(_lib.NetStruct(pos, vd, feat, -1, 0, -1, 0))
            for M in ROWS:
                for cus in (256, 1, 304):
                    if lib.nerf_mlp_layered_plan_check(net, M, cus) != 0:
                        bad.append(((pos, vd, feat, M, cus), lib.nerf_amd_last_error().decode()))
    assert not bad, bad[:5]


def test_plan_check_reports_what_it_refuses():
    lib = _lib.load()
    net = ctypes.byref(_lib.NetStruct(63, 27, 256, 10, 1, 4, 1))
    assert lib.nerf_mlp_layered_plan_check(net, 0, 256) == 0
    assert lib.nerf_mlp_layered_plan_check(net, -1, 256) == 1 and b"out of range" in lib.nerf_amd_last_error()
    assert lib.nerf_mlp_layered_plan_check(ctypes.byref(_lib.NetStruct(0, 27, 256, -1, 0, -1, 0)), 10, 256) == 1
    assert lib.nerf_mlp_layered_plan_check(None, 4096 * 192, 0) == 0          # NULL = the shipped network, 256 CUs
