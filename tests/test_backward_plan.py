"""The fused family's backward plan on the host (no GPU): nerf_mlp_backward_plan_check walks make_plan's table the way the dW
kernels do -- every 32-row tile of every item covered exactly once by the workgroups the plan names, partial tiles packed
back to back, everything behind the gradient planes inside the workspace -- for the fp32 kernels' cost table and for the
split-f16 kernel's own (csrc/mlp_backward.hip: X2_COST_*)."""
import ctypes

import pytest

from torch_nerf.amd import _lib

SIZES = [1, 31, 32, 33, 127, 128, 129, 1000, 4096, 20001, 4096 * 64, 4096 * 192, 4096 * 192 + 1, 3_000_000]


@pytest.mark.parametrize("f16x2", [0, 1])
def test_every_tile_is_covered_once_and_the_workspace_holds_the_plan(f16x2):
    lib = _lib.load()
    nets = [None, ctypes.byref(_lib.NetStruct(63, 27, 256, 10, 1, 4, 1)), ctypes.byref(_lib.NetStruct(39, 15, 256, 6, 1, 2, 1)),
            ctypes.byref(_lib.NetStruct(60, 24, 256, 10, 0, 4, 0))]
    for net in nets:
        for cus in (1, 2, 7, 64, 104, 256, 304, 384):
            for M in SIZES:
                rc = lib.nerf_mlp_backward_plan_check(net, M, cus, f16x2)
                assert rc == 0, (M, cus, f16x2, lib.nerf_amd_last_error())


def test_plan_check_reports_what_it_refuses():
    lib = _lib.load()
    assert lib.nerf_mlp_backward_plan_check(None, 0, 256, 1) == 0
    assert lib.nerf_mlp_backward_plan_check(None, -1, 256, 0) != 0 and b"out of range" in lib.nerf_amd_last_error()
    assert lib.nerf_mlp_backward_plan_check(None, 4096, 512, 1) != 0 and b"384" in lib.nerf_amd_last_error()
    assert lib.nerf_mlp_backward_plan_check(None, 4096, 512, 0) == 0
    wide = ctypes.byref(_lib.NetStruct(75, 27, 256, 12, 1, 4, 1))         # layered family: not this backward's
    assert lib.nerf_mlp_backward_plan_check(wide, 4096, 256, 0) != 0 and b"layered" in lib.nerf_amd_last_error()
    assert lib.nerf_mlp_backward_plan_check(None, 4096 * 192, 0, 1) == 0        # cus <= 0: 256
