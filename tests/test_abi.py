"""The C-ABI library loads on a CPU-only box and exports every symbol include/nerf_amd.h declares."""
import os
import re

from torch_nerf.amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "nerf_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(nerf_[a-z0-9_]+)\s*\(", text))


def test_library_builds_loads_and_exports_header():
    _lib.build()
    lib = _lib.load()
    declared = _header_symbols()
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.nerf_amd_abi_version() == 1
    assert lib.nerf_mlp_param_count() == 595844
    # sizes only -- no compute without a GPU
    assert lib.nerf_mlp_packed_bytes() == 13312 + (78 + 68) * 32768
    assert lib.nerf_mlp_saved_bytes(128) == 128 * (2528 * 4 + 9 * 32)
    assert lib.nerf_mlp_saved_bytes(129) == 256 * (2528 * 4 + 9 * 32)  # rows padded to 128


def test_plane_layout_is_a_bijection_with_coalesced_stores_and_conflict_free_fragments():
    """The TF layout of the activation record / gradient planes (csrc/mlp_layout.h), through the host-side
    nerf_mlp_plane_offset: (1) a 32-sample tile is a permutation of its 32*W floats; (2) what one wavefront
    store instruction writes -- the 4-feature group (fb, q) of 32 samples x 2 lane halves -- is one contiguous
    1 KiB; (3) the dW kernel's fragment read (a sample pair x 32 consecutive features, one float per lane)
    touches 32 different LDS banks per half-wave."""
    import numpy as np
    lib = _lib.load()
    for W in (256, 128, 64, 32):
        off = np.array([[lib.nerf_mlp_plane_offset(W, 64 + m, k) for k in range(W)] for m in range(32)])
        assert sorted(off.reshape(-1).tolist()) == list(range(2 * 32 * W, 3 * 32 * W))   # tile 2 of the plane
        for fb in range(W // 32):
            for q in range(4):
                grp = np.array([[off[i, 32 * fb + 8 * q + 4 * h + e] for e in range(4)] for h in range(2) for i in range(32)])
                assert grp.max() - grp.min() == 255 and len(set(grp.reshape(-1).tolist())) == 256
                assert grp.min() % 256 == 0
        for s in range(16):
            for fb in range(W // 32):
                for h in range(2):                       # lanes (i, h): sample 2s + h, feature 32 fb + i
                    banks = {int(off[2 * s + h, 32 * fb + i]) % 32 for i in range(32)}
                    assert len(banks) == 32
    assert lib.nerf_mlp_plane_offset(100, 0, 0) == -1 and lib.nerf_mlp_plane_offset(256, 0, 256) == -1


def test_render_is_fused_is_a_host_side_rule():
    """Which sample counts run as the single-kernel pass (csrc/render_fused.hip): G in {1, 2, 4} rays whose samples are
    a whole number of 128-sample tiles, rows fitting the LDS left beside the weight ring.  Pure host logic."""
    lib = _lib.load()
    for sc, sf, fine, want in [(64, 128, 0, 1), (64, 128, 1, 1), (64, 0, 0, 1), (128, 64, 1, 1), (64, 64, 1, 1),
                               (32, 0, 0, 1), (32, 64, 1, 1), (64, 192, 1, 1),
                               (40, 0, 0, 0), (40, 24, 1, 1), (1000, 16, 1, 0), (100, 0, 0, 0), (0, 0, 0, 0), (64, -1, 1, 0)]:
        assert lib.nerf_render_is_fused(sc, sf, fine) == want, (sc, sf, fine)
