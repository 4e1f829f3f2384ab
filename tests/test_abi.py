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
