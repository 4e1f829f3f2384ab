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
    assert lib.nerf_amd_abi_version() == 5
    assert lib.nerf_mlp_param_count(None) == 595844
    # sizes only -- no compute without a GPU
    assert lib.nerf_mlp_packed_bytes(None) == 13312 + (78 + 74) * 32768   # 68 transposed chunks + 3 input-gradient pairs
    assert lib.nerf_mlp_saved_bytes(None, 128) == 128 * (2528 * 4 + 9 * 32)
    assert lib.nerf_mlp_saved_bytes(None, 129) == 256 * (2528 * 4 + 9 * 32)  # rows padded to 128


def test_network_descriptions_choose_the_kernel_family():
    """nerf_net_t -> nerf_mlp_path: the reference's constructor space NeRF(pos_dim, view_dir_dim, feat_dim)
    (network/nerf.py:24-63) and the yaml knobs in front of it (runner_utils.py:584-612).  Host logic only."""
    import ctypes
    from torch_nerf.amd import synth
    lib = _lib.load()

    def net(*v):
        return ctypes.byref(_lib.NetStruct(*v))

    assert lib.nerf_mlp_path(None) == _lib.PATH_FUSED
    for lp, ld, inc, feat, path in [(10, 4, 1, 256, 0), (6, 2, 1, 256, 0), (4, 4, 1, 256, 0), (10, 4, 0, 256, 0),
                                    (10, 4, 1, 128, 1), (12, 6, 1, 64, 1), (11, 4, 1, 256, 1), (10, 5, 0, 256, 0),
                                    (10, 5, 1, 256, 1)]:
        e_p, e_d = 6 * lp + 3 * inc, 6 * ld + 3 * inc
        n = net(e_p, e_d, feat, lp, inc, ld, inc)
        assert lib.nerf_mlp_path(n) == path, (lp, ld, inc, feat)
        assert lib.nerf_mlp_param_count(n) == synth.param_count(e_p, e_d, feat)
        assert (lib.nerf_mlp_packed_bytes(n) > 0) == (path == 0)      # the layered family packs inside its calls
        # record = constant block + forward stream + 256 padded rows of planes (two inputs, h0..h7, fc_8[1:], ReLU bit planes, h9)
        r32 = lambda v: (v + 31) // 32 * 32
        planes = 256 * 4 * (r32(e_p) + r32(e_d) + 9 * r32(feat) + r32(feat // 2))
        assert lib.nerf_mlp_layered_record_bytes(n, 10) > planes
        import ctypes as _ct
        w = _ct.c_int(0)
        assert lib.nerf_mlp_layered_plane(n, 10, 11, _ct.byref(w)) == lib.nerf_mlp_layered_record_bytes(n, 10) - 256 * 4 * r32(feat // 2)
        assert w.value == r32(feat // 2)
        assert lib.nerf_mlp_layered_workspace_bytes(n, 1000) > 0
    # encoders the kernels do not know (levels < 0): any widths, pre-encoded entries only
    assert lib.nerf_mlp_path(net(16, 16, 256, -1, 0, -1, 0)) == 0          # e.g. SHEncoder(3, 4) on both inputs
    assert lib.nerf_mlp_path(net(100, 16, 256, -1, 0, -1, 0)) == 1
    # a description that contradicts itself is refused
    assert lib.nerf_mlp_path(net(63, 27, 256, 9, 1, 4, 1)) < 0 and b"levels" in lib.nerf_amd_last_error()
    assert lib.nerf_mlp_path(net(0, 27, 256, -1, 0, -1, 0)) < 0


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


def test_entry_points_refuse_what_they_cannot_serve_before_touching_the_gpu():
    """Argument validation is host code and runs without a GPU: a network outside the fused family is refused by the
    fused entries with NERF_ERR_UNSUPPORTED (and pointed at nerf_mlp_layered_*), raw-input entries need encode
    levels, the bf16 variant needs the levels of both encoders, SH degrees beyond 5 do not exist, M = 0 is a no-op."""
    import ctypes
    lib = _lib.load()
    OK, ARG, UNSUPPORTED = 0, 1, 2
    net = lambda *v: ctypes.byref(_lib.NetStruct(*v))
    f128 = net(63, 27, 128, 10, 1, 4, 1)
    assert lib.nerf_mlp_pack(f128, None, None, None) == ARG                       # null pointers first
    one = ctypes.c_float(0.0)
    p = ctypes.cast(ctypes.pointer(one), ctypes.c_void_p)                          # any non-null address: never dereferenced
    assert lib.nerf_mlp_pack(f128, p, p, None) == UNSUPPORTED and b"nerf_mlp_layered" in lib.nerf_amd_last_error()
    assert lib.nerf_mlp_forward(f128, p, p, p, 4, 1, p, p, None, None) == UNSUPPORTED
    assert lib.nerf_mlp_backward(f128, p, p, p, p, 4, 1, p, p, p, p, p, p, None, None, p, None) == UNSUPPORTED
    assert lib.nerf_render_pass(f128, p, p, p, 4, 64, 0, p, 0.1, None, p, None, None, p, p, None, None, None, None) == UNSUPPORTED
    assert lib.nerf_mlp_packed_bytes(f128) == -1 and lib.nerf_mlp_saved_bytes(f128, 10) == -1
    sh = net(16, 16, 256, -1, 0, -1, 0)                                            # fused widths, encoders unknown
    assert lib.nerf_mlp_forward(sh, p, p, p, 4, 0, p, p, None, None) == UNSUPPORTED and b"levels" in lib.nerf_amd_last_error()
    assert lib.nerf_render_pass(sh, p, p, p, 4, 64, 0, p, 0.1, None, p, None, None, p, p, None, None, None, None) == UNSUPPORTED
    assert lib.nerf_mlp_forward_bf16(sh, p, p, p, 4, p, p, None) == UNSUPPORTED and b"levels" in lib.nerf_amd_last_error()
    mixed = net(39, 12, 256, 6, 1, 2, 0)                                            # include_input differs between the two
    assert lib.nerf_mlp_forward_bf16(mixed, p, p, p, 4, p, p, None) == UNSUPPORTED
    assert lib.nerf_mlp_forward_bf16(f128, p, p, p, 4, p, p, None) == UNSUPPORTED   # bf16: the fused family only
    other = net(39, 15, 256, 6, 1, 2, 1)
    assert lib.nerf_mlp_packed_bf16_bytes(other) > 0                                # run-time levels: served since round 4
    # split-f16 variant (ABI v5): the fused family behind PositionalEncoders, mixed include_input allowed
    assert lib.nerf_mlp_forward_f16x2(sh, p, p, p, 4, p, p, None) == UNSUPPORTED and b"levels" in lib.nerf_amd_last_error()
    assert lib.nerf_mlp_forward_f16x2(f128, p, p, p, 4, p, p, None) == UNSUPPORTED
    assert lib.nerf_mlp_packed_f16x2_bytes(f128) == -1 and lib.nerf_mlp_packed_f16x2_bytes(other) == 13312 + (73 + 68) * 32768   # forward + transposed stream
    wide = net(75, 33, 256, 12, 1, 5, 1)                                            # three position + two direction k-blocks
    assert lib.nerf_mlp_packed_f16x2_bytes(wide) == 13312 + (2 * 3 + 64 + 5) * 32768
    assert lib.nerf_mlp_packed_f16x2_bytes(net(129, 27, 256, -1, 0, -1, 0)) == -1 and b"split-f16" in lib.nerf_amd_last_error()
    assert lib.nerf_mlp_pack_f16x2(None, None, None, None) == ARG
    assert lib.nerf_mlp_forward_f16x2(None, p, p, p, 0, p, p, None) == OK and lib.nerf_mlp_forward_f16x2(None, p, p, p, -1, p, p, None) == ARG
    bad = net(63, 27, 256, 9, 1, 4, 1)
    assert lib.nerf_mlp_forward(bad, p, p, p, 4, 1, p, p, None, None) == ARG
    assert lib.nerf_mlp_layered_forward(bad, p, p, p, 4, 1, p, p, p, 4, 0, None) == ARG
    unknown = net(75, 27, 256, -1, 0, -1, 0)                                        # layered widths, encoders unknown: no raw entry
    assert lib.nerf_mlp_layered_forward(unknown, p, p, p, 4, 0, p, p, p, 4, 0, None) == UNSUPPORTED and b"levels" in lib.nerf_amd_last_error()
    assert lib.nerf_shenc(p, 4, 6, p, None) == ARG and lib.nerf_shenc(p, 4, 0, p, None) == ARG
    assert lib.nerf_shenc_backward(p, p, 4, 9, p, None) == ARG
    # M = 0: nothing to do, nothing launched
    assert lib.nerf_mlp_forward(None, p, p, p, 0, 1, p, p, None, None) == OK
    assert lib.nerf_mlp_layered_forward(f128, p, p, p, 0, 1, p, p, p, 4, 0, None) == OK
    assert lib.nerf_mlp_layered_forward(f128, p, p, p, 8, 1, p, p, p, 4, 1, None) == ARG        # a kept record must hold every row
    assert lib.nerf_shenc(p, 0, 4, p, None) == OK and lib.nerf_posenc_backward(p, p, 0, 3, 10, 1, p, None) == OK
    assert lib.nerf_mlp_forward(None, p, p, p, -1, 1, p, p, None, None) == ARG
