"""ctypes binding of libnerf_amd.so (C ABI: include/nerf_amd.h).

The library is built in-tree (``make -C torch-nerf_amd/csrc``) so that it travels with
the source tree.  There is no CPU fallback: if the library is missing, or a tensor is
not on a GPU, the ops raise.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(os.path.dirname(_HERE))            # .../torch-nerf_amd
CSRC = os.path.join(PKG_ROOT, "csrc")
LIB_PATH = os.environ.get("NERF_AMD_LIB", os.path.join(PKG_ROOT, "lib", "libnerf_amd.so"))

_c_i64 = ctypes.c_int64
_c_int = ctypes.c_int
_c_f = ctypes.c_float
_c_d = ctypes.c_double
_p = ctypes.c_void_p

# name -> (restype, argtypes); mirrors include/nerf_amd.h one to one
SIGNATURES = {
    "nerf_amd_abi_version": (_c_int, []),
    "nerf_amd_last_error": (ctypes.c_char_p, []),
    "nerf_screen_coords": (_c_int, [_c_i64, _c_i64, _p, _c_i64, _c_i64, _p, _p]),
    "nerf_generate_rays": (_c_int, [_p, _p, _c_i64, _c_i64, _c_i64, _c_i64, _c_f, _c_f, _c_f, _c_f,
                                    ctypes.POINTER(_c_f), _c_int, _c_d, _c_d, _p, _p, _p]),
    "nerf_sample_stratified": (_c_int, [_p, _p, _c_i64, _c_int, _p, _c_f, _p, _p, _p, _p, _p, _p]),
    "nerf_sample_hierarchical": (_c_int, [_p, _p, _c_i64, _c_int, _c_int, _p, _c_f, _p, _p, _p, _p, _p,
                                          _p, _p, _p, _p, _p]),
    "nerf_posenc": (_c_int, [_p, _c_i64, _c_int, _c_int, _c_int, _p, _p]),
    # every nerf_mlp_* / nerf_render_* entry starts with `const nerf_net_t *net` (NULL = the shipped 63 / 27 / 256)
    "nerf_mlp_path": (_c_int, [_p]),
    "nerf_mlp_param_count": (_c_i64, [_p]),
    "nerf_mlp_packed_bytes": (_c_i64, [_p]),
    "nerf_mlp_pack": (_c_int, [_p, _p, _p, _p]),
    "nerf_mlp_plane_offset": (_c_i64, [_c_int, _c_i64, _c_int]),
    "nerf_mlp_saved_bytes": (_c_i64, [_p, _c_i64]),
    "nerf_mlp_forward": (_c_int, [_p, _p, _p, _p, _c_i64, _c_int, _p, _p, _p, _p]),
    "nerf_mlp_packed_bf16_bytes": (_c_i64, [_p]),
    "nerf_mlp_pack_bf16": (_c_int, [_p, _p, _p, _p]),
    "nerf_mlp_forward_bf16": (_c_int, [_p, _p, _p, _p, _c_i64, _p, _p, _p]),
    "nerf_mlp_packed_f16x2_bytes": (_c_i64, [_p]),
    "nerf_mlp_pack_f16x2": (_c_int, [_p, _p, _p, _p]),
    "nerf_mlp_forward_f16x2": (_c_int, [_p, _p, _p, _p, _c_i64, _p, _p, _p]),
    "nerf_mlp_forward_f16x2_record": (_c_int, [_p, _p, _p, _p, _c_i64, _p, _p, _p, _p]),
    "nerf_mlp_backward_f16x2": (_c_int, [_p, _p, _p, _c_i64, _p, _p, _p, _p, _p, _p, _p, _p]),
    "nerf_mlp_backward_workspace_bytes": (_c_i64, [_p, _c_i64]),
    "nerf_mlp_backward": (_c_int, [_p, _p, _p, _p, _p, _c_i64, _c_int, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "nerf_mlp_layered_record_bytes": (_c_i64, [_p, _c_i64]),
    "nerf_mlp_layered_workspace_bytes": (_c_i64, [_p, _c_i64]),
    "nerf_mlp_layered_plane": (_c_i64, [_p, _c_i64, _c_int, ctypes.POINTER(_c_int)]),
    "nerf_mlp_layered_plan_check": (_c_int, [_p, _c_i64, _c_int]),
    "nerf_mlp_backward_plan_check": (_c_int, [_p, _c_i64, _c_int, _c_int]),
    "nerf_mlp_layered_forward": (_c_int, [_p, _p, _p, _p, _c_i64, _c_int, _p, _p, _p, _c_i64, _c_int, _p]),
    "nerf_mlp_layered_backward": (_c_int, [_p, _p, _p, _p, _c_i64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "nerf_shenc": (_c_int, [_p, _c_i64, _c_int, _p, _p]),
    "nerf_shenc_backward": (_c_int, [_p, _p, _c_i64, _c_int, _p, _p]),
    "nerf_posenc_backward": (_c_int, [_p, _p, _c_i64, _c_int, _c_int, _c_int, _p, _p]),
    "nerf_composite_forward": (_c_int, [_p, _p, _p, _c_i64, _c_int, _p, _p, _p]),
    "nerf_composite_backward": (_c_int, [_p, _p, _p, _p, _p, _c_i64, _c_int, _p, _p, _p]),
    "nerf_render_is_fused": (_c_int, [_c_int, _c_int, _c_int]),
    "nerf_render_workspace_bytes": (_c_i64, [_c_i64, _c_int]),
    "nerf_render_pass": (_c_int, [_p, _p, _p, _p, _c_i64, _c_int, _c_int, _p, _c_f, _p, _p, _p, _p, _p, _p, _p, _p,
                                  _p, _p]),
    "nerf_render_rays": (_c_int, [_p, _p, _p, _p, _c_i64, _c_int, _c_int, _p, _c_f, _p, _p, _p, _p, _p, _p,
                                  _p, _p]),
    "nerf_counter_uniform": (_c_int, [ctypes.c_uint64, _c_i64, _c_i64, _p, _p]),
    "nerf_adam_step": (_c_int, [_p, _p, _p, _p, _c_i64, _c_i64, _c_d, _c_d, _c_d, _c_d, _c_d, _p]),
}



class NetStruct(ctypes.Structure):
    """nerf_net_t (include/nerf_amd.h)."""
    _fields_ = [("pos_dim", ctypes.c_int32), ("view_dir_dim", ctypes.c_int32), ("feat_dim", ctypes.c_int32),
                ("pos_levels", ctypes.c_int32), ("pos_include_input", ctypes.c_int32),
                ("dir_levels", ctypes.c_int32), ("dir_include_input", ctypes.c_int32)]


PATH_FUSED, PATH_LAYERED = 0, 1

_lib = None


def build(force: bool = False) -> str:
    """Compile libnerf_amd.so for gfx950 with hipcc (works without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j8"] + (["-B"] if force else [])
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return LIB_PATH


def audit() -> None:
    """Static ISA audit of the kernels that use hand-issued LDS reads / inline-asm VMEM (make -C csrc audit)."""
    subprocess.check_call(["make", "-C", CSRC, "-j8", "audit"], stdout=subprocess.DEVNULL)


def load():
    """Load the shared library (after torch, so both share one HIP runtime)."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  -- torch's bundled libamdhip64 must be resident first
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `make -C {CSRC}` (or __graft_entry__.build()). "
            "There is no CPU fallback for the rendering path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.nerf_amd_abi_version() != 5:
        raise RuntimeError("libnerf_amd.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().nerf_amd_last_error()
        raise RuntimeError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")
