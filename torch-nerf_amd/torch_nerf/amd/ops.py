"""Tensor-level wrappers over the C ABI and the torch.autograd.Function custom ops.

Everything here launches HIP kernels on torch's current stream with raw device
pointers of torch-allocated tensors.  PyTorch is plumbing (memory, streams, autograd
graph); the arithmetic is in libnerf_amd.so.  No CPU fallback exists: a CPU tensor or
a missing library raises.
"""
import ctypes
from typing import Optional, Sequence, Tuple

import torch

from . import _lib

_F32P = ctypes.POINTER(ctypes.c_float)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _gpu(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise ValueError(f"{name}: expected torch.Tensor, got {type(t)}")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU: the HIP rendering path has no CPU fallback")
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


# --------------------------------------------------------------------------- rays
def screen_coords(height: int, width: int, device, pix: Optional[torch.Tensor] = None,
                  first: int = 0, count: Optional[int] = None) -> torch.Tensor:
    """(n,2) int64 screen coordinates on `device` (volume_renderer.py:171-190)."""
    lib = _lib.load()
    if pix is not None:
        pix = _gpu(pix, "pix", torch.int64)
        n = pix.numel()
    else:
        n = height * width - first if count is None else count
    out = torch.empty((n, 2), dtype=torch.int64, device=device if pix is None else pix.device)
    with torch.cuda.device(out.device):
        _lib.check(lib.nerf_screen_coords(height, width, _ptr(pix), first, n, _ptr(out), _stream()),
                   "nerf_screen_coords")
    return out


def generate_rays(height: int, width: int, intrinsic4: Sequence[float], extrinsic: torch.Tensor,
                  project_to_ndc: bool, focal: float, z_near: float, device,
                  coords: Optional[torch.Tensor] = None, pix: Optional[torch.Tensor] = None,
                  first: int = 0, count: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Ray origins/directions (n,3) on the GPU (sampler_base.py:134-197, :199-257)."""
    lib = _lib.load()
    if coords is not None:
        coords = _gpu(coords, "coords", torch.int64)
        n, device = coords.shape[0], coords.device
    elif pix is not None:
        pix = _gpu(pix, "pix", torch.int64)
        n, device = pix.numel(), pix.device
    else:
        n = height * width - first if count is None else count
    ext = extrinsic.detach().to("cpu", torch.float32)[:3, :4].contiguous()
    ext_host = (ctypes.c_float * 12)(*ext.reshape(-1).tolist())
    o = torch.empty((n, 3), dtype=torch.float32, device=device)
    d = torch.empty((n, 3), dtype=torch.float32, device=device)
    fx, fy, cx, cy = (float(v) for v in intrinsic4)
    with torch.cuda.device(o.device):
        _lib.check(lib.nerf_generate_rays(_ptr(coords), _ptr(pix), first, n, height, width, fx, fy, cx, cy,
                                          ext_host, int(bool(project_to_ndc)), float(focal),
                                          float(z_near), _ptr(o), _ptr(d), _stream()),
                   "nerf_generate_rays")
    return o, d


# --------------------------------------------------------------------------- sampling
def sample_stratified(ray_o, ray_d, t_bins, partition_size: float, u1, want_t: bool = False):
    """Coarse branch of StratifiedSampler.sample_along_rays -> (pts, dirs, delta[, t])."""
    lib = _lib.load()
    ray_o, ray_d, t_bins, u1 = _gpu(ray_o, "ray_o"), _gpu(ray_d, "ray_d"), _gpu(t_bins, "t_bins"), _gpu(u1, "u1")
    n, S = u1.shape
    dev = u1.device
    pts = torch.empty((n, S, 3), dtype=torch.float32, device=dev)
    dirs = torch.empty((n, S, 3), dtype=torch.float32, device=dev)
    delta = torch.empty((n, S), dtype=torch.float32, device=dev)
    t = torch.empty((n, S), dtype=torch.float32, device=dev) if want_t else None
    with torch.cuda.device(dev):
        _lib.check(lib.nerf_sample_stratified(_ptr(ray_o), _ptr(ray_d), n, S, _ptr(t_bins),
                                              float(partition_size), _ptr(u1), _ptr(t), _ptr(pts),
                                              _ptr(dirs), _ptr(delta), _stream()),
                   "nerf_sample_stratified")
    return (pts, dirs, delta, t) if want_t else (pts, dirs, delta)


def sample_hierarchical(ray_o, ray_d, t_bins, partition_size: float, weights, u1, u2, u3,
                        want_idx: bool = False, want_t: bool = False):
    """Hierarchical branch + sample_pdf.  `weights` (n,Sc) fp32 GPU is mutated in place (+1e-5)."""
    lib = _lib.load()
    if not (weights.is_cuda and weights.dtype == torch.float32 and weights.is_contiguous()):
        raise RuntimeError("weights must be a contiguous fp32 GPU tensor (it is updated in place)")
    ray_o, ray_d, t_bins = _gpu(ray_o, "ray_o"), _gpu(ray_d, "ray_d"), _gpu(t_bins, "t_bins")
    u1, u2, u3 = _gpu(u1, "u1"), _gpu(u2, "u2"), _gpu(u3, "u3")
    n, Sc = u1.shape
    Sf = u2.shape[1]
    S = Sc + Sf
    dev = u1.device
    pts = torch.empty((n, S, 3), dtype=torch.float32, device=dev)
    dirs = torch.empty((n, S, 3), dtype=torch.float32, device=dev)
    delta = torch.empty((n, S), dtype=torch.float32, device=dev)
    idx = torch.empty((n, Sf), dtype=torch.int64, device=dev) if want_idx else None
    t = torch.empty((n, S), dtype=torch.float32, device=dev) if want_t else None
    with torch.cuda.device(dev):
        _lib.check(lib.nerf_sample_hierarchical(_ptr(ray_o), _ptr(ray_d), n, Sc, Sf, _ptr(t_bins),
                                                float(partition_size), _ptr(weights), _ptr(u1), _ptr(u2),
                                                _ptr(u3), _ptr(idx), _ptr(t), _ptr(pts), _ptr(dirs),
                                                _ptr(delta), _stream()),
                   "nerf_sample_hierarchical")
    out = [pts, dirs, delta]
    if want_idx:
        out.append(idx)
    if want_t:
        out.append(t)
    return tuple(out)


# --------------------------------------------------------------------------- encoding
def posenc(x: torch.Tensor, embed_level: int, include_input: bool) -> torch.Tensor:
    lib = _lib.load()
    x = _gpu(x, "in_signal")
    if x.ndim != 2:
        raise ValueError(f"Expected a 2D tensor (N, C). Got {x.ndim}-D.")
    M, C = x.shape
    E = 2 * embed_level * C + (C if include_input else 0)
    out = torch.empty((M, E), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.nerf_posenc(_ptr(x), M, C, embed_level, int(bool(include_input)), _ptr(out),
                                   _stream()), "nerf_posenc")
    return out


def posenc_backward(x: torch.Tensor, g_out: torch.Tensor, embed_level: int, include_input: bool) -> torch.Tensor:
    """Gradient autograd returns for `in_signal` of PositionalEncoder.encode (positional_encoder.py:104)."""
    lib = _lib.load()
    x, g_out = _gpu(x, "in_signal"), _gpu(g_out, "g_out")
    M, C = x.shape
    g_x = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(lib.nerf_posenc_backward(_ptr(x), _ptr(g_out), M, C, embed_level, int(bool(include_input)),
                                            _ptr(g_x), _stream()), "nerf_posenc_backward")
    return g_x


class PosencFunction(torch.autograd.Function):
    """PositionalEncoder.encode as a differentiable op (the reference's encode is plain autograd ops)."""

    @staticmethod
    def forward(ctx, x, embed_level, include_input):
        ctx.level, ctx.include = int(embed_level), bool(include_input)
        ctx.save_for_backward(x)
        return posenc(x, embed_level, include_input)

    @staticmethod
    def backward(ctx, g_out):
        (x,) = ctx.saved_tensors
        return posenc_backward(x, g_out.contiguous(), ctx.level, ctx.include), None, None


def shenc(x: torch.Tensor, degree: int) -> torch.Tensor:
    """SHEncoder.encode (spherical_harmonics_encoder.py:86-139): (M,3) -> (M, degree^2)."""
    lib = _lib.load()
    x = _gpu(x, "in_signal")
    if x.ndim != 2 or x.shape[1] != 3:
        raise ValueError(f"Expected a (N, 3) tensor. Got {tuple(x.shape)}.")
    out = torch.empty((x.shape[0], degree * degree), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.nerf_shenc(_ptr(x), x.shape[0], int(degree), _ptr(out), _stream()), "nerf_shenc")
    return out


class ShencFunction(torch.autograd.Function):
    """SHEncoder.encode as a differentiable op."""

    @staticmethod
    def forward(ctx, x, degree):
        ctx.degree = int(degree)
        ctx.save_for_backward(x)
        return shenc(x, degree)

    @staticmethod
    def backward(ctx, g_out):
        (x,) = ctx.saved_tensors
        lib = _lib.load()
        # the forward ran on the contiguous fp32 copy: the reverse kernel must see the same rows (a column slice or a
        # transposed view has other strides; empty_like would keep them for g_in as well)
        x, g_out = _gpu(x, "in_signal"), _gpu(g_out, "g_out")
        g_in = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.nerf_shenc_backward(_ptr(x), _ptr(g_out), x.shape[0], ctx.degree, _ptr(g_in), _stream()),
                       "nerf_shenc_backward")
        return g_in, None


# --------------------------------------------------------------------------- MLP
class Net:
    """Which network a call works on -- nerf_net_t of include/nerf_amd.h: NeRF(pos_dim, view_dir_dim, feat_dim)
    (network/nerf.py:24-63) plus, for the entries that encode raw points in registers, the two
    PositionalEncoder(3, levels, include_input) in front of it (levels < 0: encoders unknown to the kernels)."""

    def __init__(self, pos_dim=63, view_dir_dim=27, feat_dim=256, pos_levels=10, pos_include_input=True,
                 dir_levels=4, dir_include_input=True):
        self.key = (int(pos_dim), int(view_dir_dim), int(feat_dim), int(pos_levels), int(bool(pos_include_input)),
                    int(dir_levels), int(bool(dir_include_input)))
        self.struct = _lib.NetStruct(*self.key)
        self.pos_dim, self.view_dir_dim, self.feat_dim = self.key[:3]
        self._path = None

    @classmethod
    def dims_only(cls, pos_dim, view_dir_dim, feat_dim=256):
        return cls(pos_dim, view_dir_dim, feat_dim, -1, False, -1, False)

    @property
    def ref(self):
        return ctypes.byref(self.struct)

    @property
    def path(self) -> int:
        """_lib.PATH_FUSED (register-resident kernels) or _lib.PATH_LAYERED (one GEMM launch per layer)."""
        if self._path is None:
            self._path = int(_lib.load().nerf_mlp_path(self.ref))
            if self._path < 0:
                raise ValueError(f"invalid network description {self.key}: {_lib.load().nerf_amd_last_error().decode()}")
        return self._path

    @property
    def fused(self) -> bool:
        return self.path == _lib.PATH_FUSED

    @property
    def knows_encoders(self) -> bool:
        return self.key[3] >= 0 and self.key[5] >= 0

    @property
    def is_shipped(self) -> bool:
        return self.key == (63, 27, 256, 10, 1, 4, 1)

    @property
    def f16x2_ok(self) -> bool:
        """True if the split-f16 inference kernel (fp32-grade results on the f16 matrix pipe, csrc/mlp_forward_f16x2.hip)
        serves this network: feat_dim 256 behind two PositionalEncoders with pos_dim <= 128 and view_dir_dim <= 64 -- the
        fused family and the wider inputs of coord_encode_level 11..20 / dir_encode_level 5..10."""
        return self.feat_dim == 256 and self.pos_dim <= 128 and self.view_dir_dim <= 64 and self.knows_encoders

    @property
    def bf16_ok(self) -> bool:
        """True if the bf16-MFMA inference kernel serves this network (BASELINE configs[2]): the fused family behind two
        PositionalEncoders that share one include_input (the yaml has ONE such knob: positional_encoding.yaml:4)."""
        return self.fused and self.knows_encoders and self.key[4] == self.key[6]

    def __repr__(self):
        return f"Net{self.key}"


def _ref(net: Optional["Net"]):
    return None if net is None else net.ref


def mlp_param_count(net: Optional[Net] = None) -> int:
    return int(_lib.load().nerf_mlp_param_count(_ref(net)))


def _pack_buffer(out, nbytes, device, what):
    """`out` if it is a stream buffer of the right size on the right device (re-packed in place, stream-ordered behind
    whatever still reads it), else a fresh one."""
    if out is None:
        return torch.empty((nbytes,), dtype=torch.uint8, device=device)
    if out.dtype != torch.uint8 or out.numel() != nbytes or out.device != device or not out.is_contiguous():
        raise ValueError(f"{what}: `out` is not a {nbytes}-byte uint8 buffer on {device}")
    return out


def mlp_pack(flat_params: torch.Tensor, net: Optional[Net] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Flat state_dict blob (fp32, GPU) -> packed LDS-image stream of the fused kernels (uint8 GPU tensor; `out`: an
    earlier stream of the same network, overwritten in place)."""
    lib = _lib.load()
    flat_params = _gpu(flat_params, "flat_params")
    if flat_params.numel() != lib.nerf_mlp_param_count(_ref(net)):
        raise ValueError(f"expected {lib.nerf_mlp_param_count(_ref(net))} parameters, got {flat_params.numel()}")
    nbytes = lib.nerf_mlp_packed_bytes(_ref(net))
    if nbytes < 0:
        raise RuntimeError(f"mlp_pack: {lib.nerf_amd_last_error().decode()}")
    packed = _pack_buffer(out, nbytes, flat_params.device, "mlp_pack")
    with torch.cuda.device(flat_params.device):
        _lib.check(lib.nerf_mlp_pack(_ref(net), _ptr(flat_params), _ptr(packed), _stream()), "nerf_mlp_pack")
    return packed


# When set to a list, every MLP launch appends (tag, M, start_event, end_event) recorded on the
# launch stream: bench.py uses it to time the dominant kernel inside the timed region.
KERNEL_EVENTS = None
# what rocprofv3 calls the kernel behind the "render_pass" tag on the inference path (bench.py roofline object)
DOMINANT_KERNEL = "render_fused_kernel (sampling + posenc + 11-layer MLP + integral in one kernel)"
DOMINANT_KERNEL_BF16 = "mlp_forward_bf16_kernel (fused posenc + 11-layer MLP on v_mfma_f32_32x32x16_bf16)"
DOMINANT_KERNEL_F16X2 = ("mlp_forward_f16x2_kernel (fused posenc + 11-layer MLP, operands split in two f16 parts, three "
                         "v_mfma_f32_16x16x32_f16 per k-step)")


def _timed(tag, M):
    if KERNEL_EVENTS is None:
        return None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    KERNEL_EVENTS.append((tag, M, e0, e1))
    e0.record()
    return e1


def _check_rows(pos, view_dir, encoded, net):
    want_p, want_d = (3, 3) if not encoded else ((63, 27) if net is None else (net.pos_dim, net.view_dir_dim))
    if pos.ndim != 2 or view_dir.ndim != 2 or pos.shape[0] != view_dir.shape[0] or pos.shape[1] != want_p or \
            view_dir.shape[1] != want_d:
        raise ValueError(f"expected pos (M, {want_p}) and view_dir (M, {want_d}); got {tuple(pos.shape)}, "
                         f"{tuple(view_dir.shape)}")


def mlp_forward(packed: torch.Tensor, pos: torch.Tensor, view_dir: torch.Tensor, encoded: bool,
                save: bool = False, net: Optional[Net] = None):
    """Fused encode + NeRF forward.  Returns (sigma (M,), rgb (M,3)[, saved])."""
    lib = _lib.load()
    pos, view_dir = _gpu(pos, "pos"), _gpu(view_dir, "view_dir")
    _check_rows(pos, view_dir, encoded, net)
    M = pos.shape[0]
    sigma = torch.empty((M,), dtype=torch.float32, device=pos.device)
    rgb = torch.empty((M, 3), dtype=torch.float32, device=pos.device)
    saved = None
    if save:
        saved = torch.empty((lib.nerf_mlp_saved_bytes(_ref(net), M) // 4,), dtype=torch.float32, device=pos.device)
    with torch.cuda.device(pos.device):
        end = _timed("mlp_forward", M)
        _lib.check(lib.nerf_mlp_forward(_ref(net), _ptr(packed), _ptr(pos), _ptr(view_dir), M, int(bool(encoded)),
                                        _ptr(sigma), _ptr(rgb), _ptr(saved), _stream()),
                   "nerf_mlp_forward")
        if end is not None:
            end.record()
    return (sigma, rgb, saved) if save else (sigma, rgb)


def mlp_pack_bf16(flat_params: torch.Tensor, net: Optional[Net] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Flat fp32 state_dict blob -> bf16 fragment stream for mlp_forward_bf16 (uint8 GPU tensor; `out` as in mlp_pack)."""
    lib = _lib.load()
    flat_params = _gpu(flat_params, "flat_params")
    if flat_params.numel() != lib.nerf_mlp_param_count(_ref(net)):
        raise ValueError(f"expected {lib.nerf_mlp_param_count(_ref(net))} parameters, got {flat_params.numel()}")
    packed = _pack_buffer(out, lib.nerf_mlp_packed_bf16_bytes(_ref(net)), flat_params.device, "mlp_pack_bf16")
    with torch.cuda.device(flat_params.device):
        _lib.check(lib.nerf_mlp_pack_bf16(_ref(net), _ptr(flat_params), _ptr(packed), _stream()), "nerf_mlp_pack_bf16")
    return packed


def mlp_forward_bf16(packed_bf16: torch.Tensor, pos: torch.Tensor, view_dir: torch.Tensor, net: Optional[Net] = None):
    """Inference-only bf16-MFMA variant of the fused encode + NeRF forward; pos, view_dir raw (M,3)."""
    lib = _lib.load()
    pos, view_dir = _gpu(pos, "pos"), _gpu(view_dir, "view_dir")
    _check_rows(pos, view_dir, False, net)
    M = pos.shape[0]
    sigma = torch.empty((M,), dtype=torch.float32, device=pos.device)
    rgb = torch.empty((M, 3), dtype=torch.float32, device=pos.device)
    with torch.cuda.device(pos.device):
        end = _timed("mlp_forward_bf16", M)
        _lib.check(lib.nerf_mlp_forward_bf16(_ref(net), _ptr(packed_bf16), _ptr(pos), _ptr(view_dir), M, _ptr(sigma),
                                             _ptr(rgb), _stream()), "nerf_mlp_forward_bf16")
        if end is not None:
            end.record()
    return sigma, rgb


def mlp_pack_f16x2(flat_params: torch.Tensor, net: Optional[Net] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Flat fp32 state_dict blob -> split-f16 stream for mlp_forward_f16x2: every weight scaled by its layer's power of two
    and split in two f16 parts (uint8 GPU tensor; `out` as in mlp_pack)."""
    lib = _lib.load()
    flat_params = _gpu(flat_params, "flat_params")
    if flat_params.numel() != lib.nerf_mlp_param_count(_ref(net)):
        raise ValueError(f"expected {lib.nerf_mlp_param_count(_ref(net))} parameters, got {flat_params.numel()}")
    nbytes = lib.nerf_mlp_packed_f16x2_bytes(_ref(net))
    if nbytes < 0:
        raise RuntimeError(f"mlp_pack_f16x2: {lib.nerf_amd_last_error().decode()}")
    packed = _pack_buffer(out, nbytes, flat_params.device, "mlp_pack_f16x2")
    with torch.cuda.device(flat_params.device):
        _lib.check(lib.nerf_mlp_pack_f16x2(_ref(net), _ptr(flat_params), _ptr(packed), _stream()), "nerf_mlp_pack_f16x2")
    return packed


def mlp_forward_f16x2(packed_f16x2: torch.Tensor, pos: torch.Tensor, view_dir: torch.Tensor, net: Optional[Net] = None,
                      save: bool = False):
    """Split-f16 variant of the fused encode + NeRF forward (the fp32 bound on the f16 matrix pipe); pos, view_dir raw
    (M,3).  save=True (fused family): also returns the activation record mlp_backward reads -- the training forward."""
    lib = _lib.load()
    pos, view_dir = _gpu(pos, "pos"), _gpu(view_dir, "view_dir")
    _check_rows(pos, view_dir, False, net)
    if not (isinstance(packed_f16x2, torch.Tensor) and packed_f16x2.is_cuda and packed_f16x2.is_contiguous()
            and packed_f16x2.numel() * packed_f16x2.element_size() == lib.nerf_mlp_packed_f16x2_bytes(_ref(net))):
        raise ValueError("mlp_forward_f16x2: `packed_f16x2` is not a mlp_pack_f16x2() stream of this library")
    M = pos.shape[0]
    sigma = torch.empty((M,), dtype=torch.float32, device=pos.device)
    rgb = torch.empty((M, 3), dtype=torch.float32, device=pos.device)
    if save:
        nbytes = lib.nerf_mlp_saved_bytes(_ref(net), M)
        if nbytes < 0:
            raise RuntimeError(f"mlp_forward_f16x2(save=True): {lib.nerf_amd_last_error().decode()}")
        saved = torch.empty((nbytes // 4,), dtype=torch.float32, device=pos.device)
        with torch.cuda.device(pos.device):
            end = _timed("mlp_forward", M)        # (the training legs' accounting: this IS the record forward)
            _lib.check(lib.nerf_mlp_forward_f16x2_record(_ref(net), _ptr(packed_f16x2), _ptr(pos), _ptr(view_dir), M,
                                                         _ptr(sigma), _ptr(rgb), _ptr(saved), _stream()),
                       "nerf_mlp_forward_f16x2_record")
            if end is not None:
                end.record()
        return sigma, rgb, saved
    with torch.cuda.device(pos.device):
        end = _timed("mlp_forward_f16x2", M)
        _lib.check(lib.nerf_mlp_forward_f16x2(_ref(net), _ptr(packed_f16x2), _ptr(pos), _ptr(view_dir), M, _ptr(sigma),
                                              _ptr(rgb), _stream()), "nerf_mlp_forward_f16x2")
        if end is not None:
            end.record()
    return sigma, rgb


def mlp_backward(packed, flat_params, pos, view_dir, encoded, sigma, rgb, saved, g_sigma, g_rgb,
                 net: Optional[Net] = None, want_pos: bool = False, want_dir: bool = False, packed_f16x2=None):
    """Parameter gradients as one flat tensor in state_dict order; with want_pos / want_dir -> (g_params, g_pos,
    g_view_dir): the gradients w.r.t. the ENCODED inputs, (M, pos_dim) / (M, view_dir_dim), None where not asked.
    packed_f16x2 (a mlp_pack_f16x2 stream of the same parameters; parameter gradients only): the reverse chain and the
    dW GEMMs run on the split-f16 kernels (nerf_mlp_backward_f16x2)."""
    lib = _lib.load()
    M = pos.shape[0]
    g_sigma, g_rgb = _gpu(g_sigma, "g_sigma"), _gpu(g_rgb, "g_rgb")
    g_params = torch.empty((lib.nerf_mlp_param_count(_ref(net)),), dtype=torch.float32, device=pos.device)
    e_p, e_d = (63, 27) if net is None else (net.pos_dim, net.view_dir_dim)
    g_pos = torch.empty((M, e_p), dtype=torch.float32, device=pos.device) if want_pos else None
    g_dir = torch.empty((M, e_d), dtype=torch.float32, device=pos.device) if want_dir else None
    ws_bytes = lib.nerf_mlp_backward_workspace_bytes(_ref(net), M)
    ws = torch.empty((max(ws_bytes, 4) // 4,), dtype=torch.float32, device=pos.device)
    if packed_f16x2 is not None and not (want_pos or want_dir):
        with torch.cuda.device(pos.device):
            end = _timed("mlp_backward", M)
            _lib.check(lib.nerf_mlp_backward_f16x2(_ref(net), _ptr(packed), _ptr(packed_f16x2), M, _ptr(sigma), _ptr(rgb),
                                                   _ptr(saved), _ptr(g_sigma), _ptr(g_rgb), _ptr(g_params), _ptr(ws),
                                                   _stream()), "nerf_mlp_backward_f16x2")
            if end is not None:
                end.record()
        return g_params
    with torch.cuda.device(pos.device):
        end = _timed("mlp_backward", M)
        _lib.check(lib.nerf_mlp_backward(_ref(net), _ptr(packed), _ptr(flat_params), _ptr(pos), _ptr(view_dir), M,
                                         int(bool(encoded)), _ptr(sigma), _ptr(rgb), _ptr(saved),
                                         _ptr(g_sigma), _ptr(g_rgb), _ptr(g_params), _ptr(g_pos), _ptr(g_dir),
                                         _ptr(ws), _stream()),
                   "nerf_mlp_backward")
        if end is not None:
            end.record()
    return (g_params, g_pos, g_dir) if (want_pos or want_dir) else g_params


def _split_like(g_flat, shapes):
    grads, off = [], 0
    for shp in shapes:
        n = 1
        for s in shp:
            n *= s
        grads.append(g_flat[off:off + n].view(shp))
        off += n
    return grads


class NerfMLPFunction(torch.autograd.Function):
    """sigma, rgb = NeRF(encode(pos), encode(dir)) with hand-written forward and backward (fused family).

    apply(pos, view_dir, encoded, record, packed, flat_params, net, *params): `params` are the 22
    nn.Parameters in state_dict order (present so autograd routes their gradients);
    `flat_params` is their concatenation and `packed` its LDS-image stream; `net` an ops.Net (None = shipped).
    `record` selects the training-mode kernel that also writes the activation record for backward; the caller
    decides it from torch.is_grad_enabled() (grad mode is always off inside forward()).
    Differentiable w.r.t. the parameters AND the two inputs, like the reference's autograd graph (nerf.py:102-119):
    the dX chain returns the gradients of the encoded inputs (three thin GEMMs more, csrc/mlp_backward.hip); for RAW
    inputs (encoded=False) the reverse of PositionalEncoder.encode (csrc/posenc.hip) is chained behind them.
    """

    @staticmethod
    def forward(ctx, pos, view_dir, encoded, record, packed, flat_params, net, *params):
        # `packed`: the fp32 LDS-image stream, or (stream, split-f16 stream): with the second one given, a RECORDING forward
        # on raw points runs on the split-f16 kernel (NeRF.f16x2_training) and writes the same record; the backward is
        # the fp32 kernels' either way
        packed_x2 = None
        if isinstance(packed, tuple):
            packed, packed_x2 = packed
        need_grad = bool(record)
        ctx.encoded = bool(encoded)
        ctx.net = net
        ctx.shapes = [p.shape for p in params]
        if need_grad:
            pos, view_dir = _gpu(pos, "pos"), _gpu(view_dir, "view_dir")
            ctx.f16x2 = packed_x2 is not None and not encoded
            if ctx.f16x2:
                sigma, rgb, saved = mlp_forward_f16x2(packed_x2, pos, view_dir, net=net, save=True)
                ctx.save_for_backward(pos, view_dir, packed, flat_params, sigma, rgb, saved, packed_x2)
            else:
                sigma, rgb, saved = mlp_forward(packed, pos, view_dir, encoded, save=True, net=net)
                ctx.save_for_backward(pos, view_dir, packed, flat_params, sigma, rgb, saved)
        else:
            sigma, rgb = mlp_forward(packed, pos, view_dir, encoded, save=False, net=net)
        return sigma, rgb

    @staticmethod
    def backward(ctx, g_sigma, g_rgb):
        packed_x2 = None
        if ctx.f16x2:
            pos, view_dir, packed, flat_params, sigma, rgb, saved, packed_x2 = ctx.saved_tensors
        else:
            pos, view_dir, packed, flat_params, sigma, rgb, saved = ctx.saved_tensors
        if g_sigma is None:
            g_sigma = torch.zeros_like(sigma)
        if g_rgb is None:
            g_rgb = torch.zeros_like(rgb)
        want_pos, want_dir = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        out = mlp_backward(packed, flat_params, pos, view_dir, ctx.encoded, sigma, rgb, saved,
                           g_sigma, g_rgb, net=ctx.net, want_pos=want_pos, want_dir=want_dir, packed_f16x2=packed_x2)
        g_flat, g_pos, g_dir = out if (want_pos or want_dir) else (out, None, None)
        if not ctx.encoded:    # raw points / directions: through the encoders' reverse pass
            key = (63, 27, 256, 10, 1, 4, 1) if ctx.net is None else ctx.net.key
            if g_pos is not None:
                g_pos = posenc_backward(pos, g_pos, key[3], bool(key[4]))
            if g_dir is not None:
                g_dir = posenc_backward(view_dir, g_dir, key[5], bool(key[6]))
        return (g_pos, g_dir, None, None, None, None, None, *_split_like(g_flat, ctx.shapes))


# ---- the layered family: any NeRF(pos_dim, view_dir_dim, feat_dim), pre-encoded inputs, input gradients
LAYERED_INFERENCE_ROWS = 65536     # rows of activation scratch an inference call walks the batch with


def mlp_layered_forward(flat_params, pos, view_dir, net: Net, record: bool = False, encoded: bool = True):
    """NeRF.forward through the layered family (csrc/mlp_layered.hip): ONE persistent launch walks all eleven layers
    (register-resident for the widths reg_ok admits, plane-parked otherwise).
    encoded=True: pos (M,pos_dim), view_dir (M,view_dir_dim) pre-encoded; encoded=False: RAW points / directions (M,3)
    and `net` names both PositionalEncoders -- the encodings go straight into the kernel's input planes.
    record=True keeps the activation / ReLU-bit planes of all M rows for mlp_layered_backward; record=False is the
    inference call: only scratch for LAYERED_INFERENCE_ROWS rows, and no planes written at all on register-resident
    networks.  -> (sigma (M,), rgb (M,3)[, record tensor])."""
    lib = _lib.load()
    flat_params, pos, view_dir = _gpu(flat_params, "flat_params"), _gpu(pos, "pos"), _gpu(view_dir, "view_dir")
    _check_rows(pos, view_dir, encoded, net)
    if flat_params.numel() != lib.nerf_mlp_param_count(net.ref):
        raise ValueError(f"expected {lib.nerf_mlp_param_count(net.ref)} parameters, got {flat_params.numel()}")
    M = pos.shape[0]
    rows = M if record else min(M, LAYERED_INFERENCE_ROWS)
    sigma = torch.empty((M,), dtype=torch.float32, device=pos.device)
    rgb = torch.empty((M, 3), dtype=torch.float32, device=pos.device)
    rec = torch.empty((max(lib.nerf_mlp_layered_record_bytes(net.ref, rows), 4) // 4,), dtype=torch.float32,
                      device=pos.device)
    with torch.cuda.device(pos.device):
        end = _timed("mlp_layered_forward", M)
        _lib.check(lib.nerf_mlp_layered_forward(net.ref, _ptr(flat_params), _ptr(pos), _ptr(view_dir), M,
                                                int(bool(encoded)), _ptr(sigma), _ptr(rgb), _ptr(rec), max(rows, 1),
                                                int(bool(record)), _stream()),
                   "nerf_mlp_layered_forward")
        if end is not None:
            end.record()
    return (sigma, rgb, rec) if record else (sigma, rgb)


def mlp_layered_backward(flat_params, pos, view_dir, net: Net, sigma, rgb, rec, g_sigma, g_rgb,
                         want_pos: bool = False, want_dir: bool = False):
    """-> (g_params flat, g_pos (M,pos_dim) | None, g_view_dir (M,view_dir_dim) | None)."""
    lib = _lib.load()
    M = pos.shape[0]
    g_sigma, g_rgb = _gpu(g_sigma, "g_sigma"), _gpu(g_rgb, "g_rgb")
    dev = pos.device
    g_params = torch.empty((lib.nerf_mlp_param_count(net.ref),), dtype=torch.float32, device=dev)
    # (gradients w.r.t. the ENCODED rows, whatever `pos` / `view_dir` are -- raw (M, 3) points when the forward took those)
    g_pos = torch.empty((M, net.pos_dim), dtype=torch.float32, device=dev) if want_pos else None
    g_dir = torch.empty((M, net.view_dir_dim), dtype=torch.float32, device=dev) if want_dir else None
    ws = torch.empty((max(lib.nerf_mlp_layered_workspace_bytes(net.ref, M), 4) // 4,), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        end = _timed("mlp_layered_backward", M)
        _lib.check(lib.nerf_mlp_layered_backward(net.ref, _ptr(flat_params), _ptr(pos), _ptr(view_dir), M, _ptr(sigma),
                                                 _ptr(rgb), _ptr(rec), _ptr(g_sigma), _ptr(g_rgb), _ptr(g_params),
                                                 _ptr(g_pos), _ptr(g_dir), _ptr(ws), _stream()),
                   "nerf_mlp_layered_backward")
        if end is not None:
            end.record()
    return g_params, g_pos, g_dir


class NerfLayeredFunction(torch.autograd.Function):
    """sigma, rgb = NeRF(pos, view_dir) through the layered kernels; differentiable w.r.t. the 22 parameters AND the two
    inputs (what the reference's autograd provides, nerf.py:102-119; for RAW inputs the reverse of
    PositionalEncoder.encode is chained behind the kernels' input gradients, like NerfMLPFunction).
    apply(pos, view_dir, record, flat_params, net, encoded, *params)."""

    @staticmethod
    def forward(ctx, pos, view_dir, record, flat_params, net, encoded, *params):
        ctx.net = net
        ctx.encoded = bool(encoded)
        ctx.shapes = [p.shape for p in params]
        pos, view_dir = _gpu(pos, "pos"), _gpu(view_dir, "view_dir")
        if record:
            sigma, rgb, rec = mlp_layered_forward(flat_params, pos, view_dir, net, record=True, encoded=encoded)
            ctx.save_for_backward(pos, view_dir, flat_params, sigma, rgb, rec)
        else:
            sigma, rgb = mlp_layered_forward(flat_params, pos, view_dir, net, record=False, encoded=encoded)
        return sigma, rgb

    @staticmethod
    def backward(ctx, g_sigma, g_rgb):
        pos, view_dir, flat_params, sigma, rgb, rec = ctx.saved_tensors
        if g_sigma is None:
            g_sigma = torch.zeros_like(sigma)
        if g_rgb is None:
            g_rgb = torch.zeros_like(rgb)
        want_pos, want_dir = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        net = ctx.net
        g_flat, g_pos, g_dir = mlp_layered_backward(flat_params, pos, view_dir, net, sigma, rgb, rec, g_sigma, g_rgb,
                                                    want_pos=want_pos, want_dir=want_dir)
        if not ctx.encoded:    # the kernels returned gradients w.r.t. the encoded rows: back through the encoders
            key = net.key
            if g_pos is not None:
                g_pos = posenc_backward(pos, g_pos, key[3], bool(key[4]))
            if g_dir is not None:
                g_dir = posenc_backward(view_dir, g_dir, key[5], bool(key[6]))
        return (g_pos, g_dir, None, None, None, None, *_split_like(g_flat, ctx.shapes))


# --------------------------------------------------------------------------- counter RNG (shared with shard.py)
def counter_uniform(key: int, first: int, count: int, device) -> torch.Tensor:
    """`count` fp32 uniforms u(key, first + i) on the GPU (csrc/draws.hip); key is the 64-bit stream key."""
    lib = _lib.load()
    out = torch.empty((count,), dtype=torch.float32, device=device)
    if not out.is_cuda:
        raise RuntimeError("counter_uniform: the HIP path needs a GPU device")
    with torch.cuda.device(out.device):
        _lib.check(lib.nerf_counter_uniform(key & 0xFFFFFFFFFFFFFFFF, int(first), int(count), _ptr(out), _stream()),
                   "nerf_counter_uniform")
    return out


# --------------------------------------------------------------------------- optimizer (row f1)
def adam_step(params: torch.Tensor, grads: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor,
              step: int, lr: float, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8,
              grad_scale: float = 1.0) -> None:
    """One torch.optim.Adam step (runner_utils.py:691-695 configuration) over flat fp32 blobs, in place."""
    lib = _lib.load()
    n = params.numel()
    for name, t in (("params", params), ("grads", grads), ("exp_avg", exp_avg), ("exp_avg_sq", exp_avg_sq)):
        if not (isinstance(t, torch.Tensor) and t.is_cuda):
            raise RuntimeError(f"adam_step: {name} must live on the GPU: there is no CPU fallback")
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != n:
            raise ValueError(f"adam_step: {name} must be a contiguous fp32 blob of {n} values")
    with torch.cuda.device(params.device):
        _lib.check(lib.nerf_adam_step(_ptr(params), _ptr(grads), _ptr(exp_avg), _ptr(exp_avg_sq), n, int(step),
                                      float(lr), float(beta1), float(beta2), float(eps), float(grad_scale),
                                      _stream()), "nerf_adam_step")


# --------------------------------------------------------------------------- integrator
def composite_forward(sigma, radiance, delta):
    lib = _lib.load()
    sigma, radiance, delta = _gpu(sigma, "sigma"), _gpu(radiance, "radiance"), _gpu(delta, "delta")
    n, S = sigma.shape
    rgb = torch.empty((n, 3), dtype=torch.float32, device=sigma.device)
    w = torch.empty((n, S), dtype=torch.float32, device=sigma.device)
    with torch.cuda.device(sigma.device):
        _lib.check(lib.nerf_composite_forward(_ptr(sigma), _ptr(radiance), _ptr(delta), n, S, _ptr(rgb),
                                              _ptr(w), _stream()), "nerf_composite_forward")
    return rgb, w


def composite_backward(sigma, radiance, delta, g_rgb, g_w=None):
    lib = _lib.load()
    n, S = sigma.shape
    g_rgb = _gpu(g_rgb, "g_rgb")
    g_w = None if g_w is None else _gpu(g_w, "g_w")
    gs = torch.empty((n, S), dtype=torch.float32, device=sigma.device)
    gc = torch.empty((n, S, 3), dtype=torch.float32, device=sigma.device)
    with torch.cuda.device(sigma.device):
        _lib.check(lib.nerf_composite_backward(_ptr(sigma), _ptr(radiance), _ptr(delta), _ptr(g_rgb),
                                               _ptr(g_w), n, S, _ptr(gs), _ptr(gc), _stream()),
                   "nerf_composite_backward")
    return gs, gc


class CompositeFunction(torch.autograd.Function):
    """rgb, w = quadrature integral; differentiable w.r.t. sigma and radiance."""

    @staticmethod
    def forward(ctx, sigma, radiance, delta):
        sigma, radiance, delta = _gpu(sigma, "sigma"), _gpu(radiance, "radiance"), _gpu(delta, "delta")
        rgb, w = composite_forward(sigma, radiance, delta)
        ctx.save_for_backward(sigma, radiance, delta)
        return rgb, w

    @staticmethod
    def backward(ctx, g_rgb, g_w):
        sigma, radiance, delta = ctx.saved_tensors
        if g_rgb is None:
            g_rgb = torch.zeros((sigma.shape[0], 3), dtype=torch.float32, device=sigma.device)
        gs, gc = composite_backward(sigma, radiance, delta, g_rgb, g_w)
        return gs, gc, None


# --------------------------------------------------------------------------- fused pass
def render_is_fused(n_coarse: int, n_fine: int, fine: bool) -> bool:
    """True if render_rays runs this pass as ONE kernel (csrc/render_fused.hip)."""
    return bool(_lib.load().nerf_render_is_fused(int(n_coarse), int(n_fine), int(bool(fine))))


def render_rays(packed, ray_o, ray_d, t_bins, partition_size, u1, weights=None, u2=None, u3=None, bf16=False,
                want_idx=False, want_t=False, net: Optional[Net] = None, f16x2=False):
    """One inference render_scene pass as a single enqueue -> (rgb (n,3), weights (n,S)[, idx (n,Sf)][, t (n,S)]).
    For the sample counts of the reference's configurations (64, 64+128) that is ONE kernel: sampling, the fused
    encode + MLP and the integral, with no intermediate in HBM.  `weights` (fine pass) is floored in place.
    bf16=True: `packed` is a mlp_pack_bf16 stream and the MLP runs on the bf16 MFMA path (three launches).
    f16x2=True: `packed` is a mlp_pack_f16x2 stream and the MLP runs on the f16 matrix pipe with split operands -- the fp32
    bound at three times the fp32 kernel's speed (three launches)."""
    if bf16 and f16x2:
        raise ValueError("render_rays: bf16 and f16x2 are two different MLP kernels, pick one")
    if bf16 or f16x2:
        if want_idx or want_t:
            raise ValueError("render_rays(bf16 / f16x2) returns (rgb, weights) only: ask sample_hierarchical for idx / t")
        if weights is None:
            pts, dirs, delta = sample_stratified(ray_o, ray_d, t_bins, partition_size, u1)
        else:
            pts, dirs, delta = sample_hierarchical(ray_o, ray_d, t_bins, partition_size, weights, u1, u2, u3)
        n, S = delta.shape
        fwd = mlp_forward_f16x2 if f16x2 else mlp_forward_bf16
        sigma, rgb = fwd(packed, pts.view(n * S, 3), dirs.view(n * S, 3), net=net)
        return composite_forward(sigma.view(n, S), rgb.view(n, S, 3), delta)
    lib = _lib.load()
    ray_o, ray_d, t_bins, u1 = _gpu(ray_o, "ray_o"), _gpu(ray_d, "ray_d"), _gpu(t_bins, "t_bins"), _gpu(u1, "u1")
    n, Sc = u1.shape
    Sf = 0
    if weights is not None:
        if not (weights.is_cuda and weights.dtype == torch.float32 and weights.is_contiguous()):
            raise RuntimeError("weights must be a contiguous fp32 GPU tensor (it is updated in place)")
        u2, u3 = _gpu(u2, "u2"), _gpu(u3, "u3")
        Sf = u2.shape[1]
    # the C ABI sees raw pointers: shapes are checked HERE (the reference fails in sample_pdf / gather on a mismatch;
    # the kernel would read out of bounds and write the +1e-5 floor out of bounds)
    if u1.ndim != 2 or tuple(ray_o.shape) != (n, 3) or tuple(ray_d.shape) != (n, 3):
        raise ValueError(f"render_rays: expected ray_o, ray_d of shape ({n}, 3) for u1 {tuple(u1.shape)}; got "
                         f"{tuple(ray_o.shape)}, {tuple(ray_d.shape)}")
    if t_bins.numel() != Sc:
        raise ValueError(f"render_rays: t_bins holds {t_bins.numel()} bins, u1 has {Sc} columns")
    if weights is not None:
        if tuple(weights.shape) != (n, Sc):
            raise ValueError(f"render_rays: weights must be ({n}, {Sc}) like the coarse pass returned them; got "
                             f"{tuple(weights.shape)}")
        if tuple(u2.shape) != (n, Sf) or tuple(u3.shape) != (n, Sf):
            raise ValueError(f"render_rays: u2, u3 must both be ({n}, {Sf}); got {tuple(u2.shape)}, {tuple(u3.shape)}")
    if not (isinstance(packed, torch.Tensor) and packed.is_cuda and packed.is_contiguous()
            and packed.numel() * packed.element_size() == lib.nerf_mlp_packed_bytes(_ref(net))):
        raise ValueError("render_rays: `packed` is not a mlp_pack() stream of this library")
    S = Sc + Sf
    dev = u1.device
    rgb = torch.empty((n, 3), dtype=torch.float32, device=dev)
    w_out = torch.empty((n, S), dtype=torch.float32, device=dev)
    idx = torch.empty((n, Sf), dtype=torch.int64, device=dev) if (want_idx and weights is not None) else None
    t = torch.empty((n, S), dtype=torch.float32, device=dev) if want_t else None
    ws = None
    if not lib.nerf_render_is_fused(Sc, Sf, int(weights is not None)):
        ws = torch.empty((lib.nerf_render_workspace_bytes(n, S),), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        end = _timed("render_pass", n * S)
        _lib.check(lib.nerf_render_pass(_ref(net), _ptr(packed), _ptr(ray_o), _ptr(ray_d), n, Sc, Sf, _ptr(t_bins),
                                        float(partition_size), _ptr(weights), _ptr(u1), _ptr(u2), _ptr(u3),
                                        _ptr(rgb), _ptr(w_out), _ptr(idx), _ptr(t), _ptr(ws), _stream()),
                   "nerf_render_pass")
        if end is not None:
            end.record()
    out = [rgb, w_out]
    if want_idx:
        out.append(idx)
    if want_t:
        out.append(t)
    return tuple(out)
