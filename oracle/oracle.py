"""ctypes/numpy front end of ``nerf_oracle.c`` -- TEST INFRASTRUCTURE ONLY.

Parity status: pinned against the imported reference via ``tests/golden`` (see the
header of ``nerf_oracle.c``).  Each wrapper names the reference function it checks
(R/ = /root/reference/torch_nerf/src/).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# NERF_ORACLE_SO: another build of the same source (the sanitized one of `make -C oracle asan`, tests/test_oracle_sanitized.py)
_SO = os.environ.get("NERF_ORACLE_SO") or os.path.join(_HERE, "_build", "libnerf_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i64p = ctypes.POINTER(ctypes.c_int64)


def build(force: bool = False) -> str:
    """Compile the C oracle with gcc (``make -C oracle``)."""
    src = os.path.join(_HERE, "nerf_oracle.c")
    if os.environ.get("NERF_ORACLE_SO"):
        return _SO          # built by whoever named it
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.orc_aten_sum_lastdim.restype = ctypes.c_float
        _lib.orc_nerf_param_count.restype = ctypes.c_int64
        _lib.orc_exponential_lr.restype = ctypes.c_double
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _pf(a):
    return a.ctypes.data_as(_f32p)


def _pi(a):
    return a.ctypes.data_as(_i64p)


def screen_coords(H, W, pix=None):
    """R/renderer/volume_renderer.py:171-190 (rows ``pix`` of the H*W table)."""
    if pix is None:
        n, p = H * W, None
    else:
        pix = np.ascontiguousarray(pix, dtype=np.int64)
        n, p = pix.shape[0], _pi(pix)
    out = np.empty((n, 2), dtype=np.int64)
    lib().orc_screen_coords(p, ctypes.c_int64(n), ctypes.c_int64(H), ctypes.c_int64(W), _pi(out))
    return out


def raygen(coords, fx, fy, cx, cy, extrinsic):
    """R/renderer/ray_samplers/sampler_base.py:70-113,134-166."""
    coords = np.ascontiguousarray(coords, dtype=np.int64)
    ext = _f32(np.asarray(extrinsic)[:3, :4])
    n = coords.shape[0]
    o = np.empty((n, 3), np.float32)
    d = np.empty((n, 3), np.float32)
    lib().orc_raygen(_pi(coords), ctypes.c_int64(n), ctypes.c_float(fx), ctypes.c_float(fy),
                     ctypes.c_float(cx), ctypes.c_float(cy), _pf(ext), _pf(o), _pf(d))
    return o, d


def map_rays_to_ndc(focal, z_near, H, W, o, d):
    """R/renderer/ray_samplers/sampler_base.py:199-257."""
    o = _f32(o).copy()
    d = _f32(d).copy()
    lib().orc_map_rays_to_ndc(ctypes.c_double(focal), ctypes.c_double(z_near), ctypes.c_int64(H),
                              ctypes.c_int64(W), ctypes.c_int64(o.shape[0]), _pf(o), _pf(d))
    return o, d


def stratified_sample(o, d, t_bins, ps, u1):
    """R/renderer/ray_samplers/stratified_sampler.py:91-128 (coarse branch)."""
    o, d, t_bins, u1 = _f32(o), _f32(d), _f32(t_bins), _f32(u1)
    n, S = u1.shape
    t = np.empty((n, S), np.float32)
    pts = np.empty((n, S, 3), np.float32)
    dirs = np.empty((n, S, 3), np.float32)
    delta = np.empty((n, S), np.float32)
    lib().orc_stratified_sample(_pf(o), _pf(d), ctypes.c_int64(n), ctypes.c_int64(S), _pf(t_bins),
                                ctypes.c_float(np.float32(ps)), _pf(u1), _pf(t), _pf(pts),
                                _pf(dirs), _pf(delta))
    return t, pts, dirs, delta


def aten_sum_lastdim(row):
    row = _f32(row)
    return np.float32(lib().orc_aten_sum_lastdim(_pf(row), ctypes.c_int64(row.shape[0])))


def hierarchical_sample(o, d, t_bins, ps, weights, u1, u2, u3):
    """R/renderer/ray_samplers/stratified_sampler.py:57-90 + ray_samplers/utils.py:8-58.

    Returns (idx, t, pts, dirs, delta, weights_after) -- ``weights_after`` is the
    in-place ``+= 1e-5`` side effect of utils.py:31.
    """
    o, d, t_bins = _f32(o), _f32(d), _f32(t_bins)
    w = _f32(weights).copy()
    u1, u2, u3 = _f32(u1), _f32(u2), _f32(u3)
    n, Sc = u1.shape
    Sf = u2.shape[1]
    S = Sc + Sf
    idx = np.empty((n, Sf), np.int64)
    t = np.empty((n, S), np.float32)
    pts = np.empty((n, S, 3), np.float32)
    dirs = np.empty((n, S, 3), np.float32)
    delta = np.empty((n, S), np.float32)
    lib().orc_hierarchical_sample(_pf(o), _pf(d), ctypes.c_int64(n), ctypes.c_int64(Sc),
                                  ctypes.c_int64(Sf), _pf(t_bins), ctypes.c_float(np.float32(ps)),
                                  _pf(w), _pf(u1), _pf(u2), _pf(u3), _pi(idx), _pf(t), _pf(pts),
                                  _pf(dirs), _pf(delta))
    return idx, t, pts, dirs, delta, w


def posenc(x, L, include_input=True):
    """R/signal_encoder/positional_encoder.py:49-104."""
    x = _f32(x)
    M, C = x.shape
    E = 2 * L * C + (C if include_input else 0)
    out = np.empty((M, E), np.float32)
    lib().orc_posenc(_pf(x), ctypes.c_int64(M), ctypes.c_int(C), ctypes.c_int(L),
                     ctypes.c_int(1 if include_input else 0), _pf(out))
    return out


def shenc(x, degree):
    """R/signal_encoder/spherical_harmonics_encoder.py:86-139."""
    x = _f32(x)
    out = np.empty((x.shape[0], degree * degree), np.float32)
    lib().orc_shenc(_pf(x), ctypes.c_int64(x.shape[0]), ctypes.c_int(degree), _pf(out))
    return out


def shenc_backward(x, g_out, degree):
    """Gradient autograd returns for in_signal of SHEncoder.encode."""
    x, g_out = _f32(x), _f32(g_out)
    g_in = np.empty_like(x)
    lib().orc_shenc_backward(_pf(x), _pf(g_out), ctypes.c_int64(x.shape[0]), ctypes.c_int(degree), _pf(g_in))
    return g_in


def nerf_param_count(E_p=63, E_d=27, F=256):
    return int(lib().orc_nerf_param_count(ctypes.c_int(E_p), ctypes.c_int(E_d), ctypes.c_int(F)))


def mlp_forward(params, pos_enc, dir_enc, F=256):
    """R/network/nerf.py:65-121 on pre-encoded inputs; ``params`` = flat state_dict blob."""
    params, pos_enc, dir_enc = _f32(params), _f32(pos_enc), _f32(dir_enc)
    M, E_p = pos_enc.shape
    E_d = dir_enc.shape[1]
    assert params.size == nerf_param_count(E_p, E_d, F)
    sigma = np.empty((M,), np.float32)
    rgb = np.empty((M, 3), np.float32)
    lib().orc_mlp_forward(_pf(params), ctypes.c_int(E_p), ctypes.c_int(E_d), ctypes.c_int(F),
                          _pf(pos_enc), _pf(dir_enc), ctypes.c_int64(M), _pf(sigma), _pf(rgb))
    return sigma, rgb


def mlp_backward(params, pos_enc, dir_enc, g_sigma, g_rgb, F=256):
    """Parameter gradients of R/network/nerf.py:102-119 (autograd in the reference)."""
    params, pos_enc, dir_enc = _f32(params), _f32(pos_enc), _f32(dir_enc)
    g_sigma, g_rgb = _f32(g_sigma), _f32(g_rgb)
    M, E_p = pos_enc.shape
    E_d = dir_enc.shape[1]
    g = np.zeros_like(params)
    lib().orc_mlp_backward(_pf(params), ctypes.c_int(E_p), ctypes.c_int(E_d), ctypes.c_int(F),
                           _pf(pos_enc), _pf(dir_enc), ctypes.c_int64(M), _pf(g_sigma), _pf(g_rgb),
                           _pf(g))
    return g


def mlp_backward_ex(params, pos_enc, dir_enc, g_sigma, g_rgb, F=256, want_inputs=True, want_masks=False,
                    force_masks=None):
    """mlp_backward plus the gradients w.r.t. the encoded inputs (what autograd returns for `pos` / `view_dir` of
    nerf.py:65-68) and, optionally, the ReLU decisions (M, 8 F + F/2 + 1) uint8 of h0..h7, h9 and the density head.
    force_masks (same layout): differentiate with THESE decisions instead of the oracle's own (see nerf_oracle.c)."""
    params, pos_enc, dir_enc = _f32(params), _f32(pos_enc), _f32(dir_enc)
    g_sigma, g_rgb = _f32(g_sigma), _f32(g_rgb)
    M, E_p = pos_enc.shape
    E_d = dir_enc.shape[1]
    g = np.zeros_like(params)
    g_pos = np.zeros((M, E_p), np.float32) if want_inputs else None
    g_dir = np.zeros((M, E_d), np.float32) if want_inputs else None
    masks = np.zeros((M, 8 * F + F // 2 + 1), np.uint8) if want_masks else None
    if force_masks is not None:
        force_masks = np.ascontiguousarray(force_masks, dtype=np.uint8)
        assert force_masks.shape == (M, 8 * F + F // 2 + 1)
    lib().orc_mlp_backward_ex(_pf(params), ctypes.c_int(E_p), ctypes.c_int(E_d), ctypes.c_int(F),
                              _pf(pos_enc), _pf(dir_enc), ctypes.c_int64(M), _pf(g_sigma), _pf(g_rgb), _pf(g),
                              None if g_pos is None else _pf(g_pos), None if g_dir is None else _pf(g_dir),
                              None if masks is None else masks.ctypes.data_as(ctypes.c_void_p),
                              None if force_masks is None else force_masks.ctypes.data_as(ctypes.c_void_p))
    return g, g_pos, g_dir, masks


def mlp_preacts(params, pos_enc, dir_enc, F=256):
    """Pre-activations of the ten ReLUs of nerf.py:102-118 in the mask layout of mlp_backward_ex:
    (M, 8 F + F/2 + 1) = [fc_in .. fc_7 | fc_9 | fc_8[0]].  Sequential; meant for a handful of samples."""
    params, pos_enc, dir_enc = _f32(params), _f32(pos_enc), _f32(dir_enc)
    M, E_p = pos_enc.shape
    E_d = dir_enc.shape[1]
    pre = np.zeros((M, 8 * F + F // 2 + 1), np.float32)
    lib().orc_mlp_preacts(_pf(params), ctypes.c_int(E_p), ctypes.c_int(E_d), ctypes.c_int(F), _pf(pos_enc), _pf(dir_enc),
                          ctypes.c_int64(M), _pf(pre))
    return pre


def composite_forward(sigma, radiance, delta):
    """R/renderer/integrators/quadrature_integrator.py:14-67."""
    sigma, radiance, delta = _f32(sigma), _f32(radiance), _f32(delta)
    n, S = sigma.shape
    rgb = np.empty((n, 3), np.float32)
    w = np.empty((n, S), np.float32)
    lib().orc_composite_forward(_pf(sigma), _pf(radiance), _pf(delta), ctypes.c_int64(n),
                                ctypes.c_int64(S), _pf(rgb), _pf(w))
    return rgb, w


def composite_backward(sigma, radiance, delta, g_rgb, g_w=None):
    """Reverse of quadrature_integrator.py:41-65 (autograd in the reference)."""
    sigma, radiance, delta, g_rgb = _f32(sigma), _f32(radiance), _f32(delta), _f32(g_rgb)
    n, S = sigma.shape
    gs = np.empty((n, S), np.float32)
    gc = np.empty((n, S, 3), np.float32)
    gw = None if g_w is None else _f32(g_w)
    lib().orc_composite_backward(_pf(sigma), _pf(radiance), _pf(delta), _pf(g_rgb),
                                 None if gw is None else _pf(gw), ctypes.c_int64(n),
                                 ctypes.c_int64(S), _pf(gs), _pf(gc))
    return gs, gc


def adam_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam.step as configured at runners/runner_utils.py:691-695; in place on p, m, v."""
    for a in (p, g, m, v):
        assert a.dtype == np.float32 and a.flags.c_contiguous and a.shape == p.shape
    lib().orc_adam_step(_pf(p), _pf(g), _pf(m), _pf(v), ctypes.c_int64(p.size), ctypes.c_int64(step),
                        ctypes.c_double(lr), ctypes.c_double(beta1), ctypes.c_double(beta2), ctypes.c_double(eps))


def exponential_lr(init_lr, end_lr, num_iter, steps_done):
    """ExponentialLR as configured at runners/runner_utils.py:701-711, after `steps_done` scheduler steps."""
    return float(lib().orc_exponential_lr(ctypes.c_double(init_lr), ctypes.c_double(end_lr),
                                          ctypes.c_int64(num_iter), ctypes.c_int64(steps_done)))


def render_rays(params, o, d, t_bins, ps, u1, weights=None, u2=None, u3=None, L_pos=10, L_dir=4):
    """One render_scene pass on given rays (a6/a7 -> a8 -> a10 -> a11); test helper."""
    if weights is None:
        t, pts, dirs, delta = stratified_sample(o, d, t_bins, ps, u1)
        idx = None
    else:
        idx, t, pts, dirs, delta, _ = hierarchical_sample(o, d, t_bins, ps, weights, u1, u2, u3)
    n, S = delta.shape
    pe = posenc(pts.reshape(-1, 3), L_pos)
    de = posenc(dirs.reshape(-1, 3), L_dir)
    sigma, rgb = mlp_forward(params, pe, de)
    pix, w = composite_forward(sigma.reshape(n, S), rgb.reshape(n, S, 3), delta)
    return dict(idx=idx, t=t, pts=pts, dirs=dirs, delta=delta, sigma=sigma.reshape(n, S),
                radiance=rgb.reshape(n, S, 3), rgb=pix, weights=w)
