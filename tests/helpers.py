"""Shared test helpers."""
import numpy as np

from torch_nerf.amd import synth


def check_grad_digest(flat_grad, g, prefix, rtol, atol_scale, norm_rtol=1e-4, dims=(63, 27, 256)):
    """Compare a flat gradient blob with the per-tensor digest stored in a golden file
    (norm, leading slice, strided sample; see tests/golden/make_golden.py:grad_digest)."""
    grads = synth.split_flat_params(np.asarray(flat_grad, np.float32), *dims)
    for k, v in grads.items():
        v = v.reshape(-1)
        norm_ref = float(g[prefix + k + ".norm"][0])
        atol = atol_scale * max(norm_ref / np.sqrt(v.size), 1e-12)
        head = g[prefix + k + ".head"]
        strided = g[prefix + k + ".stride"]
        np.testing.assert_allclose(v[:head.size], head, rtol=rtol, atol=atol, err_msg=k)
        step = max(1, v.size // 192)
        np.testing.assert_allclose(v[::step][:strided.size], strided, rtol=rtol, atol=atol, err_msg=k)
        norm = np.sqrt(np.sum(v.astype(np.float64) ** 2))
        assert abs(norm - norm_ref) <= norm_rtol * norm_ref + 1e-12, (k, norm, norm_ref)


# tests/golden/f11_net_variants.npz: tag -> (coord_encode_level, dir_encode_level, include_input, feat_dim)
NET_VARIANTS = {"l6_l2": (6, 2, True, 256), "l4_l4": (4, 4, True, 256), "l10_l4_noinput": (10, 4, False, 256),
                "l10_l4_f128": (10, 4, True, 128), "l12_l6_f64": (12, 6, True, 64)}


def variant_params(g, tag):
    """(flat parameter blob, (pos_dim, view_dir_dim, feat_dim)) of one F11 variant, as make_golden.py built it."""
    e_p, e_d, feat = (int(v) for v in g[tag + "_dims"][:3])
    return synth.nerf_flat_params(seed=5, pos_dim=e_p, view_dir_dim=e_d, feat_dim=feat, sigma_bias=0.5,
                                  sigma_gain=4.0), (e_p, e_d, feat)
