"""Shared test helpers."""
import numpy as np

from torch_nerf.amd import synth


def check_grad_digest(flat_grad, g, prefix, rtol, atol_scale, norm_rtol=1e-4):
    """Compare a flat gradient blob with the per-tensor digest stored in a golden file
    (norm, leading slice, strided sample; see tests/golden/make_golden.py:grad_digest)."""
    grads = synth.split_flat_params(np.asarray(flat_grad, np.float32))
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
