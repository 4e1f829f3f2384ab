"""The eager-torch CPU port (bench.py's cpu_baseline) against the reference's golden vectors."""
import numpy as np
import torch

from oracle import torch_port as TP
from torch_nerf.amd import synth


def _params(seed):
    flat = synth.nerf_flat_params(seed=seed, sigma_bias=1.0, sigma_gain=30.0)
    return {k: torch.from_numpy(v.copy()) for k, v in synth.split_flat_params(flat).items()}


def test_port_reproduces_reference_end_to_end(golden):
    g = golden("f7_e2e")
    H, W, focal, near, far = g["meta"]
    draws = tuple(torch.from_numpy(g[k]) for k in ("u1c", "u1", "u2", "u3"))
    with torch.no_grad():
        c_rgb, c_w, f_rgb, f_w, _ = TP.render_batch(_params(3), _params(4), torch.from_numpy(g["pix"]), int(H),
                                                    int(W), float(focal), torch.from_numpy(g["pose"]),
                                                    float(near), float(far), 64, 128, draws)
    # same ATen ops in the same order on the same CPU build: bit-identical or within an ulp or two
    np.testing.assert_allclose(c_rgb.numpy(), g["coarse_rgb"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(c_w.numpy(), g["coarse_w_after"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(f_rgb.numpy(), g["fine_rgb"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(f_w.numpy(), g["fine_w"], rtol=0, atol=1e-5)
