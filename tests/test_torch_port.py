"""The eager-torch CPU port (bench.py's cpu_baseline) against the reference's golden vectors."""
import numpy as np
import pytest
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


@pytest.mark.parametrize("fixture", ["f14_train_loop", "f14_train_loop_l12_l5"])
def test_port_reproduces_reference_training_loop(golden, fixture):
    """Golden F14: 20 iterations of runners/train.py:120-218 on the imported reference.  The port runs the same ATen
    ops in the same order under torch's own autograd and Adam, so the whole trajectory -- per-step losses, pixels,
    the parameters after the last step -- must come out to rounding; this also checks that tests/helpers.py
    regenerates the fixture's inputs."""
    from helpers import f14_inputs, param_digest_error
    g = golden(fixture)
    n, steps, init_lr, end_lr, num_iter, eps = g["config"]
    n, steps = int(n), int(steps)
    levels = tuple(int(v) for v in g["levels"])
    e_p, e_d = 6 * levels[0] + 3, 6 * levels[1] + 3
    torch.set_num_threads(8)
    flats = [synth.nerf_flat_params(seed=s, pos_dim=e_p, view_dir_dim=e_d, sigma_bias=1.0, sigma_gain=30.0) for s in (3, 4)]
    nets = [{k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in synth.split_flat_params(f, e_p, e_d, 256).items()}
            for f in flats]
    params = [p for net in nets for p in net.values()]
    optimizer = torch.optim.Adam(params, lr=init_lr, eps=eps)
    scheduler = torch.optim.lr_scheduler.ExponentialLR(optimizer, pow(end_lr / init_lr, 1 / num_iter))
    mse = torch.nn.MSELoss()
    focal = float(synth.blender_focal(800))
    for step in range(steps):
        pose, pix, gt, draws = f14_inputs(step, n)
        optimizer.zero_grad()
        assert abs(optimizer.param_groups[0]["lr"] - g["lr"][step]) < 1e-12
        c_rgb, _, f_rgb, _, _ = TP.render_batch(nets[0], nets[1], torch.from_numpy(pix), 800, 800, focal,
                                                torch.from_numpy(pose), 2.0, 6.0, 64, 128,
                                                tuple(torch.from_numpy(d) for d in draws), levels=levels)
        c_loss, f_loss = mse(torch.from_numpy(gt), c_rgb), mse(torch.from_numpy(gt), f_rgb)
        assert abs(c_loss.item() - g["coarse_loss"][step]) < 2e-6 and abs(f_loss.item() - g["fine_loss"][step]) < 2e-6, step
        if step in g["keep"]:
            np.testing.assert_allclose(f_rgb.detach().numpy(), g[f"s{step}_fine_rgb"], rtol=0, atol=2e-5)
        (c_loss + f_loss).backward()
        optimizer.step()
        scheduler.step()
    for tag, net, flat in (("coarse", nets[0], flats[0]), ("fine", nets[1], flats[1])):
        now = np.concatenate([p.detach().numpy().reshape(-1) for p in net.values()])
        err = param_digest_error(now, flat, g, tag)
        # the reference against ITSELF at another thread count (another sgemm summation order): losses move by <= 3e-7,
        # the norm of the 20-step update by 1e-4, its 99th-percentile element by 0.09 and single elements by up to 0.6 of
        # the update's rms (Adam divides by sqrt(v): elements whose gradient is rounding noise are chaotic).  At the
        # fixture's 8 threads the port is bit-identical to it; the bounds cover the other thread counts
        assert err["dp_rel_rms"] < 2.0 and err["dp_p99_rel_rms"] < 0.3 and err["dp_norm_rel"] < 1e-3, (tag, err)
