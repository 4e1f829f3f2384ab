"""Eager-PyTorch CPU restatement of the reference's rendering path -- TEST INFRASTRUCTURE ONLY.

Purpose: the `cpu_baseline` leg of bench.py.  The reference's own path is eager PyTorch on
whatever device it is given; its Python files cannot travel to the GPU box, so this module
restates the same op sequence (own code, functional style) and is timed with device='cpu'
on the GPU box's host cores.  It is validated against the golden vectors captured from
the imported reference in tests/test_torch_port.py (fixtures F7), so timing it is timing
the reference's arithmetic: same ATen ops, same shapes, same materialised intermediates
((N,S,3) points and directions, (M,63)/(M,27) encodings, per-layer (M,256) activations).

Parity status: pinned (tests/golden/f7_e2e.npz).  Never imported by torch-nerf_amd/.
Citations: R/ = /root/reference/torch_nerf/src/.
"""
import torch
import torch.nn.functional as F

LAYERS = ("fc_in", "fc_1", "fc_2", "fc_3", "fc_4", "fc_5", "fc_6", "fc_7", "fc_8", "fc_9", "fc_out")


def encode(x, levels):
    """R/signal_encoder/positional_encoder.py:84-104 (include_input=True)."""
    feats = [x]
    for lv in range(levels):
        f = float(2 ** lv)
        feats += [torch.sin(f * x), torch.cos(f * x)]
    return torch.cat(feats, -1)


def mlp(p, pos, view):
    """R/network/nerf.py:102-119; `p` maps 'fc_k.weight' / 'fc_k.bias' to tensors."""
    def lin(name, x):
        return F.linear(x, p[name + ".weight"], p[name + ".bias"])

    x = torch.relu(lin("fc_in", pos))
    for name in ("fc_1", "fc_2", "fc_3", "fc_4"):
        x = torch.relu(lin(name, x))
    x = torch.cat([pos, x], -1)
    for name in ("fc_5", "fc_6", "fc_7"):
        x = torch.relu(lin(name, x))
    x = lin("fc_8", x)
    sigma = torch.relu(x[:, 0])
    x = torch.relu(lin("fc_9", torch.cat([x[:, 1:], view], -1)))
    return sigma, torch.sigmoid(lin("fc_out", x))


def integrate(sigma, radiance, delta):
    """R/renderer/integrators/quadrature_integrator.py:41-65."""
    tau = sigma * delta
    zero = torch.zeros((sigma.shape[0], 1))
    trans = torch.exp(-torch.cumsum(torch.cat([zero, tau], -1), -1)[..., :-1])
    w = trans * (1.0 - torch.exp(-tau))
    return torch.sum(w.unsqueeze(-1) * radiance, 1), w


def sample(o, d, near, far, n_coarse, u1, weights=None, u2=None, u3=None):
    """R/renderer/ray_samplers/stratified_sampler.py:57-126 + utils.py:31-56 with explicit draws."""
    n = o.shape[0]
    bins = torch.linspace(near, far, n_coarse + 1)[:-1].unsqueeze(0).repeat(n, 1)
    ps = (far - near) / n_coarse
    t = bins + ps * u1
    idx = None
    if weights is not None:
        weights += 1e-5
        pdf = weights / torch.sum(weights, -1, keepdim=True)
        cdf = torch.cumsum(pdf, -1)
        cdf = torch.cat([torch.zeros((n, 1)), cdf[..., :-1]], -1)
        idx = torch.searchsorted(cdf, u2.contiguous(), right=True) - 1
        t_fine = torch.gather(bins, 1, idx) + ps * u3
        t, _ = torch.sort(torch.cat([t, t_fine], -1), -1)
    delta = torch.diff(torch.cat([t, 1e8 * torch.ones((n, 1))], -1), n=1, dim=-1)
    s = t.shape[1]
    dirs = d.unsqueeze(1).repeat(1, s, 1)
    pts = o.unsqueeze(1).repeat(1, s, 1) + t.unsqueeze(-1) * dirs
    return pts, dirs, delta, idx


def rays(pix, height, width, focal, pose):
    """R/renderer/volume_renderer.py:179-188 + ray_samplers/sampler_base.py:92-103,164-165."""
    u = (pix % width).float()
    v = ((height - 1) - torch.div(pix, width, rounding_mode="floor")).float()
    cam = torch.stack([(u - width / 2.0) / focal, (v - height / 2.0) / focal, -torch.ones_like(u)], -1)
    return torch.zeros_like(cam) + pose[:3, -1], cam @ pose[:3, :3].t()


def render_pass(p, o, d, near, far, n_coarse, u1, weights=None, u2=None, u3=None, levels=(10, 4)):
    """One render_scene call (R/renderer/volume_renderer.py:136-169), single ray batch.  `levels` = (coord_encode_level,
    dir_encode_level) of the two PositionalEncoders (shipped yaml: 10 / 4)."""
    pts, dirs, delta, idx = sample(o, d, near, far, n_coarse, u1, weights, u2, u3)
    n, s, _ = pts.shape
    sigma, rgb = mlp(p, encode(pts.reshape(n * s, 3), levels[0]), encode(dirs.reshape(n * s, 3), levels[1]))
    pix, w = integrate(sigma.reshape(n, s), rgb.reshape(n, s, 3), delta)
    # the reference returns torch.cat over its ray batches (volume_renderer.py:256-259): a NEW tensor, which is
    # why the fine pass may then do `weights += 1e-5` in place without invalidating the coarse graph
    return torch.cat([pix], 0), torch.cat([w], 0), idx


def render_batch(p_coarse, p_fine, pix, height, width, focal, pose, near, far, n_coarse, n_fine, draws, levels=(10, 4)):
    """Coarse + fine pass for one pixel batch, as runners/train.py:172-201 issues them."""
    u1c, u1, u2, u3 = draws
    o, d = rays(pix, height, width, focal, pose)
    c_rgb, c_w, _ = render_pass(p_coarse, o, d, near, far, n_coarse, u1c, levels=levels)
    f_rgb, f_w, idx = render_pass(p_fine, o, d, near, far, n_coarse, u1, c_w, u2, u3, levels=levels)
    return c_rgb, c_w, f_rgb, f_w, idx
