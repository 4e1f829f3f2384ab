"""A procedural scene with analytic density and colour: ground truth without a dataset download.

SURVEY.md section 8, row f4: the Blender / LLFF image sets are not in this image or on the GPU box, so
"does training converge" is shown against views of an analytic field instead.  The field and its views
are DATA SYNTHESIS, done once before training with ordinary torch tensor ops (deliberately independent
of the HIP kernels under test); the orbit poses, focal length, bounds and quadrature rule are the
Blender ones (utils/data/load_blender.py:78-109, :170-176; quadrature_integrator.py:14-67).

Scene: a unit-ish sphere with a sinusoidal colour pattern and a red/white checkered box beside it,
soft surfaces (density 40 * sigmoid(-sdf / 0.03)), black background, view-independent colour.
"""
from typing import Tuple

import numpy as np
import torch

from . import synth

__all__ = ["field", "render_view", "make_views"]

_SPHERE_R = 0.75
_BOX_C = (0.95, 0.5, -0.2)
_BOX_H = 0.3


def field(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """x (..., 3) -> sigma (...,) >= 0, rgb (..., 3) in [0, 1]."""
    sd_sphere = torch.linalg.vector_norm(x, dim=-1) - _SPHERE_R
    q = (x - x.new_tensor(_BOX_C)).abs() - _BOX_H
    sd_box = torch.linalg.vector_norm(q.clamp(min=0.0), dim=-1) + q.amax(-1).clamp(max=0.0)
    in_sphere = torch.sigmoid(-sd_sphere / 0.03)
    in_box = torch.sigmoid(-sd_box / 0.03)
    sigma = 40.0 * torch.maximum(in_sphere, in_box)
    phase = x.new_tensor((0.0, 2.1, 4.2))
    c_sphere = 0.5 + 0.5 * torch.sin(5.0 * x + phase)
    checker = (torch.floor(x * 4.0).sum(-1) % 2.0).unsqueeze(-1)
    c_box = checker * x.new_tensor((0.9, 0.15, 0.1)) + (1.0 - checker) * x.new_tensor((0.95, 0.95, 0.9))
    w = (in_box / (in_sphere + in_box + 1e-12)).unsqueeze(-1)
    return sigma, (1.0 - w) * c_sphere + w * c_box


@torch.no_grad()
def render_view(pose: torch.Tensor, height: int, width: int, focal: float, t_near: float = 2.0,
                t_far: float = 6.0, samples: int = 384, rows_per_pass: int = 64) -> torch.Tensor:
    """(H*W, 3) colours of the analytic field seen from `pose` (4,4 camera-to-world, device tensor);
    rays and pixel order as the reference defines them (sampler_base.py:70-113, volume_renderer.py:171-190)."""
    dev = pose.device
    R, origin = pose[:3, :3].float(), pose[:3, 3].float()
    t = t_near + (t_far - t_near) * (torch.arange(samples, device=dev, dtype=torch.float32) + 0.5) / samples
    dt = (t_far - t_near) / samples
    out = torch.empty((height * width, 3), dtype=torch.float32, device=dev)
    for r0 in range(0, height, rows_per_pass):
        rows = torch.arange(r0, min(height, r0 + rows_per_pass), device=dev)
        idx = (rows[:, None] * width + torch.arange(width, device=dev)[None, :]).reshape(-1)
        u = (idx % width).float()
        v = ((height - 1) - idx // width).float()
        d_cam = torch.stack([(u - width / 2) / focal, (v - height / 2) / focal, -torch.ones_like(u)], -1)
        d = d_cam @ R.T
        pts = origin + t[None, :, None] * d[:, None, :]
        sigma, rgb = field(pts)
        tau = sigma * dt
        T = torch.exp(-(torch.cumsum(tau, -1) - tau))
        w = T * (1.0 - torch.exp(-tau))
        out[idx] = (w.unsqueeze(-1) * rgb).sum(1)
    return out


def make_views(n_views: int, height: int, width: int, device, phi_deg: float = -30.0, radius: float = 4.0,
               theta_offset: float = 0.0):
    """(images (V, H*W, 3) on `device`, poses (V,4,4) on host, focal): an orbit of Blender-style views."""
    focal = synth.blender_focal(width)
    thetas = np.linspace(-180.0, 180.0, n_views + 1)[:-1] + theta_offset
    poses = torch.from_numpy(np.stack([synth.pose_spherical(float(th), phi_deg, radius) for th in thetas]))
    images = torch.stack([render_view(p.to(device), height, width, focal) for p in poses])
    return images, poses, focal
