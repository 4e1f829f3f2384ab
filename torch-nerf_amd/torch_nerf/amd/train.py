"""Device-resident, data-parallel training step.

What the reference does per batch (runners/train.py:120-218): build the camera, choose 4096 pixels on
the host (``np.random.choice`` over all H*W inside render_scene, volume_renderer.py:122-128, or the
centre crop of train.py:145-163), render coarse then fine, gather the ground-truth pixels on the HOST
and upload them (``pixel_gt[indices].cuda()``, :180/:202), read three losses back with ``.item()``
(:183-210), backward, ``optimizer.step()``, ``scheduler.step()``.

Here the same step runs with nothing leaving the GPU:
  * ground-truth images stay in HBM (``DeviceImages``; a 100-view 800x800 set is 768 MB of 288 GB) and
    the batch's pixels are gathered there;
  * the pixel batch is chosen on the device from a seeded generator that is identical on every rank
    (uniform without replacement, like the reference; the index STREAM differs from numpy's -- exact
    index parity is kept by the drop-in ``VolumeRenderer.render_scene`` path, which still asks numpy);
  * losses are returned as device scalars -- the caller decides when (if ever) to synchronise;
  * with ``torch.distributed`` initialised the 4096-ray batch is cut into contiguous shards, one per
    rank; each rank runs coarse + fine forward/backward on its shard and ``optim.FusedAdam`` all-reduces
    the joined gradient blob once.  The loss is scaled so that the averaged gradient is exactly the
    gradient of the global mean, also for unequal shards.  Sampling draws are a function of
    (seed, step, position in the global batch), so the update does not depend on the number of GPUs
    beyond fp32 summation order.
"""
from typing import Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from . import ops
from .shard import ray_draws, shard_range

__all__ = ["DeviceImages", "choose_pixels", "centre_crop_indices", "train_step"]


class DeviceImages:
    """Ground-truth views resident in HBM: images (V, H*W, 3) fp32 on the GPU, poses (V, 4, 4) on the host
    (the ray kernel takes the camera-to-world block as 12 host floats)."""

    def __init__(self, images: torch.Tensor, poses: torch.Tensor, height: int, width: int, focal: float):
        if images.ndim == 4:                       # (V, H, W, 3) as the reference's datasets hold them
            images = images.reshape(images.shape[0], -1, images.shape[-1])
        if images.shape[1] != height * width or images.shape[2] != 3:
            raise ValueError(f"expected (V, {height * width}, 3) pixels, got {tuple(images.shape)}")
        if not images.is_cuda:
            raise RuntimeError("DeviceImages keeps the views on the GPU; move them there once")
        self.images = images.float().contiguous()
        self.poses = poses.detach().to("cpu", torch.float32)
        self.height, self.width, self.focal = int(height), int(width), float(focal)

    def __len__(self) -> int:
        return self.images.shape[0]

    def pixels(self, view: int, pixel_indices: torch.Tensor) -> torch.Tensor:
        """Rows `pixel_indices` (device int64) of view `view`: the gather of train.py:180, in HBM."""
        return self.images[view].index_select(0, pixel_indices)


def centre_crop_indices(height: int, width: int, device) -> torch.Tensor:
    """Flat indices of the central half-size window used while epoch < 10 (train.py:145-160)."""
    ci, cj = (height - 1) // 2, (width - 1) // 2
    rows = torch.arange(ci - ci // 2, ci + ci // 2, device=device)
    cols = torch.arange(cj - cj // 2, cj + cj // 2, device=device)
    return (rows[:, None] * width + cols[None, :]).reshape(-1)


def choose_pixels(height: int, width: int, count: int, generator: torch.Generator,
                  centre_crop: bool = False) -> torch.Tensor:
    """`count` distinct flat pixel indices (device int64), uniform without replacement over the image
    (volume_renderer.py:122-128) or over the centre window (train.py:145-163).  `generator` must be a
    GPU generator seeded identically on every rank so that all ranks agree on the batch."""
    device = generator.device
    if centre_crop:
        window = centre_crop_indices(height, width, device)
        return window[torch.randperm(window.numel(), device=device, generator=generator)[:count]]
    return torch.randperm(height * width, device=device, generator=generator)[:count]


def _render_pair(camera, coarse_net, fine_net, pix, n_coarse: int, n_fine: int, project_to_ndc: bool,
                 draws, sampler) -> Tuple[torch.Tensor, torch.Tensor]:
    """Coarse and fine pixel colours for the rays through `pix`, differentiable w.r.t. both networks."""
    device = pix.device
    bundle = sampler.generate_rays_from_pixels(camera, project_to_ndc, pixel_indices=pix, device=device)
    t_bins, ps = sampler._create_t_bins(camera.t_near, camera.t_far, n_coarse, device)
    u1c, u1, u2, u3 = draws
    n = pix.numel()
    pts, dirs, delta = ops.sample_stratified(bundle.ray_origin, bundle.ray_dir, t_bins, ps, u1c)
    sigma, rad = coarse_net.forward_fused(pts.view(-1, 3), dirs.view(-1, 3))
    c_rgb, c_w = ops.CompositeFunction.apply(sigma.view(n, n_coarse), rad.view(n, n_coarse, 3), delta)
    weights = c_w.detach().clone()                 # mutated in place by the sampler (utils.py:31)
    pts, dirs, delta = ops.sample_hierarchical(bundle.ray_origin, bundle.ray_dir, t_bins, ps, weights, u1, u2, u3)
    S = n_coarse + n_fine
    sigma, rad = fine_net.forward_fused(pts.view(-1, 3), dirs.view(-1, 3))
    f_rgb, _ = ops.CompositeFunction.apply(sigma.view(n, S), rad.view(n, S, 3), delta)
    return c_rgb, f_rgb


def train_step(camera, coarse_net, fine_net, optimizer, pixel_gt: torch.Tensor, pixel_indices: torch.Tensor,
               n_coarse: int, n_fine: int, project_to_ndc: bool, seed: int, step: int,
               scheduler=None, group: Optional[dist.ProcessGroup] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """One optimisation step on the GLOBAL batch `pixel_indices` (device int64, identical on every rank).

    pixel_gt: (H*W, 3) fp32 ground truth of this view on the device.  Returns this rank's
    (coarse_sse, fine_sse) -- sums of squared errors over its shard as device scalars; summed over ranks
    and divided by 3 * len(pixel_indices) they are the reference's coarse_loss / fine_loss.
    The optimizer must average gradients over ranks (optim.FusedAdam does) when world > 1."""
    from torch_nerf.src.renderer.ray_samplers import StratifiedSampler

    on = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if on else 1
    rank = dist.get_rank(group) if on else 0
    total = pixel_indices.numel()
    lo, hi = shard_range(total, rank, world)
    pix = pixel_indices[lo:hi]
    draws = ray_draws(seed * 1000003 + step, lo, hi - lo, n_coarse, n_fine, pix.device)
    optimizer.zero_grad(set_to_none=True)
    c_rgb, f_rgb = _render_pair(camera, coarse_net, fine_net, pix, n_coarse, n_fine, project_to_ndc, draws,
                                StratifiedSampler())
    gt = pixel_gt.index_select(0, pix)
    c_sse = torch.sum((c_rgb - gt) ** 2)
    f_sse = torch.sum((f_rgb - gt) ** 2)
    # MSELoss(coarse) + MSELoss(fine) over the global batch (train.py:181-204); `world` undoes the
    # optimizer's 1/world average so that shards of different length weigh correctly
    ((c_sse + f_sse) * (world / (3.0 * total))).backward()
    optimizer.step()
    if scheduler is not None:
        scheduler.step()
    return c_sse.detach(), f_sse.detach()
