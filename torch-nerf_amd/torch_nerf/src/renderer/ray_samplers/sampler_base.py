"""Ray bundle + ray generation on the GPU.

API of torch_nerf/src/renderer/ray_samplers/sampler_base.py (RayBundle :11-59,
RaySamplerBase.generate_rays :134-197, map_rays_to_ndc :199-257).  The arithmetic is
csrc/rays.hip; rays are produced directly in HBM (the reference builds them on the CPU
and copies them over in sample_along_rays).
"""
from typing import Optional, Tuple

import torch

from torch_nerf.amd import ops
import torch_nerf.src.renderer.cameras as cameras

__all__ = ["RayBundle", "RaySamplerBase"]


class RayBundle(object):
    """Origins (N,3), directions (N,3), near/far bounds and the NDC flag of a set of rays."""

    def __init__(self, ray_origin: torch.Tensor, ray_dir: torch.Tensor, t_near: float, t_far: float,
                 is_ndc: bool):
        self._ray_origin, self._ray_dir = ray_origin, ray_dir
        self._t_near, self._t_far, self._is_ndc = t_near, t_far, is_ndc

    ray_origin = property(lambda self: self._ray_origin)
    ray_dir = property(lambda self: self._ray_dir)
    t_near = property(lambda self: self._t_near)
    t_far = property(lambda self: self._t_far)
    is_ndc = property(lambda self: self._is_ndc)


def _current_device() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("ray generation runs on the GPU (HIP kernels); no GPU is visible")
    return torch.device("cuda", torch.cuda.current_device())


class RaySamplerBase(object):
    def __init__(self):
        pass

    @staticmethod
    def _camera_numbers(camera: cameras.PerspectiveCamera, project_to_ndc: bool):
        K = camera.intrinsic
        k4 = (float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]))
        focal = camera.focal_lengths[0]
        if project_to_ndc:
            fx, fy = camera.focal_lengths
            if fx != fy:
                raise ValueError(
                    "Focal length used for computing NDC is ambiguous."
                    f"Two different focal lengths ({fx}, {fy}) exists but only one can be used.")
            if camera.t_near < 0:
                raise ValueError(
                    f"Expected a real number greater than or equal to 0. Got {camera.t_near}.")
        return k4, focal

    def generate_rays(self, pixel_coords: torch.Tensor, camera: cameras.PerspectiveCamera,
                      project_to_ndc: bool) -> RayBundle:
        """Rays through the given screen coordinates (N,2): d = ((u-cx)/fx, (v-cy)/fy, -1) R^T, o = t."""
        k4, focal = self._camera_numbers(camera, project_to_ndc)
        device = pixel_coords.device if pixel_coords.is_cuda else _current_device()
        o, d = ops.generate_rays(camera.img_height, camera.img_width, k4, camera.extrinsic,
                                 project_to_ndc, focal, camera.t_near, device,
                                 coords=pixel_coords.to(device))
        return RayBundle(o, d, t_near=camera.t_near, t_far=camera.t_far, is_ndc=project_to_ndc)

    def generate_rays_from_pixels(self, camera: cameras.PerspectiveCamera, project_to_ndc: bool,
                                  pixel_indices: Optional[torch.Tensor] = None, first: int = 0,
                                  count: Optional[int] = None, device=None) -> RayBundle:
        """Same rays, addressed by flat pixel index (or a contiguous range): the screen-coordinate
        table of volume_renderer.py:171-190 is evaluated inside the kernel instead of being built."""
        k4, focal = self._camera_numbers(camera, project_to_ndc)
        device = _current_device() if device is None else torch.device(device)
        pix = None
        if pixel_indices is not None:
            pix = pixel_indices
            if not pix.is_cuda and not pix.is_pinned():
                # the runners hand over a pageable CPU tensor (np.random.choice).  A pageable host-to-device copy
                # waits for the stream to drain before it starts, i.e. for the previous step's backward pass, and only
                # then can this step's first kernel be queued; through a pinned staging buffer the copy and everything
                # behind it is simply queued (32 KB memcpy; the pinned block is recycled by torch's host allocator
                # once the copy has run)
                pix = pix.to(torch.int64).pin_memory()
            pix = pix.to(device, torch.int64, non_blocking=True)
        o, d = ops.generate_rays(camera.img_height, camera.img_width, k4, camera.extrinsic,
                                 project_to_ndc, focal, camera.t_near, device, pix=pix, first=first,
                                 count=count)
        return RayBundle(o, d, t_near=camera.t_near, t_far=camera.t_far, is_ndc=project_to_ndc)

    def map_rays_to_ndc(self, focal_length: float, z_near: float, img_height: int, img_width: int,
                        ray_origin: torch.Tensor, ray_dir: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """NDC projection of world-frame rays (closed form; small, evaluated with torch ops)."""
        if z_near < 0:
            raise ValueError(f"Expected a real number greater than or equal to 0. Got {z_near}.")
        sx, sy = -(2 * focal_length / img_width), -(2 * focal_length / img_height)
        oz = ray_origin[:, 2]
        ox_z, oy_z = ray_origin[:, 0] / oz, ray_origin[:, 1] / oz
        origin = torch.stack([sx * ox_z, sy * oy_z, 1 + (2 * z_near / oz)], dim=-1)
        direction = torch.stack([sx * ((ray_dir[:, 0] / ray_dir[:, 2]) - ox_z),
                                 sy * ((ray_dir[:, 1] / ray_dir[:, 2]) - oy_z),
                                 -(2 * z_near / oz)], dim=-1)
        return origin, direction

    def sample_along_rays(self, *args, **kwargs) -> torch.Tensor:
        raise NotImplementedError()
