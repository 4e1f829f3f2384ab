"""Stratified + hierarchical sampling along rays, one HIP kernel per call.

Interface and draw order of torch_nerf/src/renderer/ray_samplers/stratified_sampler.py
(:17-128 sample_along_rays, :130-164 _create_t_bins).  The random numbers are drawn here
with the same torch calls, in the same order and shapes as the reference
(U1 rand (N,Sc) -> U2 rand (N,Sf) -> U3 rand (N,Sf)), so a seed reproduces its stream on
the same device; the kernels take the uniforms as inputs.
"""
from typing import Tuple, Union

import torch

from torch_nerf.amd import ops

from torch_nerf.src.renderer.ray_samplers.sampler_base import RayBundle, RaySamplerBase
from torch_nerf.src.renderer.ray_samplers.utils import sample_pdf  # noqa: F401  (re-export, as the reference)

__all__ = ["StratifiedSampler", "RayBundle", "RaySamplerBase"]


class StratifiedSampler(RaySamplerBase):
    def sample_along_rays(self, ray_bundle: RayBundle, num_samples: Union[int, Tuple[int, int]], device,
                          weights: torch.Tensor = None):
        """-> (sample_pts (N,S,3), ray_dir (N,S,3), delta (N,S)) on `device`.

        weights given  -> hierarchical: num_samples = (coarse, fine), S = coarse + fine, sorted;
                          `weights` is floored in place (+= 1e-5) like the reference's sample_pdf
        weights absent -> stratified:   num_samples = S
        """
        hierarchical = weights is not None
        if hierarchical:
            if not isinstance(weights, torch.Tensor):
                raise ValueError(f"Expected an instance of torch.Tensor. Got {type(weights)}.")
            if not isinstance(num_samples, (tuple, list)):
                raise ValueError(
                    "Expected a tuple for parameter 'num_samples' when hierarchical sampling is used. "
                    f"Got a parameter of type {type(num_samples)}.")
            n_coarse, n_fine = num_samples
        else:
            if not isinstance(num_samples, int):
                raise ValueError(
                    "Expected an integer for parameter 'num_samples' when hierarchical sampling is unused. "
                    f"Got a parameter of type {type(num_samples)}.")
            n_coarse, n_fine = num_samples, 0

        t_bins, partition_size = self._create_t_bins(ray_bundle.t_near, ray_bundle.t_far, n_coarse, device)
        dev = t_bins.device
        origin = ray_bundle.ray_origin.to(dev)
        direction = ray_bundle.ray_dir.to(dev)
        n_rays = origin.shape[0]
        u1 = torch.rand((n_rays, n_coarse), device=dev)                      # :77 / :109
        if not hierarchical:
            return ops.sample_stratified(origin, direction, t_bins, partition_size, u1)

        u2 = torch.rand((n_rays, n_fine), device=dev)                        # utils.py:43
        u3 = torch.rand((n_rays, n_fine), device=dev)                        # utils.py:56
        w = weights.detach().to(dev)
        w_c = w if (w.is_contiguous() and w.dtype == torch.float32) else w.contiguous().float()
        out = ops.sample_hierarchical(origin, direction, t_bins, partition_size, w_c, u1, u2, u3)
        if w_c is not w:
            w.copy_(w_c)                                                     # keep the in-place side effect
        return out

    def _create_t_bins(self, t_start: float, t_end: float, num_partitions: int, device):
        """Left edges of `num_partitions` equal bins of [t_start, t_end) and the bin width."""
        edges = torch.linspace(t_start, t_end, num_partitions + 1, device=device)
        return edges[:-1], (t_end - t_start) / num_partitions
