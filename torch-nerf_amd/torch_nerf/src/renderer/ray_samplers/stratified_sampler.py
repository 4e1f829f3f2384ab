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
        n_coarse, n_fine = self.check_sample_counts(num_samples, weights)
        t_bins, partition_size = self._create_t_bins(ray_bundle.t_near, ray_bundle.t_far, n_coarse, device)
        dev = t_bins.device
        origin = ray_bundle.ray_origin.to(dev)
        direction = ray_bundle.ray_dir.to(dev)
        u1, u2, u3 = self.draw_uniforms(origin.shape[0], n_coarse, n_fine, dev)
        if not hierarchical:
            return ops.sample_stratified(origin, direction, t_bins, partition_size, u1)
        w = weights.detach().to(dev)
        w_c = w if (w.is_contiguous() and w.dtype == torch.float32) else w.contiguous().float()
        out = ops.sample_hierarchical(origin, direction, t_bins, partition_size, w_c, u1, u2, u3)
        if w_c is not w:
            w.copy_(w_c)                                                     # keep the in-place side effect
        return out

    @staticmethod
    def check_sample_counts(num_samples, weights) -> Tuple[int, int]:
        """(n_coarse, n_fine) after the reference's argument checks (stratified_sampler.py:58-64, :92-96);
        n_fine = 0 for the stratified branch."""
        if weights is not None:
            if not isinstance(weights, torch.Tensor):
                raise ValueError(f"Expected an instance of torch.Tensor. Got {type(weights)}.")
            if not isinstance(num_samples, (tuple, list)):
                raise ValueError(
                    "Expected a tuple for parameter 'num_samples' when hierarchical sampling is used. "
                    f"Got a parameter of type {type(num_samples)}.")
            return int(num_samples[0]), int(num_samples[1])
        if not isinstance(num_samples, int):
            raise ValueError(
                "Expected an integer for parameter 'num_samples' when hierarchical sampling is unused. "
                f"Got a parameter of type {type(num_samples)}.")
        return num_samples, 0

    @staticmethod
    def draw_uniforms(n_rays: int, n_coarse: int, n_fine: int, dev):
        """The draws of one sample_along_rays call, with the reference's torch calls in its order and shapes:
        U1 (N,Sc) coarse jitter (:77 / :109) [-> U2 (N,Sf) cdf ordinates (utils.py:43) -> U3 (N,Sf) in-bin jitter
        (utils.py:56)]; a seed therefore reproduces the reference's stream on the same device."""
        u1 = torch.rand((n_rays, n_coarse), device=dev)
        if n_fine <= 0:
            return u1, None, None
        return u1, torch.rand((n_rays, n_fine), device=dev), torch.rand((n_rays, n_fine), device=dev)

    _T_BINS = {}      # (t_start, t_end, partitions, device) -> left bin edges; a pure function of its arguments

    def _create_t_bins(self, t_start: float, t_end: float, num_partitions: int, device):
        """Left edges of `num_partitions` equal bins of [t_start, t_end) and the bin width.  The edges are
        torch.linspace on the device, exactly as the reference builds them (stratified_sampler.py:150-162); they are
        remembered per (bounds, count, device) -- every render_scene call of a run asks for the same ones, and the launch
        is 5 us of a 1 ms bf16 render step.  The tensor is read-only for every consumer (kernels take it const)."""
        dev = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        if dev.type == "cuda" and dev.index is None:       # "cuda" means the CURRENT device: key the cache by its index,
            dev = torch.device("cuda", torch.cuda.current_device())   # or a later set_device would get the old GPU's tensor
        key = (float(t_start), float(t_end), int(num_partitions), str(dev))
        edges = self._T_BINS.get(key)
        if edges is None:
            if len(self._T_BINS) > 64:
                self._T_BINS.clear()
            with torch.inference_mode(False):              # a cached INFERENCE tensor could not be used under autograd later
                edges = torch.linspace(t_start, t_end, num_partitions + 1, device=dev)[:-1].contiguous()
            self._T_BINS[key] = edges
        return edges, (t_end - t_start) / num_partitions
