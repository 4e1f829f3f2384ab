"""Inverse-CDF sampling helper with the reference's signature
(torch_nerf/src/renderer/ray_samplers/utils.py:8-58), evaluated by the hierarchical
sampling kernel (csrc/sampling.hip)."""
import torch

from torch_nerf.amd import ops


def sample_pdf(bins: torch.Tensor, partition_size: float, weights: torch.Tensor, num_sample: int) -> torch.Tensor:
    """t-values drawn from the piecewise-constant pdf `weights` over `bins` (N,S).

    Like the reference: `weights` is floored IN PLACE (+= 1e-5); the draws are
    torch.rand((N, num_sample)) for the cdf ordinates followed by a second one for the
    in-bin jitter.  All rows of `bins` must be the same bin edges (they are, in the
    reference's only call site, stratified_sampler.py:81-86).
    """
    n, sc = weights.shape
    dev = weights.device
    u2 = torch.rand((n, num_sample), device=dev)
    u3 = torch.rand((n, num_sample), device=dev)
    zeros3 = torch.zeros((n, 3), device=dev)
    u1 = torch.zeros((n, sc), device=dev)
    w = weights.detach()
    w_c = w if (w.is_contiguous() and w.dtype == torch.float32) else w.contiguous().float()
    _, _, _, idx = ops.sample_hierarchical(zeros3, zeros3, bins[0].contiguous(), partition_size, w_c, u1, u2,
                                           u3, want_idx=True)
    if w_c is not w:
        w.copy_(w_c)
    return torch.gather(bins, 1, idx) + partition_size * u3
