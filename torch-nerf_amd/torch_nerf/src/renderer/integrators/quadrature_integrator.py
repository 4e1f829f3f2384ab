"""Quadrature volume-rendering integral on the GPU.

Same interface as torch_nerf/src/renderer/integrators/quadrature_integrator.py:14-67;
the ten ATen ops of the reference are one wavefront-per-ray HIP kernel
(csrc/composite.hip) with a hand-written backward, exposed as a torch.autograd.Function.
"""
import torch

from torch_nerf.amd import ops
from torch_nerf.src.renderer.integrators.integrator_base import IntegratorBase


class QuadratureIntegrator(IntegratorBase):
    def integrate_along_rays(self, sigma: torch.Tensor, radiance: torch.Tensor, delta: torch.Tensor):
        """w_i = T_i (1 - exp(-sigma_i delta_i)),  rgb = sum_i w_i c_i.  Differentiable in sigma, radiance."""
        return ops.CompositeFunction.apply(sigma, radiance, delta)
