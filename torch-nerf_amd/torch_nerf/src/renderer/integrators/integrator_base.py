"""Contract every integrator of this package fulfils.

The drop-in surface keeps the reference's class name and call signature
(torch_nerf/src/renderer/integrators/integrator_base.py:8-26) because VolumeRenderer stores
"an integrator" and calls exactly one method on it.  The only implementation shipped here is
QuadratureIntegrator, whose arithmetic is the wave-scan kernel in csrc/composite.hip.
"""
import abc


class IntegratorBase(abc.ABC):
    """Turns per-sample density / radiance / interval length into per-ray colour and weights."""

    def __init__(self, *unused_args, **unused_kwargs):   # the reference's subclasses forward arbitrary arguments
        super().__init__()

    @abc.abstractmethod
    def integrate_along_rays(self, sigma, radiance, delta):
        """sigma (N,S), radiance (N,S,3), delta (N,S), all on the GPU -> (rgb (N,3), weights (N,S)).

        Must be differentiable w.r.t. sigma and radiance (the training loop back-propagates through it)."""
