"""Integrator interface (torch_nerf/src/renderer/integrators/integrator_base.py:8-26)."""


class IntegratorBase(object):
    def __init__(self, *arg, **kwargs):
        pass

    def integrate_along_rays(self, sigma, radiance, delta):
        """(sigma (N,S), radiance (N,S,3), delta (N,S)) -> (rgb (N,3), weights (N,S))."""
        raise NotImplementedError()
