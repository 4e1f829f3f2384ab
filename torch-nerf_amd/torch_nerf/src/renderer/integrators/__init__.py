"""Numerical integrators; `quadrature_integrator.QuadratureIntegrator` is the one the runners build."""
