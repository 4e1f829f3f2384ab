"""Cameras, ray samplers, integrators and the volume renderer (HIP-backed)."""
