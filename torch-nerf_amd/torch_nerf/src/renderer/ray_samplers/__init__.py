"""Ray bundle + samplers.  The runners use `ray_samplers.StratifiedSampler()` and, indirectly,
`RayBundle` / `RaySamplerBase`; `sample_pdf` is re-exported for API parity."""
from torch_nerf.src.renderer.ray_samplers.sampler_base import RayBundle, RaySamplerBase
from torch_nerf.src.renderer.ray_samplers.stratified_sampler import StratifiedSampler
from torch_nerf.src.renderer.ray_samplers.utils import sample_pdf

__all__ = ["RayBundle", "RaySamplerBase", "StratifiedSampler", "sample_pdf"]
