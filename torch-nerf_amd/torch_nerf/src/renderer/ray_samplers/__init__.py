from torch_nerf.src.renderer.ray_samplers.sampler_base import *  # noqa: F401,F403
from torch_nerf.src.renderer.ray_samplers.stratified_sampler import *  # noqa: F401,F403
