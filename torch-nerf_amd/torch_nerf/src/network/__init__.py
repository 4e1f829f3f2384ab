"""Radiance-field networks: `NeRF` (HIP kernels) and the out-of-scope `InstantNeRF` name."""
from torch_nerf.src.network.instant_ngp import InstantNeRF
from torch_nerf.src.network.nerf import NeRF

__all__ = ["NeRF", "InstantNeRF"]
