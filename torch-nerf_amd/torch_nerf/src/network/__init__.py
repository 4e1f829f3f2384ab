"""Radiance-field networks: `NeRF` (HIP kernels).  `InstantNeRF` is only a name: the runners resolve
`network.InstantNeRF` (runners/runner_utils.py:617), but Instant-NGP is an alternative model family outside the hot path
 -- constructing it says so."""
import torch.nn as nn

from torch_nerf.src.network.nerf import NeRF


class InstantNeRF(nn.Module):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("InstantNeRF is out of scope of the MI355X volume-rendering hot path")


__all__ = ["NeRF", "InstantNeRF"]
