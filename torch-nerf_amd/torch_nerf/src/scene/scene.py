"""`Scene`: the object the reference's runners may hand to the renderer instead of a bare primitive
(torch_nerf/src/scene/scene.py:7-45).  It owns a primitive -- here always a PrimitiveCube whose
queries run the fused HIP MLP -- and relays point queries to it unchanged."""
from typing import Tuple

import torch

from torch_nerf.src.scene.primitives import PrimitiveBase

__all__ = ["Scene"]


class Scene:
    def __init__(self, primitives: PrimitiveBase):
        # the reference annotates a Sequence but calls query_points on the object itself (scene.py:45);
        # the behaviour, not the annotation, is what is reproduced
        self._primitives = primitives

    @property
    def primitives(self) -> PrimitiveBase:
        return self._primitives

    def query_points(self, pos: torch.Tensor, view_dir: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """pos, view_dir (N,S,3) -> (sigma (N,S), radiance (N,S,3)); see PrimitiveCube.query_points."""
        sigma, radiance = self._primitives.query_points(pos, view_dir)
        return sigma, radiance
