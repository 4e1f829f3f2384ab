"""Thin scene wrapper (torch_nerf/src/scene/scene.py:7-45): forwards queries to its primitive."""
from typing import Sequence, Tuple

import torch

from torch_nerf.src.scene.primitives import PrimitiveBase


class Scene:
    def __init__(self, primitives: Sequence[PrimitiveBase]):
        self._primitives = primitives

    def query_points(self, pos: torch.Tensor, view_dir: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        return self._primitives.query_points(pos, view_dir)
