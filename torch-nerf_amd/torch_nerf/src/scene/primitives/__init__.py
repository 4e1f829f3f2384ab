"""Scene primitives (query structures wrapping a radiance field and its encoders)."""
from torch_nerf.src.scene.primitives.cube import PrimitiveCube
from torch_nerf.src.scene.primitives.primitive_base import PrimitiveBase

__all__ = ["PrimitiveBase", "PrimitiveCube"]
