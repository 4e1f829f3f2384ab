"""Scene representations queried by the renderer: `PrimitiveBase`, `PrimitiveCube`, `Scene`."""
from torch_nerf.src.scene.primitives.primitive_base import PrimitiveBase
from torch_nerf.src.scene.primitives.cube import PrimitiveCube
from torch_nerf.src.scene.scene import Scene

__all__ = ["PrimitiveBase", "PrimitiveCube", "Scene"]
