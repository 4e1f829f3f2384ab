from torch_nerf.src.scene.primitives.primitive_base import PrimitiveBase  # noqa: F401
from torch_nerf.src.scene.primitives.cube import PrimitiveCube  # noqa: F401
