from torch_nerf.src.scene.primitives import *  # noqa: F401,F403
from torch_nerf.src.scene.scene import Scene  # noqa: F401
