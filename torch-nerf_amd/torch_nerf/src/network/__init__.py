from torch_nerf.src.network.nerf import *  # noqa: F401,F403
from torch_nerf.src.network.instant_ngp import *  # noqa: F401,F403
