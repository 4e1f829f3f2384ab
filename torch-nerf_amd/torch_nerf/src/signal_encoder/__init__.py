from torch_nerf.src.signal_encoder.positional_encoder import PositionalEncoder  # noqa: F401
from torch_nerf.src.signal_encoder.spherical_harmonics_encoder import SHEncoder  # noqa: F401
