"""Signal encoders: `PositionalEncoder` (HIP / fused) and the out-of-scope `SHEncoder` name."""
from torch_nerf.src.signal_encoder.spherical_harmonics_encoder import SHEncoder
from torch_nerf.src.signal_encoder.positional_encoder import PositionalEncoder

__all__ = ["PositionalEncoder", "SHEncoder"]
