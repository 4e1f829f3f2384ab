"""Signal encoders: `PositionalEncoder` (csrc/posenc.hip, or evaluated in registers by the fused kernels) and `SHEncoder` (csrc/shenc.hip)."""
from torch_nerf.src.signal_encoder.spherical_harmonics_encoder import SHEncoder
from torch_nerf.src.signal_encoder.positional_encoder import PositionalEncoder

__all__ = ["PositionalEncoder", "SHEncoder"]
