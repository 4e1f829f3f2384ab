"""Name kept so that `from torch_nerf.src.signal_encoder import SHEncoder` resolves
(runners/runner_utils.py:21).  The spherical-harmonics encoder belongs to the Instant-NGP
configuration, which is outside the volume-rendering hot path this package accelerates
(out of scope)."""
from torch_nerf.src.signal_encoder.signal_encoder_base import SignalEncoderBase


class SHEncoder(SignalEncoderBase):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("SHEncoder (Instant-NGP path) is out of scope of the MI355X hot path")
