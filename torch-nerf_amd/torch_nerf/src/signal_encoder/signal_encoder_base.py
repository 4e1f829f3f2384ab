"""Contract of a signal encoder (torch_nerf/src/signal_encoder/signal_encoder_base.py): a stateless map
applied to sample coordinates / view directions before they reach the network.  PrimitiveCube recognises the
two PositionalEncoder configurations the fused HIP kernel evaluates in registers and skips the stand-alone
`encode` call for them; any other encoder is called as is and its output handed to the generic entry point."""
import abc


class SignalEncoderBase(abc.ABC):
    def __init__(self):
        super().__init__()

    @abc.abstractmethod
    def encode(self, in_signal):
        """in_signal (N, C) -> encoded (N, out_dim)."""
