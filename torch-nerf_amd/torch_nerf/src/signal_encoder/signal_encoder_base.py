"""Signal-encoder interface (torch_nerf/src/signal_encoder/signal_encoder_base.py)."""


class SignalEncoderBase:
    def __init__(self):
        pass

    def encode(self, in_signal):
        """(N, C) -> (N, out_dim)."""
        raise NotImplementedError()
