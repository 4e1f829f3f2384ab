"""sin/cos positional encoding.

Interface of torch_nerf/src/signal_encoder/positional_encoder.py:12-114:
encode(x) = [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)] (no pi, every
block spans all C channels).  Stand-alone calls run csrc/posenc.hip; inside the renderer the
encoding is fused into the MLP kernel and this class only describes it (in_dim, level).
"""
import torch

from torch_nerf.amd import ops
from torch_nerf.src.signal_encoder.signal_encoder_base import SignalEncoderBase


class PositionalEncoder(SignalEncoderBase):
    def __init__(self, in_dim: int, embed_level: int, include_input: bool):
        super().__init__()
        self._in_dim = in_dim
        self._embed_level = embed_level
        self._include_input = include_input
        self._out_dim = 2 * embed_level * in_dim + (in_dim if include_input else 0)

    def encode(self, in_signal: torch.Tensor) -> torch.Tensor:
        if torch.is_grad_enabled() and isinstance(in_signal, torch.Tensor) and in_signal.requires_grad:
            if in_signal.ndim != 2:
                raise ValueError(f"Expected a 2D tensor (N, C). Got {in_signal.ndim}-D.")
            return ops.PosencFunction.apply(in_signal, self._embed_level, self._include_input)
        return ops.posenc(in_signal, self._embed_level, self._include_input)

    in_dim = property(lambda self: self._in_dim)
    out_dim = property(lambda self: self._out_dim)
    embed_level = property(lambda self: self._embed_level)
    include_input = property(lambda self: self._include_input)
