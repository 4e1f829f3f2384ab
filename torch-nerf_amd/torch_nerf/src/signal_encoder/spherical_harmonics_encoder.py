"""Spherical-harmonics encoder.

Interface of torch_nerf/src/signal_encoder/spherical_harmonics_encoder.py:11-139: SHEncoder(in_dim, degree),
encode((N,3)) -> (N, degree**2), properties in_dim / degree / out_dim.  The runners build it for BOTH network inputs
under `signal_encoder: sh` (runners/runner_utils.py:595-604; configs/signal_encoder/sh.yaml: degree 4 -> NeRF(16, 16)).
The polynomial runs in csrc/shenc.hip, products in the reference's order (bit-identical fp32 values); gradients
w.r.t. the input flow through ops.ShencFunction.  The 16-wide outputs feed NeRF.forward's pre-encoded entry of the
fused kernel family.
"""
import torch

from torch_nerf.amd import ops
from torch_nerf.src.signal_encoder.signal_encoder_base import SignalEncoderBase


class SHEncoder(SignalEncoderBase):
    def __init__(self, in_dim: int, degree: int):
        super().__init__()
        self._in_dim = in_dim
        self._degree = degree
        self._out_dim = degree ** 2

    def encode(self, in_signal: torch.Tensor) -> torch.Tensor:
        """(N, 3) -> (N, degree**2).  Degrees beyond 5 have no terms in the reference either (its output keeps
        uninitialised columns there); they are refused here."""
        if torch.is_grad_enabled() and isinstance(in_signal, torch.Tensor) and in_signal.requires_grad:
            return ops.ShencFunction.apply(in_signal, self._degree)
        return ops.shenc(in_signal, self._degree)

    in_dim = property(lambda self: self._in_dim)
    degree = property(lambda self: self._degree)
    out_dim = property(lambda self: self._out_dim)
