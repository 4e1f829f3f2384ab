"""Cubic scene primitive: (positions, view directions) -> (density, radiance).

Interface of torch_nerf/src/scene/primitives/cube.py:12-81.  When the primitive wraps the
HIP-backed NeRF with the two standard positional encoders, the reshape -> encode -> encode ->
MLP -> reshape chain of the reference (:59-76) is ONE fused kernel: the encodings are built
in registers and never written to memory.  Any other combination goes through the
encoders' and the network's own (HIP-backed) entry points, step by step.
"""
from typing import Dict, Optional, Tuple

import torch

from torch_nerf.src.scene.primitives.primitive_base import PrimitiveBase
from torch_nerf.src.signal_encoder.signal_encoder_base import SignalEncoderBase


class PrimitiveCube(PrimitiveBase):
    def __init__(self, radiance_field: torch.nn.Module, encoders: Optional[Dict[str, SignalEncoderBase]] = None):
        super().__init__(encoders=encoders)
        if not isinstance(radiance_field, torch.nn.Module):
            raise ValueError(f"Expected a parameter of type torch.nn.Module. Got {type(radiance_field)}.")
        self._radiance_field = radiance_field

    def fused_net(self):
        """ops.Net of the network behind its two encoders when query_points can run as the single fused encode+MLP
        kernel (HIP NeRF of the fused family + two PositionalEncoders of matching widths), else None."""
        net, enc = self._radiance_field, self._encoders
        if not hasattr(net, "fused_net") or not enc:
            return None
        pe, de = enc.get("coord_enc"), enc.get("dir_enc")
        return None if pe is None or de is None else net.fused_net(pe, de)

    @property
    def fused_query(self) -> bool:
        """True when query_points runs as the single fused encode+MLP kernel."""
        return self.fused_net() is not None

    def query_points(self, pos: torch.Tensor, view_dir: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """pos, view_dir (N,S,3) -> sigma (N,S), radiance (N,S,3)."""
        num_ray, num_sample = super().query_points(pos, view_dir)
        flat = num_ray * num_sample
        fused = self.fused_net()
        if fused is not None:   # (differentiable w.r.t. pos / view_dir as well)
            sigma, radiance = self._radiance_field.forward_fused(pos.reshape(flat, -1), view_dir.reshape(flat, -1), fused)
        else:
            enc = self._encoders or {}
            p, d = pos.reshape(flat, -1), view_dir.reshape(flat, -1)
            if "coord_enc" in enc:
                p = enc["coord_enc"].encode(p)
            if "dir_enc" in enc:
                d = enc["dir_enc"].encode(d)
            sigma, radiance = self._radiance_field(p, d)
        return sigma.reshape(num_ray, num_sample), radiance.reshape(num_ray, num_sample, -1)

    @property
    def radiance_field(self) -> torch.nn.Module:
        return self._radiance_field
