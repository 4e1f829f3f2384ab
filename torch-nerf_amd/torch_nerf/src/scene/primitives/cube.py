"""Cubic scene primitive: (positions, view directions) -> (density, radiance).

Interface of torch_nerf/src/scene/primitives/cube.py:12-81.  When the primitive wraps the
HIP-backed NeRF with two positional encoders, the reshape -> encode -> encode -> MLP -> reshape chain of the
reference (:59-76) hands RAW points to the network kernel: ONE fused kernel with the encodings built in registers
(feat_dim 256, pos_dim <= 64, view_dir_dim <= 32), or the encodings written straight into the layered kernel's input
planes (any other size).  Any other combination (SHEncoder, foreign encoders) goes through the encoders' and the
network's own (HIP-backed) entry points, step by step.
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

    def raw_net(self):
        """ops.Net when query_points can hand RAW points to a kernel that encodes them itself (HIP NeRF of either family
        behind two PositionalEncoders of matching widths), else None."""
        net, enc = self._radiance_field, self._encoders
        if not hasattr(net, "raw_net") or not enc:
            return None
        pe, de = enc.get("coord_enc"), enc.get("dir_enc")
        return None if pe is None or de is None else net.raw_net(pe, de)

    @property
    def fused_query(self) -> bool:
        """True when query_points hands RAW points to ONE network kernel AND keeps nothing per sample in HBM -- the
        inference path (no gradients wanted): the renderer then ignores `num_ray_batch`, which exists to bound exactly
        that.  A call that records for a backward pass allocates activation planes for every row it is given (10 - 20 KB
        per sample), so there the caller's batching is honoured like in the reference (volume_renderer.py:229-254)."""
        if self.raw_net() is None:
            return False
        net = self._radiance_field
        recording = torch.is_grad_enabled() and any(p.requires_grad for p in net.parameters())
        return not recording

    def query_points(self, pos: torch.Tensor, view_dir: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """pos, view_dir (N,S,3) -> sigma (N,S), radiance (N,S,3)."""
        num_ray, num_sample = super().query_points(pos, view_dir)
        flat = num_ray * num_sample
        raw = self.raw_net()
        if raw is not None:   # raw points in, no encoding tensor in HBM (differentiable w.r.t. pos / view_dir as well)
            sigma, radiance = self._radiance_field.forward_fused(pos.reshape(flat, -1), view_dir.reshape(flat, -1), raw)
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
