"""Scene-primitive base (torch_nerf/src/scene/primitives/primitive_base.py:12-74)."""
from typing import Dict, Optional, Tuple
import warnings

import torch

from torch_nerf.src.signal_encoder.signal_encoder_base import SignalEncoderBase

_ENCODER_KEYS = ("coord_enc", "dir_enc")


class PrimitiveBase(object):
    def __init__(self, encoders: Optional[Dict[str, SignalEncoderBase]] = None):
        if encoders is not None:
            if not isinstance(encoders, dict):
                raise ValueError(f"Expected a parameter of type Dict. Got {type(encoders)}")
            for key in _ENCODER_KEYS:  # missing encoders only warn at construction time
                if key not in encoders:
                    warnings.warn(f"Missing an encoder type '{key}'. Got {encoders.keys()}.")
        self._encoders = encoders

    def query_points(self, pos: torch.Tensor, view_dir: torch.Tensor) -> Tuple[int, int]:
        """Shape check shared by all primitives; returns (num_ray, num_sample)."""
        if pos.shape != view_dir.shape:
            raise ValueError(f"Expected tensors of same shape. Got {pos.shape} and {view_dir.shape}, respectively.")
        num_ray, num_sample, _ = pos.shape
        return num_ray, num_sample

    @property
    def encoders(self) -> Optional[Dict[str, SignalEncoderBase]]:
        return self._encoders

    @encoders.setter
    def encoders(self, new_encoders) -> None:
        if not isinstance(new_encoders, dict):
            raise ValueError(f"Expected a parameter of type Dict. Got {type(new_encoders)}")
        for key in _ENCODER_KEYS:  # ...but are mandatory when set later
            if key not in new_encoders:
                raise ValueError(f"Missing required encoder type '{key}'. Got {new_encoders.keys()}.")
        self._encoders = new_encoders
