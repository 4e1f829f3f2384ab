"""Base of the scene primitives: holds the optional encoder dictionary and checks query shapes.

Behaviour follows torch_nerf/src/scene/primitives/primitive_base.py:12-74 of the reference:
missing encoder keys only WARN at construction but are an error when assigned later.
"""
import warnings
from typing import Dict, Optional, Tuple

import torch

from torch_nerf.src.signal_encoder.signal_encoder_base import SignalEncoderBase

_ENCODER_KEYS = ("coord_enc", "dir_enc")


def _require_dict(value) -> None:
    if not isinstance(value, dict):
        raise ValueError(f"Expected a parameter of type Dict. Got {type(value)}")


class PrimitiveBase(object):
    def __init__(self, encoders: Optional[Dict[str, SignalEncoderBase]] = None):
        if encoders is not None:
            _require_dict(encoders)
            for key in _ENCODER_KEYS:
                if key not in encoders:
                    warnings.warn(f"Missing an encoder type '{key}'. Got {encoders.keys()}.")
        self._encoders = encoders

    def query_points(self, pos: torch.Tensor, view_dir: torch.Tensor) -> Tuple[int, int]:
        """Common shape check of (N,S,3) positions / directions; returns (N, S)."""
        if pos.shape != view_dir.shape:
            raise ValueError(f"Expected tensors of same shape. Got {pos.shape} and {view_dir.shape}, respectively.")
        return pos.shape[0], pos.shape[1]

    def _get_encoders(self):
        return self._encoders

    def _set_encoders(self, new_encoders) -> None:
        _require_dict(new_encoders)
        for key in _ENCODER_KEYS:
            if key not in new_encoders:
                raise ValueError(f"Missing required encoder type '{key}'. Got {new_encoders.keys()}.")
        self._encoders = new_encoders

    encoders = property(_get_encoders, _set_encoders, doc="encoder dictionary ('coord_enc', 'dir_enc') or None")
