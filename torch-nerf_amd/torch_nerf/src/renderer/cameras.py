"""Pinhole camera container (host side, tiny).

Mirrors torch_nerf/src/renderer/cameras.py:10-153 of the reference: same constructor,
same properties, same ValueError conditions.  The kernels read four numbers of the
intrinsic matrix (fx, fy, cx, cy) and the [R|t] block of the extrinsic.
"""
from typing import Dict, Tuple, Union

import torch


def _intrinsic_matrix(fx: float, fy: float, width: float, height: float) -> torch.Tensor:
    # rows 2 and 3 are placeholders, exactly as in the reference (cameras.py:109-117)
    K = torch.zeros((4, 4), dtype=torch.float32)
    K[0, 0], K[0, 2] = fx, width / 2.0
    K[1, 1], K[1, 2] = fy, height / 2.0
    K[3, 2] = -1.0
    return K


class PerspectiveCamera(object):
    """Intrinsic (4x4 tensor or {'f_x','f_y','img_width','img_height'}), extrinsic, t_near, t_far."""

    def __init__(self, intrinsic: Union[torch.Tensor, Dict[str, float]], extrinsic: torch.Tensor,
                 t_near: float, t_far: float):
        if not isinstance(intrinsic, (torch.Tensor, dict)):
            raise ValueError(
                f"Expected torch.Tensor of Python Dict as a camera intrinsic. Got {type(intrinsic)}.")
        self._extrinsic = extrinsic
        self._t_near = t_near
        self._t_far = t_far
        if isinstance(intrinsic, dict):
            fx, fy = float(intrinsic["f_x"]), float(intrinsic["f_y"])
            w, h = float(intrinsic["img_width"]), float(intrinsic["img_height"])
            self._intrinsic = _intrinsic_matrix(fx, fy, w, h)
            self._focal_x, self._focal_y = fx, fy
            self._img_width, self._img_height = int(w), int(h)
        else:
            if intrinsic.shape != torch.Size((4, 4)):
                raise ValueError(f"Expected a tensor of shape (4, 4). Got {intrinsic.shape}.")
            self._intrinsic = intrinsic
            self._focal_x, self._focal_y = float(intrinsic[0, 0]), float(intrinsic[1, 1])
            self._img_width = int(2 * intrinsic[0, 2])
            self._img_height = int(2 * intrinsic[1, 2])

    # -- read access
    intrinsic = property(lambda self: self._intrinsic)
    extrinsic = property(lambda self: self._extrinsic)
    t_near = property(lambda self: self._t_near)
    t_far = property(lambda self: self._t_far)
    img_width = property(lambda self: self._img_width)
    img_height = property(lambda self: self._img_height)

    @property
    def focal_lengths(self) -> Tuple[float, float]:
        return (self._focal_x, self._focal_y)

    # -- write access (same checks as the reference; its t_near/t_far setters call
    #    isinstance(x, int, float) and therefore raise TypeError -- preserved)
    @intrinsic.setter
    def intrinsic(self, value: torch.Tensor) -> None:
        self._intrinsic = self._checked_matrix(value)

    @extrinsic.setter
    def extrinsic(self, value: torch.Tensor) -> None:
        self._extrinsic = self._checked_matrix(value)

    @t_near.setter
    def t_near(self, value: float) -> None:
        if not isinstance(value, int, float):  # noqa: same (faulty) call as cameras.py:182
            raise ValueError(f"Expected variable of numeric type. Got {type(value)}.")
        self._t_near = float(value)

    @t_far.setter
    def t_far(self, value: float) -> None:
        if not isinstance(value, int, float):  # noqa: same (faulty) call as cameras.py:191
            raise ValueError(f"Expected variable of numeric type. Got {type(value)}.")
        self._t_far = float(value)

    @staticmethod
    def _checked_matrix(value):
        if not isinstance(value, torch.Tensor):
            raise ValueError(f"Expected variable of type torch.Tensor. Got {type(value)}.")
        if value.shape != torch.Size((4, 4)):
            raise ValueError(f"Expected tensor of shape (4, 4). Got {value.shape}.")
        return value
