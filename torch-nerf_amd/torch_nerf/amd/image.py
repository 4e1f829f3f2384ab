"""Minimal PNG writer (zlib + struct only) for rendered frames.

The reference saves frames with torchvision.utils.save_image after clamping to [0, 1]
(runners/render.py:101-105, runner_utils.py:911-918); torchvision is not a dependency of this
package, so the writer is 20 lines of standard library.  8-bit RGB, no alpha, no interlace.
"""
import struct
import zlib

import numpy as np
import torch


def _chunk(tag: bytes, data: bytes) -> bytes:
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def to_uint8(rgb: torch.Tensor) -> np.ndarray:
    """(H, W, 3) or (3, H, W) float tensor -> (H, W, 3) uint8, like save_image: clamp, *255, +0.5, truncate."""
    if rgb.ndim != 3:
        raise ValueError(f"expected a 3-D image tensor, got {tuple(rgb.shape)}")
    if rgb.shape[0] == 3 and rgb.shape[-1] != 3:
        rgb = rgb.permute(1, 2, 0)
    x = rgb.detach().float().clamp(0.0, 1.0).mul(255.0).add_(0.5).clamp_(0, 255)
    return x.to("cpu", torch.uint8).numpy()


def save_png(path: str, rgb: torch.Tensor) -> None:
    img = to_uint8(rgb)
    h, w, _ = img.shape
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))  # filter type 0 on every scanline
    png = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
    png += _chunk(b"IDAT", zlib.compress(raw, 6)) + _chunk(b"IEND", b"")
    with open(path, "wb") as f:
        f.write(png)
