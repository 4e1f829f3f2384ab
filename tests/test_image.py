"""PNG writer: bytes decode back to the rounded image (zlib + struct only, no imaging library)."""
import struct
import zlib

import numpy as np
import torch

from torch_nerf.amd import image


def _decode(path):
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, {}
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(tag + body) & 0xFFFFFFFF
        chunks.setdefault(tag, b"")
        chunks[tag] += body
        pos += 12 + n
    w, h, depth, ctype = struct.unpack(">IIBB", chunks[b"IHDR"][:10])
    assert (depth, ctype) == (8, 2)
    raw = zlib.decompress(chunks[b"IDAT"])
    rows = [raw[y * (1 + 3 * w) + 1:(y + 1) * (1 + 3 * w)] for y in range(h)]
    return np.frombuffer(b"".join(rows), np.uint8).reshape(h, w, 3)


def test_png_roundtrip(tmp_path):
    torch.manual_seed(0)
    img = torch.rand(7, 5, 3) * 1.4 - 0.2          # includes values outside [0, 1]
    path = str(tmp_path / "x.png")
    image.save_png(path, img)
    expect = (img.clamp(0, 1) * 255 + 0.5).clamp(0, 255).to(torch.uint8).numpy()
    assert np.array_equal(_decode(path), expect)
    image.save_png(path, img.permute(2, 0, 1))      # (3, H, W) like the reference's save_image input
    assert np.array_equal(_decode(path), expect)
