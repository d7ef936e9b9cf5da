"""Training-view cache: decoded 8-bit images live on the device, the float target of an
iteration is one launch (csrc/views.hip) instead of decode + CPU resize + 12 B/pixel upload
(training/trainer.cpp:186-198, data/image_io.cpp).  Mirrors the reference's image helpers where they are
format-independent: load_image for binary PPM (the format its own tests use; other formats go through
stb_image in the reference and are the caller's business here), resize_image, load_image_resized,
image_to_tensor."""
from __future__ import annotations

import os
from typing import List, Tuple

import numpy as np
import torch

from ._lib import check, lib
from .rasterizer import _ptr, _stream, _torch_check


def load_image_u8(path) -> np.ndarray:
    """Binary PPM (P6, maxval 255) -> uint8 [H, W, 3].  RuntimeError('Failed to load image: ...') otherwise
    (image_io.cpp:20-23)."""
    try:
        buf = open(os.fspath(path), "rb").read()
    except OSError:
        raise RuntimeError("Failed to load image: %s (can't fopen)" % path)
    tok, pos = [], 0
    while len(tok) < 4 and pos < len(buf):
        while pos < len(buf) and buf[pos:pos + 1].isspace():
            pos += 1
        if buf[pos:pos + 1] == b"#":
            pos = buf.index(b"\n", pos) + 1
            continue
        end = pos
        while end < len(buf) and not buf[end:end + 1].isspace():
            end += 1
        tok.append(buf[pos:end])
        pos = end
    if len(tok) < 4 or tok[0] != b"P6" or tok[3] != b"255":
        raise RuntimeError("Failed to load image: %s (unknown image type)" % path)
    w, h = int(tok[1]), int(tok[2])
    pos += 1                                                         # the single whitespace after maxval
    if len(buf) - pos < w * h * 3:
        raise RuntimeError("Failed to load image: %s (truncated)" % path)
    return np.frombuffer(buf, np.uint8, w * h * 3, pos).reshape(h, w, 3).copy()


def image_to_float(src_u8: torch.Tensor, width: int, height: int) -> torch.Tensor:
    """uint8 [h, w, 3] on the device -> float32 [height, width, 3]: x 1/255 and the reference's resize_image."""
    _torch_check(src_u8.is_cuda and src_u8.dtype == torch.uint8 and src_u8.dim() == 3 and src_u8.shape[2] == 3,
                 "image must be a uint8 [H, W, 3] CUDA tensor")
    if width <= 0 or height <= 0:
        raise RuntimeError("Invalid target dimensions for resize")    # image_io.cpp:48-50
    src = src_u8.contiguous()
    dst = torch.empty((height, width, 3), dtype=torch.float32, device=src.device)
    check(lib.cugs_image_to_float(int(src.shape[1]), int(src.shape[0]), _ptr(src), int(width), int(height), _ptr(dst),
                                  _stream(src.device)), "cugs_image_to_float")
    return dst


def load_image_resized(path, resolution_scale: int, device) -> torch.Tensor:
    """image_io.cpp:102-118 on the device: float [H/s, W/s, 3]."""
    img = torch.from_numpy(load_image_u8(path)).to(device)
    if resolution_scale <= 1:
        return image_to_float(img, img.shape[1], img.shape[0])
    return image_to_float(img, max(1, img.shape[1] // resolution_scale), max(1, img.shape[0] // resolution_scale))


class ViewCache:
    """All training images of a dataset, decoded once, 3 B/pixel on the device."""

    def __init__(self, device):
        self.device = torch.device(device)
        self._views: List[torch.Tensor] = []

    def add(self, image_u8) -> int:
        t = torch.as_tensor(image_u8)
        _torch_check(t.dtype == torch.uint8 and t.dim() == 3 and t.shape[2] == 3, "view must be uint8 [H, W, 3]")
        self._views.append(t.to(self.device).contiguous())
        return len(self._views) - 1

    def add_file(self, path) -> int:
        return self.add(load_image_u8(path))

    def __len__(self) -> int:
        return len(self._views)

    def size(self, index: int) -> Tuple[int, int]:
        v = self._views[index]
        return int(v.shape[1]), int(v.shape[0])

    def bytes(self) -> int:
        return sum(int(v.numel()) for v in self._views)

    def target(self, index: int, width: int, height: int) -> torch.Tensor:
        """The tensor trainer.cpp:186-198 builds: the view at the camera's resolution, float [H, W, 3]."""
        return image_to_float(self._views[index], width, height)
