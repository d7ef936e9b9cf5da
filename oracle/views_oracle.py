"""CPU oracle for SURVEY §8(f) N4: float conversion and resize_image.  TEST INFRASTRUCTURE.
A numpy float32 restatement of data/image_io.cpp:35-39 (uint8 -> float, x 1/255) and :47-100 (pixel-centre
bilinear resize, clamped edges) with the reference's operation order, checked against its own tests
(tests/test_dataset.cpp:300-355).  Only tests/ and tools/ import this module."""
from __future__ import annotations

import numpy as np

F = np.float32


def convert_u8(raw: np.ndarray) -> np.ndarray:
    """image_io.cpp:35-39"""
    return raw.astype(F) * F(1.0 / 255.0)


def resize_image(src: np.ndarray, target_width: int, target_height: int) -> np.ndarray:
    """image_io.cpp:47-100 on a float32 [h, w, ch] array."""
    if target_width <= 0 or target_height <= 0:
        raise RuntimeError("Invalid target dimensions for resize")
    sh, sw, _ = src.shape
    x_scale, y_scale = F(sw) / F(target_width), F(sh) / F(target_height)
    ys, xs = np.arange(target_height, dtype=F), np.arange(target_width, dtype=F)
    src_y = (ys + F(0.5)) * y_scale - F(0.5)
    y0 = np.maximum(0, np.floor(src_y).astype(np.int64))
    y1 = np.minimum(sh - 1, y0 + 1)
    fy = (src_y - y0.astype(F)).astype(F)
    src_x = (xs + F(0.5)) * x_scale - F(0.5)
    x0 = np.maximum(0, np.floor(src_x).astype(np.int64))
    x1 = np.minimum(sw - 1, x0 + 1)
    fx = (src_x - x0.astype(F)).astype(F)
    v00, v10 = src[y0][:, x0], src[y0][:, x1]
    v01, v11 = src[y1][:, x0], src[y1][:, x1]
    fxb, fyb = fx[None, :, None], fy[:, None, None]
    top = (v00 + (v10 - v00) * fxb).astype(F)
    bot = (v01 + (v11 - v01) * fxb).astype(F)
    return (top + (bot - top) * fyb).astype(F)


def target(raw_u8: np.ndarray, width: int, height: int) -> np.ndarray:
    """trainer.cpp:186-198: convert, resize only if the sizes differ."""
    img = convert_u8(raw_u8)
    if img.shape[1] != width or img.shape[0] != height:
        img = resize_image(img, width, height)
    return img
