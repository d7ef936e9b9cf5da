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


class StreamedViewCache:
    """The same targets for view sets that do NOT fit in HBM (SURVEY §8f N4 "decode once, cache, prefetch"): the
    decoded 8-bit views stay in PINNED host memory and travel to the device one iteration ahead of their use, on a
    side stream, into a small ring of device slots - 3 B/pixel over PCIe hidden under the previous iteration's
    render instead of the reference's decode + CPU resize + 12 B/pixel upload in the iteration's own critical
    path (training/trainer.cpp:186-198).  target() gives bit-identical tensors to ViewCache.target().

        cache.prefetch(next_index)                 # right after sampling the next view, before this render
        tgt = cache.target(index, W, H)            # waits (on the device) only for that view's upload

    A view that was not prefetched is uploaded on demand (correct, just not overlapped).  Slot reuse is ordered
    by events both ways: an upload waits for the last kernel that read the slot, a reader waits for the upload."""

    def __init__(self, device, slots: int = 2):
        _torch_check(slots >= 2, "at least two slots (one in use, one in flight)")
        self.device = torch.device(device)
        self._host: List[torch.Tensor] = []                    # pinned uint8 [H, W, 3]
        self._slot_buf: List[torch.Tensor] = [torch.empty(0, dtype=torch.uint8, device=self.device) for _ in range(slots)]
        self._slot_view = [-1] * slots                         # which view a slot holds (or is receiving)
        self._slot_ready: List[torch.cuda.Event] = [None] * slots      # upload finished (recorded on the copy stream)
        self._slot_free: List[torch.cuda.Event] = [None] * slots       # last reader queued (recorded on its stream)
        self._next = 0
        self._copy_stream = torch.cuda.Stream(device=self.device)
        self.uploads = 0                                       # statistics: total uploads / of which on demand
        self.misses = 0

    def add(self, image_u8) -> int:
        t = torch.as_tensor(image_u8)
        _torch_check(t.dtype == torch.uint8 and t.dim() == 3 and t.shape[2] == 3, "view must be uint8 [H, W, 3]")
        self._host.append(t.contiguous().cpu().pin_memory())
        return len(self._host) - 1

    def add_file(self, path) -> int:
        return self.add(load_image_u8(path))

    def __len__(self) -> int:
        return len(self._host)

    def size(self, index: int) -> Tuple[int, int]:
        v = self._host[index]
        return int(v.shape[1]), int(v.shape[0])

    def host_bytes(self) -> int:
        return sum(int(v.numel()) for v in self._host)

    def _slot_of(self, index: int) -> int:
        for s, v in enumerate(self._slot_view):
            if v == index:
                return s
        return -1

    def prefetch(self, index: int) -> None:
        """Queue the upload of view `index` into the next ring slot (no-op if it is already resident or in flight)."""
        if self._slot_of(index) >= 0:
            return
        s = self._next
        self._next = (self._next + 1) % len(self._slot_buf)
        src = self._host[index]
        cur = torch.cuda.current_stream(self.device)
        if self._slot_buf[s].numel() < src.numel():
            # Growing a slot involves the caching allocator on BOTH sides.  The old buffer may still be the target of
            # an upload that nobody has read (a prefetched view that was never asked for): the current stream - whose
            # pool the block returns to, and which may hand it to its next tenant at once - waits for that upload first.
            # The new block may be one whose previous tenant is still being read by kernels queued on the current
            # stream: the copy stream waits for everything queued there so far, and the block is marked as used by
            # the copy stream so that a later free does not recycle it under an upload in flight.
            if self._slot_ready[s] is not None:
                cur.wait_event(self._slot_ready[s])
            self._slot_buf[s] = torch.empty(int(src.numel()), dtype=torch.uint8, device=self.device)
            self._slot_buf[s].record_stream(self._copy_stream)
            self._copy_stream.wait_stream(cur)
        with torch.cuda.stream(self._copy_stream):
            if self._slot_free[s] is not None:
                self._copy_stream.wait_event(self._slot_free[s])        # the slot's last reader has finished
            self._slot_buf[s][:src.numel()].copy_(src.reshape(-1), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._copy_stream)
        self._slot_view[s], self._slot_ready[s] = index, ev
        self.uploads += 1

    def target(self, index: int, width: int, height: int) -> torch.Tensor:
        s = self._slot_of(index)
        if s < 0:                                               # not prefetched: upload now, same path
            self.misses += 1
            self.prefetch(index)
            s = self._slot_of(index)
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(self._slot_ready[s])
        src = self._host[index]
        view = self._slot_buf[s][:src.numel()].view(src.shape)
        out = image_to_float(view, width, height)
        free = torch.cuda.Event()
        free.record(cur)
        self._slot_free[s] = free
        return out
