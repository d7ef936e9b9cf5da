"""Host-side mirror of the reference's loss surface (src/training/loss.hpp:21-52) over the fused HIP
kernels (csrc/loss.hip): l1_loss, ssim, ssim_loss, combined_loss - plus combined_loss_and_grad, which
returns the loss together with dL/dcolor, i.e. what trainer.cpp:214-217 obtains with clone + autograd + clone.
Scalars come back as 0-dim device tensors (no host sync), as in the reference."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from ._lib import check, lib
from .rasterizer import _ptr, _stream, _torch_check, _workspace


def _validate(img: torch.Tensor, name: str) -> None:
    """validate_image (loss.cpp:14-24)"""
    _torch_check(img.dim() == 3, f"{name} must be 3-dimensional [H, W, 3], got {img.dim()} dims")
    _torch_check(img.shape[2] == 3, f"{name} must have 3 channels, got {img.shape[2]}")
    _torch_check(img.dtype == torch.float32, f"{name} must be float32, got {img.dtype}")
    _torch_check(img.is_cuda, f"{name} must be on a CUDA device")


def _validate_pair(rendered: torch.Tensor, target: torch.Tensor) -> None:
    """validate_pair (loss.cpp:27-34)"""
    _validate(rendered, "rendered")
    _validate(target, "target")
    _torch_check(rendered.shape == target.shape,
                 f"rendered and target must have the same shape, got {tuple(rendered.shape)} vs {tuple(target.shape)}")


def _run(rendered, target, lambda_, window_size, want_map, want_grad):
    _validate_pair(rendered, target)
    _torch_check(window_size % 2 == 1, f"window_size must be odd, got {window_size}")
    _torch_check(window_size >= 3, f"window_size must be >= 3, got {window_size}")
    h, w = int(rendered.shape[0]), int(rendered.shape[1])
    dev = rendered.device
    r, t = rendered.contiguous(), target.contiguous()
    out = torch.empty(4, dtype=torch.float32, device=dev)
    smap = torch.empty((h, w), dtype=torch.float32, device=dev) if want_map else None
    grad = torch.empty((h, w, 3), dtype=torch.float32, device=dev) if want_grad else None
    ws = _workspace(dev, lib.cugs_loss_workspace_bytes(w, h), "loss")
    check(lib.cugs_combined_loss(w, h, _ptr(r), _ptr(t), float(lambda_), int(window_size), _ptr(ws), ws.numel(),
                                 _ptr(out), _ptr(smap), _ptr(grad), _stream(dev)), "cugs_combined_loss")
    return out, smap, grad


def l1_loss(rendered: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    return _run(rendered, target, 0.2, 11, False, False)[0][1]


def ssim(rendered: torch.Tensor, target: torch.Tensor, window_size: int = 11) -> torch.Tensor:
    """Per-pixel SSIM map [H, W] (mean across RGB)."""
    return _run(rendered, target, 0.2, window_size, True, False)[1]


def ssim_loss(rendered: torch.Tensor, target: torch.Tensor, window_size: int = 11) -> torch.Tensor:
    return _run(rendered, target, 0.2, window_size, False, False)[0][3]


def combined_loss(rendered: torch.Tensor, target: torch.Tensor, lambda_: float = 0.2) -> torch.Tensor:
    return _run(rendered, target, lambda_, 11, False, False)[0][0]


def combined_loss_and_grad(rendered: torch.Tensor, target: torch.Tensor,
                           lambda_: float = 0.2) -> Tuple[torch.Tensor, torch.Tensor]:
    """(loss, dL_dcolor [H,W,3]) in two launches: replaces clone + combined_loss + backward + clone."""
    out, _, grad = _run(rendered, target, lambda_, 11, False, True)
    return out[0], grad
