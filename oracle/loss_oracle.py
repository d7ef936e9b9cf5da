"""CPU oracle for SURVEY §8(f) N1: combined_loss (L1 + SSIM) and its gradient.  TEST INFRASTRUCTURE.

The reference implements the loss purely with libtorch tensor operations (src/training/loss.cpp) and
obtains dL/dcolor from libtorch autograd (src/training/trainer.cpp:214-217).  The same library is
installed here, so this oracle is the reference's own operation sequence, line for line, executed by
libtorch on CPU in float32 - not a re-derivation.  Only tests/ and bench.py's checker legs import it.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


def gaussian_kernel(window_size: int) -> torch.Tensor:
    """get_gaussian_kernel (loss.cpp:47-83): sigma 1.5, 1-D normalised, outer product normalised again,
    shaped [3,1,ws,ws] for the grouped conv2d."""
    sigma = np.float32(1.5)
    half = window_size // 2
    k1 = torch.zeros(window_size, dtype=torch.float32)
    for i in range(window_size):
        x = np.float32(i - half)
        k1[i] = float(np.exp(np.float32(-x * x / (np.float32(2.0) * sigma * sigma))))   # std::exp(float)
    k1 = k1 / k1.sum()
    k2 = k1.unsqueeze(1) * k1.unsqueeze(0)
    k2 = k2 / k2.sum()
    return k2.unsqueeze(0).unsqueeze(0).expand(3, 1, window_size, window_size).contiguous()


def l1_loss(rendered: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """loss.cpp:88-91"""
    return (rendered - target).abs().mean()


def ssim(rendered: torch.Tensor, target: torch.Tensor, window_size: int = 11) -> torch.Tensor:
    """loss.cpp:93-129: per-pixel SSIM map [H,W] (mean over RGB), zero padding, C1 = 0.01^2, C2 = 0.03^2."""
    assert window_size % 2 == 1 and window_size >= 3
    pad = window_size // 2
    kernel = gaussian_kernel(window_size)
    x = rendered.permute(2, 0, 1).unsqueeze(0)
    y = target.permute(2, 0, 1).unsqueeze(0)
    conv = lambda t: F.conv2d(t, kernel, None, 1, pad, 1, 3)
    mu_x, mu_y = conv(x), conv(y)
    mu_x_sq, mu_y_sq, mu_xy = mu_x * mu_x, mu_y * mu_y, mu_x * mu_y
    sigma_x_sq = conv(x * x) - mu_x_sq
    sigma_y_sq = conv(y * y) - mu_y_sq
    sigma_xy = conv(x * y) - mu_xy
    c1, c2 = 0.01 * 0.01, 0.03 * 0.03
    ssim_map = ((2.0 * mu_xy + c1) * (2.0 * sigma_xy + c2)) / ((mu_x_sq + mu_y_sq + c1) * (sigma_x_sq + sigma_y_sq + c2))
    return ssim_map.squeeze(0).permute(1, 2, 0).mean(dim=2)


def ssim_loss(rendered, target, window_size: int = 11) -> torch.Tensor:
    """loss.cpp:131-134"""
    return 1.0 - ssim(rendered, target, window_size).mean()


def combined_loss(rendered, target, lambda_: float = 0.2) -> torch.Tensor:
    """loss.cpp:136-140"""
    return (1.0 - lambda_) * l1_loss(rendered, target) + lambda_ * ssim_loss(rendered, target)


def combined_loss_and_grad(rendered_np: np.ndarray, target_np: np.ndarray, lambda_: float = 0.2):
    """trainer.cpp:214-217: rendered.clone().detach().requires_grad_(); loss.backward(); grad.clone()."""
    r = torch.from_numpy(np.ascontiguousarray(rendered_np, np.float32)).clone().requires_grad_(True)
    t = torch.from_numpy(np.ascontiguousarray(target_np, np.float32))
    loss = combined_loss(r, t, lambda_)
    loss.backward()
    return float(loss.item()), r.grad.clone().numpy(), float(l1_loss(r.detach(), t)), float(ssim(r.detach(), t).mean())
