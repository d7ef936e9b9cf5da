"""CPU oracle for SURVEY §8(f) N2: adaptive density control (statistics, clone / split / prune).
TEST INFRASTRUCTURE.

The reference implements densification purely with libtorch tensor operations
(src/optimizer/densification.cpp); the same library is installed here, so this oracle is the
reference's own operation sequence executed by libtorch on CPU in float32.  Two deliberate
differences, both at the boundary and both stated in DESIGN.md §4.9:
  * the two torch::randn_like draws of the split (densification.cpp:259-260) are replaced by a
    caller-supplied standard-normal tensor `noise` [2, N, 3] indexed by the PARENT's index, so the
    result is a function of its inputs (the reference's random stream cannot be reproduced);
  * the VRAM guards (densification.cpp:101-114, 141-170, 218-252) read the allocator's free memory -
    control plane, out of scope; the max_gaussians budget (top-k by average gradient) is kept.
Only tests/ and tools/bench_densify.py import this module.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch

K_RESET_OPACITY = -4.59511985013459          # densification.cpp:27: log(0.01 / 0.99)


def f32(x: float) -> float:
    """A C++ `float` constant or product as the double libtorch receives for a Scalar argument."""
    return float(np.float32(x))


@dataclass
class DensificationConfig:                    # densification.hpp:23-45
    densify_from: int = 500
    densify_until: int = 15000
    densify_every: int = 100
    opacity_reset_every: int = 3000
    grad_threshold: float = 0.0002
    opacity_threshold: float = 0.005
    percent_dense: float = 0.01
    max_screen_size: int = 20
    max_gaussians: int = 0


@dataclass
class DensificationStats:                     # densification.hpp:48-55
    num_cloned: int = 0
    num_split: int = 0
    num_pruned: int = 0
    num_before: int = 0
    num_after: int = 0


class Model:
    """The five parameter tensors (core/gaussian.hpp:36-40)."""
    def __init__(self, positions, sh_coeffs, opacities, rotations, scales):
        self.positions, self.sh_coeffs, self.opacities = positions, sh_coeffs, opacities
        self.rotations, self.scales = rotations, scales

    def num_gaussians(self) -> int:
        return int(self.positions.shape[0])


class DensificationController:
    def __init__(self, config: DensificationConfig, scene_extent: float):
        self.config_, self.scene_extent_ = config, float(scene_extent)
        self.grad_accum_ = self.grad_count_ = self.max_radii_2d_ = None

    # densification.cpp:42-52
    def should_densify(self, step: int) -> bool:
        c = self.config_
        return step >= c.densify_from and step <= c.densify_until and step % c.densify_every == 0

    def should_reset_opacity(self, step: int) -> bool:
        c = self.config_
        return c.opacity_reset_every > 0 and step >= c.densify_from and step % c.opacity_reset_every == 0

    # densification.cpp:344-349
    def reset_accumulators(self, n: int) -> None:
        self.grad_accum_ = torch.zeros(n, dtype=torch.float32)
        self.grad_count_ = torch.zeros(n, dtype=torch.float32)
        self.max_radii_2d_ = torch.zeros(n, dtype=torch.float32)

    # densification.cpp:59-88
    def accumulate_gradients(self, dL_dmeans_2d: torch.Tensor, radii: torch.Tensor) -> None:
        n = dL_dmeans_2d.shape[0]
        if self.grad_accum_ is None or self.grad_accum_.shape[0] != n:
            self.reset_accumulators(n)
        visible = radii.gt(0)
        grad_norms = dL_dmeans_2d.norm(2, dim=1)
        self.grad_accum_[visible] = self.grad_accum_[visible] + grad_norms[visible]
        self.grad_count_[visible] = self.grad_count_[visible] + 1
        self.max_radii_2d_ = torch.max(self.max_radii_2d_, radii.to(torch.float32))

    # densification.cpp:351-371
    def compute_clone_mask(self, model: Model) -> torch.Tensor:
        n = model.num_gaussians()
        avg_grad = self.grad_accum_[:n] / self.grad_count_[:n].clamp_min(1)
        high_grad = avg_grad.ge(f32(self.config_.grad_threshold))
        max_scale = torch.exp(model.scales).max(dim=1)[0]
        size_threshold = f32(np.float32(self.config_.percent_dense) * np.float32(self.scene_extent_))
        return high_grad & max_scale.lt(size_threshold)

    # densification.cpp:373-399
    def compute_split_mask(self, model: Model) -> torch.Tensor:
        n = model.num_gaussians()
        effective_n = min(n, self.grad_accum_.shape[0])
        avg_grad = self.grad_accum_[:effective_n] / self.grad_count_[:effective_n].clamp_min(1)
        if effective_n < n:
            avg_grad = torch.cat([avg_grad, torch.zeros(n - effective_n)])
        high_grad = avg_grad.ge(f32(self.config_.grad_threshold))
        max_scale = torch.exp(model.scales).max(dim=1)[0]
        size_threshold = f32(np.float32(self.config_.percent_dense) * np.float32(self.scene_extent_))
        return high_grad & max_scale.ge(size_threshold)

    # densification.cpp:401-442
    def compute_keep_mask(self, model: Model, step: int) -> torch.Tensor:
        n = model.num_gaussians()
        keep = torch.sigmoid(model.opacities.squeeze(1)).ge(f32(self.config_.opacity_threshold))
        if self.config_.opacity_reset_every > 0 and step > self.config_.opacity_reset_every:
            if self.config_.max_screen_size > 0 and self.max_radii_2d_ is not None:
                if self.max_radii_2d_.shape[0] >= n:
                    radii = self.max_radii_2d_[:n]
                else:
                    radii = torch.cat([self.max_radii_2d_, torch.zeros(n - self.max_radii_2d_.shape[0])])
                keep = keep & radii.le(float(self.config_.max_screen_size))
            max_scale = torch.exp(model.scales).max(dim=1)[0]
            ws_threshold = f32(np.float32(0.1) * np.float32(self.scene_extent_))
            keep = keep & max_scale.le(ws_threshold)
        return keep

    # densification.cpp:94-325 without the VRAM guards
    def densify(self, model: Model, step: int, noise: torch.Tensor) -> DensificationStats:
        cfg = self.config_
        stats = DensificationStats(num_before=model.num_gaussians())
        n0 = stats.num_before
        if self.grad_accum_ is None:
            self.reset_accumulators(n0)

        clone_mask = self.compute_clone_mask(model)
        num_to_clone = int(clone_mask.sum())
        if num_to_clone > 0:
            if cfg.max_gaussians > 0:
                budget = cfg.max_gaussians - model.num_gaussians()
                if num_to_clone > budget:
                    if budget <= 0:
                        num_to_clone = 0
                        clone_mask.zero_()
                    else:
                        avg_grad = self.grad_accum_ / self.grad_count_.clamp_min(1)
                        idx = avg_grad.masked_fill(~clone_mask, -1.0).topk(budget)[1]
                        clone_mask.zero_()
                        clone_mask[idx] = True
                        num_to_clone = budget
            if num_to_clone > 0:
                for name in ("positions", "scales", "sh_coeffs", "opacities", "rotations"):   # append_gaussians
                    t = getattr(model, name)
                    setattr(model, name, torch.cat([t, t[clone_mask]], 0))
                stats.num_cloned = num_to_clone

        split_mask = self.compute_split_mask(model)
        if stats.num_cloned > 0:
            split_mask[n0:] = False
        num_to_split = int(split_mask.sum())
        if num_to_split > 0:
            if cfg.max_gaussians > 0:
                budget = int((cfg.max_gaussians - model.num_gaussians()) / 2)       # C++ int division truncates
                if num_to_split > budget:
                    if budget <= 0:
                        num_to_split = 0
                        split_mask.zero_()
                    else:
                        avg_grad = self.grad_accum_ / self.grad_count_.clamp_min(1)
                        if avg_grad.shape[0] < model.num_gaussians():
                            avg_grad = torch.cat([avg_grad, torch.zeros(model.num_gaussians() - avg_grad.shape[0])])
                        idx = avg_grad.masked_fill(~split_mask, -1.0).topk(budget)[1]
                        split_mask.zero_()
                        split_mask[idx] = True
                        num_to_split = budget
            if num_to_split > 0:
                parents = split_mask[:n0]                       # only originals can be set
                selected_pos = model.positions[split_mask]
                new_scales = model.scales[split_mask] - float(np.log(np.float32(1.6)))     # std::log(1.6f)
                actual_scale = torch.exp(new_scales)
                pos1 = selected_pos + noise[0][parents] * actual_scale    # randn_like in the reference
                pos2 = selected_pos + noise[1][parents] * actual_scale
                sel = lambda t: t[split_mask]
                model.positions = torch.cat([model.positions, torch.cat([pos1, pos2], 0)], 0)
                model.scales = torch.cat([model.scales, torch.cat([new_scales, new_scales], 0)], 0)
                for name in ("sh_coeffs", "opacities", "rotations"):
                    t = getattr(model, name)
                    s = sel(t[:split_mask.shape[0]])
                    setattr(model, name, torch.cat([t, torch.cat([s, s], 0)], 0))
                stats.num_split = num_to_split

        keep_mask = self.compute_keep_mask(model, step)
        if num_to_split > 0:
            remove = torch.zeros(model.num_gaussians(), dtype=torch.bool)
            remove[:split_mask.shape[0]] = split_mask
            keep_mask = keep_mask & ~remove
        if stats.num_cloned > 0 or stats.num_split > 0:
            keep_mask[n0:] = True
        total_before_prune = model.num_gaussians()
        for name in ("positions", "sh_coeffs", "opacities", "rotations", "scales"):           # prune_gaussians
            setattr(model, name, getattr(model, name)[keep_mask].detach().clone().contiguous())
        stats.num_pruned = total_before_prune - model.num_gaussians()
        stats.num_after = model.num_gaussians()
        self.reset_accumulators(model.num_gaussians())
        return stats

    # densification.cpp:331-334
    def reset_opacity(self, model: Model) -> None:
        model.opacities.fill_(K_RESET_OPACITY)
