"""Host-side mirror of the reference's adaptive density control (optimizer/densification.hpp:23-168)
over csrc/densify.hip: DensificationConfig, DensificationStats and
DensificationController with the reference's method names and schedule.

Differences from the reference, all at the boundary:
  * densify() takes the split noise as an argument ([2, N, 3] standard normal, indexed by the parent;
    drawn from torch's device generator when omitted) instead of calling randn_like internally; with more
    than one rank the argument is mandatory (parallel.densify_replicated supplies one shared draw);
  * densify(..., optimizer=FusedAdam) carries the Adam moments through the surgery (survivors keep
    theirs, new Gaussians start at zero) instead of leaving the caller to rebuild the optimizer
    (trainer.cpp:267-304); without it the behaviour is the reference's;
  * the VRAM guards (densification.cpp:101-114, 141-170, 218-252) are not mirrored: at 288 GB of HBM the
    allocator's headroom is not what limits a model; max_gaussians is.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import _lib
from ._lib import DensifyArray, check, lib
from .rasterizer import _ptr, _stream, _torch_check
from .types import GaussianModel

K_RESET_OPACITY = -4.59511985013459           # densification.cpp:27: log(0.01 / 0.99)

_workspaces = {}


def _workspace(device: torch.device, nbytes: int) -> torch.Tensor:
    ws = _workspaces.get(device)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=device)
        _workspaces[device] = ws
    return ws


@dataclass
class DensificationConfig:
    """densification.hpp:23-45 (the two VRAM fields are not mirrored)."""
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
class DensificationStats:
    """densification.hpp:48-55"""
    num_cloned: int = 0
    num_split: int = 0
    num_pruned: int = 0
    num_before: int = 0
    num_after: int = 0
    skipped_vram: bool = False


class DensificationController:
    _PARAMS = (("positions", _lib.DENSIFY_POSITIONS), ("sh_coeffs", _lib.DENSIFY_COPY),
               ("opacities", _lib.DENSIFY_COPY), ("rotations", _lib.DENSIFY_COPY), ("scales", _lib.DENSIFY_SCALES))

    def __init__(self, config: DensificationConfig, scene_extent: float):
        self.config_ = config
        self.scene_extent_ = float(scene_extent)
        self.grad_accum_: Optional[torch.Tensor] = None
        self.grad_count_: Optional[torch.Tensor] = None
        self.max_radii_2d_: Optional[torch.Tensor] = None

    # ---- schedule (densification.cpp:42-52) ----
    def should_densify(self, step: int) -> bool:
        c = self.config_
        return step >= c.densify_from and step <= c.densify_until and step % c.densify_every == 0

    def should_reset_opacity(self, step: int) -> bool:
        c = self.config_
        return c.opacity_reset_every > 0 and step >= c.densify_from and step % c.opacity_reset_every == 0

    # ---- statistics ----
    def reset_accumulators(self, n: int, device=None) -> None:
        """densification.cpp:344-349"""
        dev = device if device is not None else (self.grad_accum_.device if self.grad_accum_ is not None
                                                 else torch.device("cuda"))
        z = lambda: torch.zeros(int(n), dtype=torch.float32, device=dev)
        self.grad_accum_, self.grad_count_, self.max_radii_2d_ = z(), z(), z()

    def accumulate_gradients(self, dL_dmeans_2d: torch.Tensor, radii: torch.Tensor) -> None:
        """densification.cpp:59-88 as one launch, no host sync."""
        _torch_check(dL_dmeans_2d.is_cuda and radii.is_cuda, "accumulate_gradients: tensors must be on CUDA")
        _torch_check(dL_dmeans_2d.dim() == 2 and dL_dmeans_2d.shape[1] == 2, "dL_dmeans_2d must be [N, 2]")
        n = int(dL_dmeans_2d.shape[0])
        _torch_check(radii.numel() == n, "radii must be [N]")
        if self.grad_accum_ is None or self.grad_accum_.shape[0] != n or self.grad_accum_.device != dL_dmeans_2d.device:
            self.reset_accumulators(n, dL_dmeans_2d.device)
        g = dL_dmeans_2d.contiguous().to(torch.float32)
        r = radii.contiguous().to(torch.int32)
        check(lib.cugs_densify_accumulate(n, _ptr(g), _ptr(r), _ptr(self.grad_accum_), _ptr(self.grad_count_),
                                          _ptr(self.max_radii_2d_), _stream(g.device)), "cugs_densify_accumulate")

    # ---- clone / split / prune ----
    def _classify(self, model: GaussianModel, step: int):
        n = model.num_gaussians()
        dev = model.positions.device
        cfg = self.config_
        flags = torch.empty(n, dtype=torch.uint8, device=dev)
        avg = torch.empty(n, dtype=torch.float32, device=dev)
        size_thr = float(np.float32(cfg.percent_dense) * np.float32(self.scene_extent_))      # :367, :394
        ws_thr = float(np.float32(0.1) * np.float32(self.scene_extent_))                      # :436
        size_pruning = cfg.opacity_reset_every > 0 and step > cfg.opacity_reset_every         # :415-416
        check(lib.cugs_densify_classify(n, _ptr(self.grad_accum_), _ptr(self.grad_count_), _ptr(self.max_radii_2d_),
                                        _ptr(model.scales.contiguous()), _ptr(model.opacities.contiguous()),
                                        float(cfg.grad_threshold), size_thr, float(cfg.opacity_threshold),
                                        int(size_pruning), float(cfg.max_screen_size), ws_thr, _ptr(flags), _ptr(avg),
                                        _stream(dev)), "cugs_densify_classify")
        return flags, avg

    def _apply_budget(self, flags: torch.Tensor, avg: torch.Tensor, n: int) -> torch.Tensor:
        """max_gaussians (densification.cpp:121-139, 184-209): keep the highest-gradient candidates.  Rare
        path; the selection itself is libtorch's topk, exactly as in the reference."""
        cap = self.config_.max_gaussians
        if cap <= 0:
            return flags
        clone = (flags & 1).bool()
        num_clone = int(clone.sum())
        if num_clone > 0:
            budget = cap - n
            if num_clone > budget:
                if budget <= 0:
                    clone = torch.zeros_like(clone)
                else:
                    idx = avg.masked_fill(~clone, -1.0).topk(budget)[1]
                    clone = torch.zeros_like(clone)
                    clone[idx] = True
                num_clone = int(clone.sum())
        split = ((flags >> 1) & 1).bool()
        num_split = int(split.sum())
        if num_split > 0:
            budget = int((cap - (n + num_clone)) / 2)               # C++ integer division truncates (:187-188)
            if num_split > budget:
                if budget <= 0:
                    split = torch.zeros_like(split)
                else:
                    idx = avg.masked_fill(~split, -1.0).topk(budget)[1]
                    split = torch.zeros_like(split)
                    split[idx] = True
        return (flags & 4) | clone.to(torch.uint8) | (split.to(torch.uint8) << 1)

    def densify(self, model: GaussianModel, step: int, noise: Optional[torch.Tensor] = None,
                optimizer=None) -> DensificationStats:
        """densification.cpp:94-325: clone, split, prune; `model` is modified in place (new tensors)."""
        n = model.num_gaussians()
        stats = DensificationStats(num_before=n, num_after=n)
        if n == 0:
            return stats
        # Replicated models (data parallelism) must split with the SAME noise on every rank: each rank's own
        # device generator would move the children differently and the replicas would drift apart silently.
        # parallel.densify_replicated() draws the noise once and broadcasts it.
        import torch.distributed as dist
        if noise is None and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            raise RuntimeError("densify: under data parallelism pass `noise` (parallel.shared_split_noise) or call "
                               "parallel.densify_replicated(): a per-rank random draw would make the replicas diverge")
        dev = model.positions.device
        _torch_check(model.positions.is_cuda, "densify: model must be on CUDA")
        if self.grad_accum_ is None or self.grad_accum_.shape[0] != n:
            self.reset_accumulators(n, dev)
        flags, avg = self._classify(model, step)
        flags = self._apply_budget(flags, avg, n).contiguous()
        ws = _workspace(dev, lib.cugs_densify_workspace_bytes(n))
        counts = (C.c_int64 * 4)()
        st = _stream(dev)
        check(lib.cugs_densify_plan(n, _ptr(flags), _ptr(ws), ws.numel(), counts, st), "cugs_densify_plan")
        kept, cloned, split, n_out = (int(c) for c in counts)
        if noise is None:
            noise = torch.randn((2, n, 3), dtype=torch.float32, device=dev) if split > 0 else \
                torch.zeros((2, n, 3), dtype=torch.float32, device=dev)
        _torch_check(tuple(noise.shape) == (2, n, 3) and noise.is_cuda, "noise must be [2, N, 3] on CUDA")
        noise = noise.contiguous().to(torch.float32)
        old_scales = model.scales.contiguous()

        srcs, dsts, descs = [], [], []
        def add(src: torch.Tensor, mode: int):
            s = src.contiguous()
            d = torch.empty((n_out,) + tuple(s.shape[1:]), dtype=torch.float32, device=dev)
            srcs.append(s); dsts.append(d)
            descs.append((s, d, int(s.numel() // max(n, 1)), mode))
            return d
        new_params = {name: add(getattr(model, name), mode) for name, mode in self._PARAMS}
        new_m = new_v = None
        if optimizer is not None:
            new_m = [add(t, _lib.DENSIFY_STATE) for t in optimizer.m_]
            new_v = [add(t, _lib.DENSIFY_STATE) for t in optimizer.v_]
        arr = (DensifyArray * len(descs))()
        for i, (s, d, rf, mode) in enumerate(descs):
            arr[i].src, arr[i].dst, arr[i].row_floats, arr[i].mode = s.data_ptr(), d.data_ptr(), rf, mode
        if n_out > 0:
            check(lib.cugs_densify_apply(n, n_out, _ptr(ws), ws.numel(), _ptr(noise), _ptr(old_scales), arr, len(descs),
                                         st), "cugs_densify_apply")
        for name, _ in self._PARAMS:
            setattr(model, name, new_params[name])
        if optimizer is not None:
            optimizer.m_, optimizer.v_ = new_m, new_v
            optimizer.grads_ = [None] * optimizer.kNumGroups

        stats.num_cloned, stats.num_split = cloned, split
        stats.num_pruned = (n + cloned + 2 * split) - n_out          # total_before_prune - num_after (:313-316)
        stats.num_after = n_out
        self.reset_accumulators(n_out, dev)                           # :321
        return stats

    def reset_opacity(self, model: GaussianModel) -> None:
        """densification.cpp:331-334"""
        model.opacities.fill_(K_RESET_OPACITY)
