"""FusedAdam and its configuration, mirroring the reference's optimizer surface.

FusedAdam        <- src/optimizer/fused_adam.hpp:29-106, fused_adam.cu:82-219
AdamConfig       <- src/optimizer/adam.hpp:30-41
ParamGroup, PositionLRConfig, position_lr, active_sh_degree_for_step, lr_defaults
                 <- src/training/lr_schedule.hpp:23-80
"""
from __future__ import annotations

import ctypes as C
import enum
import math
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import torch

from ._lib import AdamGroup, check, lib
from .types import BackwardOutput, GaussianModel


class ParamGroup(enum.IntEnum):
    kPositions = 0
    kSHCoeffs = 1
    kOpacities = 2
    kScales = 3
    kRotations = 4


@dataclass
class PositionLRConfig:
    lr_init: float = 1.6e-4
    lr_final: float = 1.6e-6
    max_steps: int = 30000


def position_lr(step: int, config: PositionLRConfig) -> float:
    """Log-linear interpolation in float32 (lr_schedule.hpp:49-57)."""
    f32 = np.float32
    if step >= config.max_steps:
        return float(f32(config.lr_final))
    if step <= 0:
        return float(f32(config.lr_init))
    t = f32(step) / f32(config.max_steps)
    log_ratio = np.log(f32(config.lr_final) / f32(config.lr_init), dtype=f32)
    return float(f32(config.lr_init) * np.exp(t * log_ratio, dtype=f32))


def active_sh_degree_for_step(step: int, max_degree: int) -> int:
    """lr_schedule.hpp:70-72"""
    return min(step // 1000, max_degree)


class lr_defaults:
    kSHCoeffs = 2.5e-3
    kOpacity = 0.05
    kScale = 5e-3
    kRotation = 1e-3


@dataclass
class AdamConfig:
    position_lr_config: PositionLRConfig = field(default_factory=PositionLRConfig)
    lr_sh_coeffs: float = lr_defaults.kSHCoeffs
    lr_opacities: float = lr_defaults.kOpacity
    lr_scales: float = lr_defaults.kScale
    lr_rotations: float = lr_defaults.kRotation
    beta1: float = 0.9
    beta2: float = 0.999
    eps: float = 1e-15


class FusedAdam:
    """Adam over the five tensors of a GaussianModel, updated IN PLACE by one HIP launch per
    step (the reference launches one kernel per group, fused_adam.cu:155-163)."""

    kNumGroups = 5
    _names = ("positions", "sh_coeffs", "opacities", "scales", "rotations")   # ParamGroup order

    def __init__(self, model: GaussianModel, config: Optional[AdamConfig] = None):
        self.model_ = model                       # a reference, as fused_adam.hpp:94
        self.config_ = config or AdamConfig()
        self.step_count_ = 0
        params = [getattr(model, nm) for nm in self._names]
        self.m_: List[torch.Tensor] = [torch.zeros_like(p) for p in params]
        self.v_: List[torch.Tensor] = [torch.zeros_like(p) for p in params]
        self.grads_: List[Optional[torch.Tensor]] = [None] * self.kNumGroups
        c = self.config_
        self.learning_rates_ = [c.position_lr_config.lr_init, c.lr_sh_coeffs, c.lr_opacities, c.lr_scales,
                                c.lr_rotations]

    def apply_gradients(self, grads: BackwardOutput) -> None:
        """Stores references, no copy (fused_adam.cu:113-120)."""
        self.grads_ = [grads.dL_dpositions, grads.dL_dsh_coeffs, grads.dL_dopacities, grads.dL_dscales,
                       grads.dL_drotations]

    def update_lr(self, step: int) -> None:
        self.learning_rates_[0] = position_lr(step, self.config_.position_lr_config)

    def zero_grad(self) -> None:
        self.grads_ = [None] * self.kNumGroups

    def get_lr(self, group: ParamGroup) -> float:
        return self.learning_rates_[int(group)]

    def begin_fused_step(self):
        """The optimizer half of cugs_project_backward_adam (render_backward(..., fused_adam=self)): counts the
        step and returns the C struct with the moments, learning rates and bias corrections.  The update itself
        happens inside the projection backward, on the model this optimizer was built for, which must be
        contiguous float32 (the kernel writes the parameters in place)."""
        from ._lib import AdamFused
        self.step_count_ += 1
        bc1, bc2 = C.c_float(), C.c_float()
        lib.cugs_adam_bias_correction(self.config_.beta1, self.config_.beta2, self.step_count_,
                                      C.byref(bc1), C.byref(bc2))
        a = AdamFused()
        for i, nm in enumerate(self._names):
            p = getattr(self.model_, nm)
            if not (p.is_cuda and p.is_contiguous() and p.dtype == torch.float32 and self.m_[i].is_contiguous()
                    and self.v_[i].is_contiguous()):
                raise RuntimeError("FusedAdam: the fused step needs contiguous float32 CUDA parameters and moments")
            a.m[i], a.v[i] = self.m_[i].data_ptr(), self.v_[i].data_ptr()
            a.lr[i] = float(self.learning_rates_[i])
        a.beta1, a.beta2, a.eps = self.config_.beta1, self.config_.beta2, self.config_.eps
        a.bc1, a.bc2 = bc1.value, bc2.value
        self.grads_ = [None] * self.kNumGroups
        return a

    def step(self) -> None:
        self.step_count_ += 1
        bc1, bc2 = C.c_float(), C.c_float()
        lib.cugs_adam_bias_correction(self.config_.beta1, self.config_.beta2, self.step_count_,
                                      C.byref(bc1), C.byref(bc2))
        groups = (AdamGroup * self.kNumGroups)()
        keep = []                                  # keep contiguous copies alive across the launch
        dev = None
        copy_back = []
        for i, nm in enumerate(self._names):
            g = self.grads_[i]
            groups[i].grad = None
            if g is None:
                continue
            p = getattr(self.model_, nm)
            if not p.is_cuda:
                raise RuntimeError("FusedAdam: param must be on CUDA")
            if not g.is_cuda:
                raise RuntimeError("FusedAdam: grad must be on CUDA")
            if p.numel() != g.numel():
                raise RuntimeError(f"FusedAdam: param/grad size mismatch: {p.numel()} vs {g.numel()}")
            dev = p.device
            pc, gc, mc, vc = p.contiguous(), g.contiguous(), self.m_[i].contiguous(), self.v_[i].contiguous()
            keep += [pc, gc, mc, vc]
            if not p.is_contiguous():
                copy_back.append((p, pc))
            if not self.m_[i].is_contiguous():
                copy_back.append((self.m_[i], mc))
            if not self.v_[i].is_contiguous():
                copy_back.append((self.v_[i], vc))
            groups[i].param, groups[i].grad = pc.data_ptr(), gc.data_ptr()
            groups[i].m, groups[i].v = mc.data_ptr(), vc.data_ptr()
            groups[i].n = pc.numel()
            groups[i].lr = float(self.learning_rates_[i])
        if dev is None:
            return
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        check(lib.cugs_fused_adam_groups(groups, self.kNumGroups, self.config_.beta1, self.config_.beta2,
                                         self.config_.eps, bc1.value, bc2.value, stream),
              "cugs_fused_adam_groups")
        for dst, src in copy_back:                 # fused_adam.cu:216-218
            dst.copy_(src)
