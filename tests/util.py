"""Helpers shared by the parity tests."""
from __future__ import annotations

import numpy as np


def max_rel_err(got, ref, floor_frac=1e-6):
    """max |g - r| / max(|r|, tau) with tau = floor_frac * max|r| (per tensor)."""
    got = np.asarray(got, np.float64).reshape(-1)
    ref = np.asarray(ref, np.float64).reshape(-1)
    if ref.size == 0:
        return 0.0
    tau = floor_frac * max(float(np.max(np.abs(ref))), 1e-300)
    return float(np.max(np.abs(got - ref) / np.maximum(np.abs(ref), tau)))


def max_err_over_max(got, ref):
    """max |g - r| / max |r|: error relative to the tensor's scale."""
    got = np.asarray(got, np.float64).reshape(-1)
    ref = np.asarray(ref, np.float64).reshape(-1)
    if ref.size == 0:
        return 0.0
    return float(np.max(np.abs(got - ref)) / max(float(np.max(np.abs(ref))), 1e-300))


def cam_args(cam):
    K = cam.intrinsics
    return dict(rotation=cam.rotation, translation=cam.translation, fx=K.fx, fy=K.fy, cx=K.cx, cy=K.cy)


def oracle_forward(orc, arrays, cam, bg=(0.0, 0.0, 0.0), degree=3, scale_mod=1.0):
    K = cam.intrinsics
    return orc.render(arrays, cam.rotation, cam.translation, K.fx, K.fy, K.cx, K.cy, cam.width, cam.height,
                      bg=bg, active_degree=degree, scale_mod=scale_mod)


def oracle_backward(orc, g, fwd, arrays, cam, bg=(0.0, 0.0, 0.0), scale_mod=1.0):
    K = cam.intrinsics
    return orc.render_backward(g, fwd, arrays, K.fx, K.fy, K.cx, K.cy, cam.width, cam.height, bg=bg,
                               scale_mod=scale_mod)


def np_(t):
    return t.detach().cpu().numpy()
