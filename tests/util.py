"""Helpers shared by the parity tests."""
from __future__ import annotations

import importlib.util
import os
import sys

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_parity():
    """oracle/parity.py (test infrastructure): the two yardsticks every report prints side by side."""
    name = "cugs_parity"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(_ROOT, "oracle", "parity.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


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


GRAD_NAMES = ("dL_dpositions", "dL_drotations", "dL_dscales", "dL_dopacities", "dL_dsh_coeffs")


def blend_stage_report(pkg, orc, dev, out, ref, g, bg, n, w, h, rows=None, threads=1):
    """The four 2-D accumulators of the blend backward (GPU, reference layout) against the oracle's fp64 sums with the
    magnitudes of their terms: the element-wise SURVEY 8d figure AND, for every element over the bar, whether
    |diff| is within the fp32 bound of its own terms (oracle/parity.py: blend_accumulator_report)."""
    import torch
    par = load_parity()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    rb = pkg.rasterize_backward(t(g), out.means_2d, out.cov_2d_inv, out.rgb, out.opacities_act, out.tile_ranges,
                                out.gaussian_indices, out.final_T, out.n_contrib, w, h, bg, n, packed=out.packed)
    want = orc.rasterize_backward_magnitudes(w, h, bg, ref["tile_ranges"], ref["values"], ref["means_2d"],
                                             ref["cov_2d_inv"], ref["rgb"], ref["opacities_act"], g, ref["final_T"],
                                             ref["n_contrib"], n, rows=rows, threads=threads)
    got = {k: np_(getattr(rb, k)) for k in par.ACCUMULATORS}
    entries = np.bincount(ref["values"], minlength=n)
    return par.blend_accumulator_report(got, want, want["mag"], ref["cov_2d_inv"], entries)
