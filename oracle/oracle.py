"""ctypes/numpy front end of oracle/liboracle.so (and oracle/_ref/libref_sh.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  Nothing in the product package imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")
REF_SH_PATH = os.path.join(_HERE, "_ref", "libref_sh.so")

_lib = C.CDLL(LIB_PATH)
_lib.orc_count_pairs.restype = C.c_int64
_lib.orc_sort.restype = C.c_int
_lib.orc_position_lr.restype = C.c_float
_lib.orc_expf.restype = C.c_float
_lib.orc_expf.argtypes = [C.c_float]
_lib.orc_project_sh_forward_mt.restype = C.c_int
_lib.orc_rasterize_backward_rows_mt.restype = C.c_int

TILE = 16


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _i(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int32)


def view_matrix(rotation: np.ndarray, translation: np.ndarray) -> np.ndarray:
    """Row-major 4x4 world-to-camera as float32[16] (projection.cu:228-233)."""
    m = np.eye(4, dtype=np.float32)
    m[:3, :3] = np.asarray(rotation, np.float32)
    m[:3, 3] = np.asarray(translation, np.float32)
    return m.reshape(-1).copy()


def project_forward(positions, rotations, scales, opacities, view, fx, fy, cx, cy, w, h, scale_mod=1.0):
    n = positions.shape[0]
    out = dict(means_2d=np.empty((n, 2), np.float32), depths=np.empty(n, np.float32),
               cov_2d_inv=np.empty((n, 3), np.float32), radii=np.empty(n, np.int32),
               tiles_touched=np.empty(n, np.int32), opacities_act=np.empty(n, np.float32))
    pos, rot, scl, opa, vw = _f(positions), _f(rotations), _f(scales), _f(opacities), _f(view)
    _lib.orc_project_forward(C.c_int(n), _p(pos), _p(rot), _p(scl), _p(opa), _p(vw), C.c_float(fx), C.c_float(fy),
                             C.c_float(cx), C.c_float(cy), C.c_int(w), C.c_int(h), C.c_float(scale_mod),
                             _p(out["means_2d"]), _p(out["depths"]), _p(out["cov_2d_inv"]), _p(out["radii"]),
                             _p(out["tiles_touched"]), _p(out["opacities_act"]))
    return out


def directions(positions, cam_center) -> np.ndarray:
    n = positions.shape[0]
    pos, cc = _f(positions), _f(cam_center)
    d = np.empty((n, 3), np.float32)
    _lib.orc_directions(C.c_int(n), _p(pos), _p(cc), _p(d))
    return d


def sh_forward(degree, sh, dirs) -> np.ndarray:
    n, _, c = sh.shape
    s, d = _f(sh), _f(dirs)
    out = np.empty((n, 3), np.float32)
    _lib.orc_sh_forward(C.c_int(degree), C.c_int(n), C.c_int(c), _p(s), _p(d), _p(out))
    return out


def sh_backward(degree, sh, dirs, dL_dcolor) -> np.ndarray:
    n, _, c = sh.shape
    s, d, g = _f(sh), _f(dirs), _f(dL_dcolor)
    out = np.empty((n, 3, c), np.float32)
    _lib.orc_sh_backward(C.c_int(degree), C.c_int(n), C.c_int(c), _p(s), _p(d), _p(g), _p(out))
    return out


def clamp_min0(a: np.ndarray) -> np.ndarray:
    b = _f(a).copy()
    _lib.orc_clamp_min0(C.c_int(b.size), _p(b))
    return b


def count_pairs(tiles_touched) -> int:
    t = _i(tiles_touched)
    return int(_lib.orc_count_pairs(C.c_int(t.shape[0]), _p(t)))


def sort(means_2d, depths, radii, tiles_touched, w, h):
    n = means_2d.shape[0]
    m, d, r, t = _f(means_2d), _f(depths), _i(radii), _i(tiles_touched)
    p = count_pairs(t) if n else 0
    ntiles = ((w + TILE - 1) // TILE) * ((h + TILE - 1) // TILE)
    keys = np.empty(p, np.uint64)
    vals = np.empty(p, np.int32)
    ranges = np.empty((ntiles, 2), np.int32)
    rc = _lib.orc_sort(C.c_int(n), _p(m), _p(d), _p(r), _p(t), C.c_int(w), C.c_int(h), C.c_int64(p), _p(keys),
                       _p(vals), _p(ranges))
    if rc != 0:
        raise RuntimeError("oracle sort: tiles_touched inconsistent with the tile rectangles")
    return dict(keys=keys, values=vals, tile_ranges=ranges, total_pairs=p)


class blend_exp_mode:
    """`with blend_exp_mode(1): ...` - the blend restatement with another exponential (0 contract, 1 Cody-Waite
    cugs_expf(power), 2 libm expf(power)); restores the contract's on exit.  Sensitivity studies only."""

    def __init__(self, mode: int):
        self.mode = int(mode)

    def __enter__(self):
        self.prev = int(_lib.orc_get_blend_exp_mode())
        _lib.orc_set_blend_exp_mode(C.c_int(self.mode))
        return self

    def __exit__(self, *exc):
        _lib.orc_set_blend_exp_mode(C.c_int(self.prev))
        return False


def host_threads() -> int:
    """Host threads this process may use (the GPU box gives a one-GPU job a share of its cores)."""
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except AttributeError:
        return max(1, os.cpu_count() or 1)


def rasterize_forward(w, h, bg, tile_ranges, gidx, means_2d, cov_2d_inv, rgb, opacities, rows=None, threads=1):
    """`threads` > 1: the rows are spread over host threads (every pixel is independent: identical bits)."""
    bg_a = _f(bg)
    tr, gi = _i(tile_ranges), _i(gidx)
    m, c, r, o = _f(means_2d), _f(cov_2d_inv), _f(rgb), _f(opacities)
    color = np.zeros((h, w, 3), np.float32)
    final_T = np.ones((h, w), np.float32)
    n_contrib = np.zeros((h, w), np.int32)
    r0, r1 = (0, h) if rows is None else rows
    if threads > 1:
        _lib.orc_rasterize_forward_rows_mt(C.c_int(threads), C.c_int(w), C.c_int(h), C.c_int(r0), C.c_int(r1), _p(bg_a),
                                           _p(tr), _p(gi), _p(m), _p(c), _p(r), _p(o), _p(color), _p(final_T),
                                           _p(n_contrib))
    else:
        _lib.orc_rasterize_forward_rows(C.c_int(w), C.c_int(h), C.c_int(r0), C.c_int(r1), _p(bg_a), _p(tr), _p(gi),
                                        _p(m), _p(c), _p(r), _p(o), _p(color), _p(final_T), _p(n_contrib))
    return dict(color=color, final_T=final_T, n_contrib=n_contrib)


def rasterize_backward(w, h, bg, tile_ranges, gidx, means_2d, cov_2d_inv, rgb, opacities, dL_dcolor, final_T,
                       n_contrib, n, rows=None, threads=1):
    """`threads` > 1: bands of 64 rows summed in fp64 by one thread each and added in band order - the same result
    for every thread count (orc_rasterize_backward_rows_mt); differs from threads=1 by fp64 association only."""
    bg_a = _f(bg)
    tr, gi = _i(tile_ranges), _i(gidx)
    m, c, r, o = _f(means_2d), _f(cov_2d_inv), _f(rgb), _f(opacities)
    g, ft, nc = _f(dL_dcolor), _f(final_T), _i(n_contrib)
    out = dict(dL_drgb=np.empty((n, 3), np.float32), dL_dopacity_act=np.empty(n, np.float32),
               dL_dmeans_2d=np.empty((n, 2), np.float32), dL_dcov_2d_inv=np.empty((n, 3), np.float32))
    r0, r1 = (0, h) if rows is None else rows
    if threads > 1:
        rc = _lib.orc_rasterize_backward_rows_mt(C.c_int(threads), C.c_int(w), C.c_int(h), C.c_int(r0), C.c_int(r1),
                                                 _p(bg_a), _p(tr), _p(gi), _p(m), _p(c), _p(r), _p(o), _p(g), _p(ft),
                                                 _p(nc), C.c_int(n), _p(out["dL_drgb"]), _p(out["dL_dopacity_act"]),
                                                 _p(out["dL_dmeans_2d"]), _p(out["dL_dcov_2d_inv"]), None)
        if rc != 0:
            raise MemoryError("oracle backward: per-thread accumulator tables do not fit")
        return out
    _lib.orc_rasterize_backward_rows(C.c_int(w), C.c_int(h), C.c_int(r0), C.c_int(r1), _p(bg_a), _p(tr), _p(gi),
                                     _p(m), _p(c), _p(r), _p(o), _p(g), _p(ft), _p(nc), C.c_int(n),
                                     _p(out["dL_drgb"]), _p(out["dL_dopacity_act"]), _p(out["dL_dmeans_2d"]),
                                     _p(out["dL_dcov_2d_inv"]))
    return out


def rasterize_backward_magnitudes(w, h, bg, tile_ranges, gidx, means_2d, cov_2d_inv, rgb, opacities, dL_dcolor,
                                  final_T, n_contrib, n, rows=None, threads=1):
    """rasterize_backward plus `mag` [n, 9] (float64): the sums of the MAGNITUDES of the terms of each accumulated
    gradient - sum |drgb_c| (3), sum |dL_dopa|, sum |dpw dx|, sum |dpw dy|, sum |dpw| dx^2, sum |dpw dx dy|,
    sum |dpw| dy^2 - which bound the rounding error of any fp32 evaluation and summation of those terms.  dL_dopa
    and dpw = dL_dpower enter with the magnitude of the OPERANDS of dL_dalpha (a difference per channel and over the
    channels), not of its value."""
    bg_a = _f(bg)
    tr, gi = _i(tile_ranges), _i(gidx)
    m, c, r, o = _f(means_2d), _f(cov_2d_inv), _f(rgb), _f(opacities)
    g, ft, nc = _f(dL_dcolor), _f(final_T), _i(n_contrib)
    out = dict(dL_drgb=np.empty((n, 3), np.float32), dL_dopacity_act=np.empty(n, np.float32),
               dL_dmeans_2d=np.empty((n, 2), np.float32), dL_dcov_2d_inv=np.empty((n, 3), np.float32),
               mag=np.zeros((n, 9), np.float64))
    if threads > 1 or rows is not None:
        r0, r1 = (0, h) if rows is None else rows
        rc = _lib.orc_rasterize_backward_rows_mt(C.c_int(max(1, threads)), C.c_int(w), C.c_int(h), C.c_int(r0),
                                                 C.c_int(r1), _p(bg_a), _p(tr), _p(gi), _p(m), _p(c), _p(r), _p(o),
                                                 _p(g), _p(ft), _p(nc), C.c_int(n), _p(out["dL_drgb"]),
                                                 _p(out["dL_dopacity_act"]), _p(out["dL_dmeans_2d"]),
                                                 _p(out["dL_dcov_2d_inv"]), _p(out["mag"]))
        if rc != 0:
            raise MemoryError("oracle backward: per-thread accumulator tables do not fit")
        return out
    _lib.orc_rasterize_backward_magnitudes(C.c_int(w), C.c_int(h), _p(bg_a), _p(tr), _p(gi), _p(m), _p(c), _p(r), _p(o),
                                           _p(g), _p(ft), _p(nc), C.c_int(n), _p(out["dL_drgb"]),
                                           _p(out["dL_dopacity_act"]), _p(out["dL_dmeans_2d"]),
                                           _p(out["dL_dcov_2d_inv"]), _p(out["mag"]))
    return out


def project_backward(positions, rotations, scales, opacities, view, fx, fy, cx, cy, scale_mod, radii,
                     dL_dmeans_2d, dL_dcov_2d_inv, dL_dopacity_act):
    n = positions.shape[0]
    pos, rot, scl, opa, vw = _f(positions), _f(rotations), _f(scales), _f(opacities), _f(view)
    rad, gm, gc, go = _i(radii), _f(dL_dmeans_2d), _f(dL_dcov_2d_inv), _f(dL_dopacity_act)
    out = dict(dL_dpositions=np.empty((n, 3), np.float32), dL_drotations=np.empty((n, 4), np.float32),
               dL_dscales=np.empty((n, 3), np.float32), dL_dopacities=np.empty((n, 1), np.float32))
    _lib.orc_project_backward(C.c_int(n), _p(pos), _p(rot), _p(scl), _p(opa), _p(vw), C.c_float(fx), C.c_float(fy),
                              C.c_float(cx), C.c_float(cy), C.c_float(scale_mod), _p(rad), _p(gm), _p(gc), _p(go),
                              _p(out["dL_dpositions"]), _p(out["dL_drotations"]), _p(out["dL_dscales"]),
                              _p(out["dL_dopacities"]))
    return out


def adam_bias_correction(beta1, beta2, step) -> Tuple[float, float]:
    a, b = C.c_float(), C.c_float()
    _lib.orc_adam_bias_correction(C.c_float(beta1), C.c_float(beta2), C.c_int(step), C.byref(a), C.byref(b))
    return a.value, b.value


def fused_adam(param, grad, m, v, lr, beta1, beta2, eps, bc1, bc2) -> None:
    """In place on param, m, v (contiguous float32 arrays)."""
    for a in (param, grad, m, v):
        assert a.dtype == np.float32 and a.flags.c_contiguous
    _lib.orc_fused_adam(C.c_int64(param.size), _p(param), _p(grad), _p(m), _p(v), C.c_float(lr), C.c_float(beta1),
                        C.c_float(beta2), C.c_float(eps), C.c_float(bc1), C.c_float(bc2))


def position_lr(step, lr_init, lr_final, max_steps) -> float:
    return float(_lib.orc_position_lr(C.c_int(step), C.c_float(lr_init), C.c_float(lr_final), C.c_int(max_steps)))


def expf(x: np.ndarray) -> np.ndarray:
    xi = _f(x).reshape(-1)
    y = np.empty_like(xi)
    _lib.orc_expf_array(C.c_int64(xi.size), _p(xi), _p(y))
    return y.reshape(np.shape(x))


def blend_exp_q(q: np.ndarray) -> np.ndarray:
    """cugs_blend_exp_q: the blend's exp(-q/2), clamped below at exp(-6) (include/cugs_detmath.h)."""
    qi = _f(q).reshape(-1)
    y = np.empty_like(qi)
    _lib.orc_blend_exp_q_array(C.c_int64(qi.size), _p(qi), _p(y))
    return y.reshape(np.shape(q))


# ---- whole pipeline on numpy arrays (what tests compare the GPU against) ----------------
def render(model: Dict[str, np.ndarray], rotation, translation, fx, fy, cx, cy, w, h, bg=(0.0, 0.0, 0.0),
           active_degree=3, scale_mod=1.0, rows=None, threads=1) -> Dict[str, np.ndarray]:
    """rasterizer.cpp:22-113 on the oracle's stages."""
    vw = view_matrix(rotation, translation)
    cc = (-(np.asarray(rotation, np.float32).T @ np.asarray(translation, np.float32))).astype(np.float32)
    c = model["sh_coeffs"].shape[2]
    deg = min(active_degree, int(np.sqrt(np.float32(c))) - 1)
    proj = project_forward(model["positions"], model["rotations"], model["scales"], model["opacities"], vw, fx, fy,
                           cx, cy, w, h, scale_mod)
    dirs = directions(model["positions"], cc)
    rgb = clamp_min0(sh_forward(deg, model["sh_coeffs"], dirs))
    srt = sort(proj["means_2d"], proj["depths"], proj["radii"], proj["tiles_touched"], w, h)
    fwd = rasterize_forward(w, h, bg, srt["tile_ranges"], srt["values"], proj["means_2d"], proj["cov_2d_inv"], rgb,
                            proj["opacities_act"], rows=rows, threads=threads)
    out = dict(proj)
    out.update(rgb=rgb, dirs=dirs, view=vw, cam_center=cc, degree=deg, **srt, **fwd)
    return out


def render_backward(dL_dcolor, fwd: Dict[str, np.ndarray], model: Dict[str, np.ndarray], fx, fy, cx, cy, w, h,
                    bg=(0.0, 0.0, 0.0), scale_mod=1.0, rows=None, threads=1) -> Dict[str, np.ndarray]:
    """rasterizer.cpp:115-186 on the oracle's stages."""
    n = model["positions"].shape[0]
    rb = rasterize_backward(w, h, bg, fwd["tile_ranges"], fwd["values"], fwd["means_2d"], fwd["cov_2d_inv"],
                            fwd["rgb"], fwd["opacities_act"], dL_dcolor, fwd["final_T"], fwd["n_contrib"], n,
                            rows=rows, threads=threads)
    pb = project_backward(model["positions"], model["rotations"], model["scales"], model["opacities"], fwd["view"],
                          fx, fy, cx, cy, scale_mod, fwd["radii"], rb["dL_dmeans_2d"], rb["dL_dcov_2d_inv"],
                          rb["dL_dopacity_act"])
    d_sh = sh_backward(fwd["degree"], model["sh_coeffs"], fwd["dirs"], rb["dL_drgb"])
    out = dict(pb)
    out.update(dL_dsh_coeffs=d_sh, **rb)
    return out


def project_sh_forward_mt(nthreads, model, rotation, translation, fx, fy, cx, cy, w, h, degree) -> int:
    """bench.py cpu_baseline leg: projection + SH for all Gaussians on `nthreads` host threads."""
    n = model["positions"].shape[0]
    c = model["sh_coeffs"].shape[2]
    vw = view_matrix(rotation, translation)
    cc = (-(np.asarray(rotation, np.float32).T @ np.asarray(translation, np.float32))).astype(np.float32)
    pos, rot, scl, opa, sh = (_f(model[k]) for k in ("positions", "rotations", "scales", "opacities", "sh_coeffs"))
    bufs = [np.empty((n, 2), np.float32), np.empty(n, np.float32), np.empty((n, 3), np.float32),
            np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.float32), np.empty((n, 3), np.float32),
            np.empty((n, 3), np.float32)]
    return int(_lib.orc_project_sh_forward_mt(C.c_int(nthreads), C.c_int(n), C.c_int(degree), C.c_int(c), _p(pos),
                                              _p(rot), _p(scl), _p(opa), _p(sh), _p(vw), _p(cc), C.c_float(fx),
                                              C.c_float(fy), C.c_float(cx), C.c_float(cy), C.c_int(w), C.c_int(h),
                                              C.c_float(1.0), *[_p(b) for b in bufs]))


# ---- oracle/_ref: the reference's own evaluate_sh_cpu --------------------------------------
def ref_sh_available() -> bool:
    return os.path.exists(REF_SH_PATH)


_ref = None


def ref_evaluate_sh_cpu(degree, sh, dirs):
    """cugs::evaluate_sh_cpu (src/core/sh.cpp:8-87) compiled from the reference tree.
    Returns (rc, rgb); rc != 0 mirrors a TORCH_CHECK failure."""
    global _ref
    if _ref is None:
        import torch  # noqa: F401  (libref_sh.so links libtorch; load its libraries first)
        _ref = C.CDLL(REF_SH_PATH)
        _ref.ref_evaluate_sh_cpu.restype = C.c_int
    n, _, c = sh.shape
    s, d = _f(sh), _f(dirs)
    out = np.empty((n, 3), np.float32)
    rc = _ref.ref_evaluate_sh_cpu(C.c_int(degree), C.c_int64(n), C.c_int(c), _p(s), _p(d), _p(out))
    return rc, out
