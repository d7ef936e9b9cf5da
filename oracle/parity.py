"""Parity yardsticks shared by tests/, __graft_entry__.smoke() and bench.py's parity probe.

TEST INFRASTRUCTURE (like everything under oracle/): nothing in the product package imports this module.

Two numbers per tensor, always reported side by side:
  rel_8d      SURVEY.md §8d: max over elements of |g - r| / max(|r|, tau), tau = 1e-6 * max|r| - the contract's metric.
  over_scale  max|g - r| / max|r| - the error against the tensor's own scale.
They differ where a gradient element is a CANCELLING sum: the blend backward adds thousands of signed fp32 terms per
Gaussian; an element whose terms cancel to 1e-4 of their magnitudes cannot come out of ANY fp32 summation (the
reference's own atomics included) within 1e-4 of the RESULT.  `blend_accumulator_report` therefore does not take that on
trust: for every element of the four 2-D accumulators whose rel_8d error exceeds the bar it checks |g - r| against
B * sum|terms| with the sum of magnitudes the oracle lists (orc_rasterize_backward_magnitudes), and names the worst.
(The projection backward behind the accumulators is the oracle's operation order on identical inputs - bit-identical
stage test - so everything the GPU can differ by sits in those accumulators.)
"""
from __future__ import annotations

import numpy as np

BAR = 1e-4
FLOOR_FRAC = 1e-6


def _flat64(a):
    return np.asarray(a, np.float64).reshape(-1)


def rel_8d(got, ref, floor_frac=FLOOR_FRAC) -> float:
    g, r = _flat64(got), _flat64(ref)
    if r.size == 0:
        return 0.0
    tau = floor_frac * max(float(np.max(np.abs(r))), 1e-300)
    return float(np.max(np.abs(g - r) / np.maximum(np.abs(r), tau)))


def over_scale(got, ref) -> float:
    g, r = _flat64(got), _flat64(ref)
    if r.size == 0:
        return 0.0
    return float(np.max(np.abs(g - r)) / max(float(np.max(np.abs(r))), 1e-300))


def tensor_report(got, ref, bar=BAR, floor_frac=FLOOR_FRAC) -> dict:
    """Both yardsticks, how many elements exceed the bar element-wise, and the worst such element."""
    g, r = _flat64(got), _flat64(ref)
    if r.size == 0:
        return {"rel_8d": 0.0, "over_scale": 0.0, "elements": 0, "over_bar": 0}
    scale = max(float(np.max(np.abs(r))), 1e-300)
    rel = np.abs(g - r) / np.maximum(np.abs(r), floor_frac * scale)
    j = int(np.argmax(rel))
    return {"rel_8d": float(rel[j]), "over_scale": float(np.max(np.abs(g - r)) / scale), "elements": int(r.size),
            "over_bar": int(np.count_nonzero(rel > bar)),
            "worst": {"index": j, "got": float(g[j]), "want": float(r[j]), "of_scale": float(abs(r[j]) / scale)}}


def gradient_report(grads: dict, ref: dict, names, bar=BAR) -> dict:
    """{name: tensor_report} plus the two maxima the bench line carries."""
    per = {n: tensor_report(grads[n], np.asarray(ref[n]).reshape(np.asarray(grads[n]).shape), bar) for n in names}
    return {"per_tensor": per,
            "grad_max_rel_err_8d": max(v["rel_8d"] for v in per.values()),
            "grad_max_err_over_scale": max(v["over_scale"] for v in per.values()),
            "elements_over_bar_8d": int(sum(v["over_bar"] for v in per.values())),
            "elements": int(sum(v["elements"] for v in per.values()))}


ACCUMULATORS = ("dL_drgb", "dL_dopacity_act", "dL_dmeans_2d", "dL_dcov_2d_inv")


def blend_accumulator_report(got: dict, want: dict, mag, cov_2d_inv, list_entries, bar=BAR) -> dict:
    """The four 2-D accumulators of the blend backward (reference layout: k_unpack_grads on the GPU side) against the
    oracle's fp64 sums, element-wise, with the cancellation evidence.

    got / want: dicts of the four tensors; mag [n, 9] float64 from orc.rasterize_backward_magnitudes (sum |drgb_c| (3),
    sum |dL_dopa|, sum |dpw dx|, sum |dpw dy|, sum |dpw| dx^2, sum |dpw dx dy|, sum |dpw| dy^2); cov_2d_inv [n, 3];
    list_entries [n]: how often each Gaussian stands in a tile list (one atomic rounding per (quad wave, entry)).

    Bound on any fp32 evaluation and summation of the listed terms: B * sum|terms| with B = (32 + 4 entries) * 2^-24
    (32 ulp cover the per-term rounding of the v_rcp / FMA forms and the in-wave summation tree; one rounding per
    atomic add comes on top).  The GPU sums the geometry as moments, so the bound of dL/dmean2d combines the moment
    magnitudes with |a|, |b|, |c|.
    Returns per tensor: both yardsticks, the number of elements over the bar, how many of THOSE exceed the bound
    (must be 0: a deviation the terms' magnitudes do not explain would be a wrong gradient), and the worst ratio."""
    n = mag.shape[0]
    entries = np.asarray(list_entries, np.float64).reshape(n)
    B = ((32.0 + 4.0 * entries) * 2.0 ** -24)[:, None]
    a, b, c = (np.abs(np.asarray(cov_2d_inv, np.float64)[:, i]) for i in range(3))
    bound = {
        "dL_drgb": mag[:, 0:3],
        "dL_dopacity_act": mag[:, 3:4],
        "dL_dmeans_2d": np.stack([a * mag[:, 4] + b * mag[:, 5], b * mag[:, 4] + c * mag[:, 5]], axis=1),
        "dL_dcov_2d_inv": np.stack([0.5 * mag[:, 6], mag[:, 7], 0.5 * mag[:, 8]], axis=1),
    }
    out = {}
    for name, m in bound.items():
        g = np.asarray(got[name], np.float64).reshape(m.shape)
        r = np.asarray(want[name], np.float64).reshape(m.shape)
        rep = tensor_report(g, r, bar)
        scale = max(float(np.max(np.abs(r))), 1e-300)
        diff = np.abs(g - r)
        rel = diff / np.maximum(np.abs(r), FLOOR_FRAC * scale)
        allowed = B * m + 1e-37
        over = rel > bar
        rep["over_bar_beyond_term_bound"] = int(np.count_nonzero(over & (diff > allowed)))
        rep["beyond_term_bound"] = int(np.count_nonzero(diff > allowed))
        ratio = diff / allowed
        j = np.unravel_index(int(np.argmax(ratio)), m.shape)
        rep["worst_diff_over_bound"] = float(ratio[j])
        if rep["over_bar"]:
            k = np.unravel_index(int(np.argmax(np.where(over, rel, 0.0))), m.shape)
            rep["worst_over_bar"] = {"gaussian": int(k[0]), "component": int(k[1]), "rel_8d": float(rel[k]),
                                     "diff": float(diff[k]), "value": float(r[k]), "sum_of_term_magnitudes": float(m[k]),
                                     "diff_over_bound": float(ratio[k])}
        out[name] = rep
    return out


def format_report(rep: dict, title: str = "") -> str:
    """Human-readable table (printed by tests with -s, by smoke() and by tools)."""
    lines = [title] if title else []
    per = rep.get("per_tensor", rep)
    for name, v in per.items():
        if not isinstance(v, dict) or "rel_8d" not in v:
            continue
        s = "  %-18s rel_8d %.3e  over_scale %.3e  %d of %d elements over %.0e" % (
            name, v["rel_8d"], v["over_scale"], v["over_bar"], v["elements"], BAR)
        if "over_bar_beyond_term_bound" in v:
            s += "; %d of them beyond the fp32 bound of their terms (worst diff/bound %.3g)" % (
                v["over_bar_beyond_term_bound"], v["worst_diff_over_bound"])
        if v.get("worst_over_bar"):
            w = v["worst_over_bar"]
            s += "\n      worst: Gaussian %d component %d value %.3e diff %.3e sum|terms| %.3e (cancels to %.1e of them)" % (
                w["gaussian"], w["component"], w["value"], w["diff"], w["sum_of_term_magnitudes"],
                abs(w["value"]) / max(w["sum_of_term_magnitudes"], 1e-300))
        lines.append(s)
    return "\n".join(lines)
