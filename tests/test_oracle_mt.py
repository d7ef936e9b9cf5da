"""The oracle's threaded blend (full-frame checks at 1080p) and the parity yardsticks - CPU only.

orc_rasterize_forward_rows_mt must give the serial function's bits; orc_rasterize_backward_rows_mt must give the same
result for every thread count (fixed 64-row bands, added in band order) and agree with the one-table serial sum to
fp64 association.  oracle/parity.py: the two yardsticks and the cancellation evidence."""
import numpy as np

from util import load_parity


def _scene(pkg, orc, n=6000, w=320, h=200, seed=3, threads=1):
    arrays = pkg.scene.make_gaussians(n, w, h, sh_degree=2, seed=seed, mu_s=-3.6)
    cam = pkg.scene.make_camera(w, h)
    K = cam.intrinsics
    ref = orc.render(arrays, cam.rotation, cam.translation, K.fx, K.fy, K.cx, K.cy, w, h, active_degree=2,
                     bg=(0.1, 0.2, 0.3), threads=threads)
    return arrays, cam, K, ref, w, h, n


def test_threaded_forward_is_bit_identical(pkg, orc):
    *_, ref1, w, h, n = _scene(pkg, orc, threads=1)
    *_, ref5, w, h, n = _scene(pkg, orc, threads=5)
    for k in ("color", "final_T", "n_contrib"):
        assert np.array_equal(ref1[k].view(np.uint32), ref5[k].view(np.uint32)), k


def test_threaded_backward_is_thread_count_independent_and_equals_serial(pkg, orc):
    arrays, cam, K, ref, w, h, n = _scene(pkg, orc)
    g = pkg.scene.make_dl_dcolor(w, h)
    kw = dict(bg=(0.1, 0.2, 0.3))
    b1 = orc.render_backward(g, ref, arrays, K.fx, K.fy, K.cx, K.cy, w, h, **kw)
    b2 = orc.render_backward(g, ref, arrays, K.fx, K.fy, K.cx, K.cy, w, h, threads=2, **kw)
    b7 = orc.render_backward(g, ref, arrays, K.fx, K.fy, K.cx, K.cy, w, h, threads=7, **kw)
    for k in b1:
        assert np.array_equal(b2[k].view(np.uint32), b7[k].view(np.uint32)), k        # any thread count: same bits
        scale = max(float(np.abs(b1[k]).max()), 1e-30)
        assert float(np.abs(b1[k].astype(np.float64) - b2[k]).max()) <= 1e-6 * scale, k   # fp64 association + one rounding
    # rows restricted: the band split starts at row0
    r = (32, 150)
    s1 = orc.render_backward(g, ref, arrays, K.fx, K.fy, K.cx, K.cy, w, h, rows=r, **kw)
    s4 = orc.render_backward(g, ref, arrays, K.fx, K.fy, K.cx, K.cy, w, h, rows=r, threads=4, **kw)
    for k in s1:
        scale = max(float(np.abs(s1[k]).max()), 1e-30)
        assert float(np.abs(s1[k].astype(np.float64) - s4[k]).max()) <= 1e-6 * scale, k


def test_threaded_magnitudes_match_serial(pkg, orc):
    arrays, cam, K, ref, w, h, n = _scene(pkg, orc)
    g = pkg.scene.make_dl_dcolor(w, h)
    a = (w, h, (0.1, 0.2, 0.3), ref["tile_ranges"], ref["values"], ref["means_2d"], ref["cov_2d_inv"], ref["rgb"],
         ref["opacities_act"], g, ref["final_T"], ref["n_contrib"], n)
    m1 = orc.rasterize_backward_magnitudes(*a)
    m3 = orc.rasterize_backward_magnitudes(*a, threads=3)
    assert np.allclose(m1["mag"], m3["mag"], rtol=1e-12, atol=0)
    assert np.all(m1["mag"] >= 0)
    # a magnitude sum bounds the value it belongs to
    assert np.all(np.abs(m1["dL_drgb"]) <= m1["mag"][:, 0:3] * (1 + 1e-6) + 1e-30)


def test_yardsticks_and_cancellation_evidence():
    par = load_parity()
    ref = np.array([1.0, 1e-3, 1e-9, 0.0])
    got = ref + np.array([1e-6, 1e-6, 1e-6, 1e-6])
    assert abs(par.over_scale(got, ref) - 1e-6) < 1e-12
    # element-wise with tau = 1e-6 * max|ref|: the small elements are measured against tau = 1e-6
    assert abs(par.rel_8d(got, ref) - 1.0) < 1e-3
    rep = par.tensor_report(got, ref)
    assert rep["over_bar"] == 3 and rep["elements"] == 4
    # accumulator report: a deviation inside B * sum|terms| is a cancelling sum, one outside is flagged
    n = 3
    want = {"dL_drgb": np.array([[1.0, 1e-7, 1.0]] * n), "dL_dopacity_act": np.ones(n), "dL_dmeans_2d": np.ones((n, 2)),
            "dL_dcov_2d_inv": np.ones((n, 3))}
    got = {k: v.copy() for k, v in want.items()}
    got["dL_drgb"][0, 1] += 5e-8                       # 50 % off a value that is 1e-7 of its terms' magnitudes
    got["dL_drgb"][1, 0] += 1e-3                       # 1e-3 off a value whose terms sum to 1
    mag = np.ones((n, 9))
    rep = par.blend_accumulator_report(got, want, mag, np.ones((n, 3)), np.ones(n))
    assert rep["dL_drgb"]["over_bar"] == 2
    assert rep["dL_drgb"]["over_bar_beyond_term_bound"] == 1           # only the second one is unexplained
    assert rep["dL_dopacity_act"]["over_bar"] == 0
    assert "worst_over_bar" in rep["dL_drgb"] and par.format_report(rep)
