"""include/cugs_detmath.h: the deterministic exp shared by the oracle and the kernels."""
import numpy as np


def _ulp(x):
    return np.spacing(np.abs(x).astype(np.float32)).astype(np.float64)


def test_expf_within_one_ulp_of_libm(orc):
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-87.0, 88.0, 600_000), rng.uniform(-6.0, 0.0, 400_000),
                        np.linspace(-1e-3, 1e-3, 1001)]).astype(np.float32)
    got = orc.expf(x).astype(np.float64)
    want = np.exp(x.astype(np.float64))
    err_ulp = np.abs(got - want) / _ulp(want.astype(np.float32))
    assert err_ulp.max() <= 1.0, err_ulp.max()


def test_expf_edges(orc):
    x = np.array([0.0, -0.0, 88.8, 1000.0, -87.4, -1000.0, np.inf, -np.inf], np.float32)
    y = orc.expf(x)
    assert y[0] == 1.0 and y[1] == 1.0
    assert np.isinf(y[2]) and np.isinf(y[3]) and np.isinf(y[6])
    assert y[4] == 0.0 and y[5] == 0.0 and y[7] == 0.0
    assert np.isnan(orc.expf(np.array([np.nan], np.float32))[0])


def test_expf_never_exceeds_one_for_nonpositive_input(orc):
    """The blend relies on alpha = o*exp(power) <= o for power <= 0 (cugs_raster_common.h)."""
    x = -np.abs(np.random.default_rng(1).standard_normal(500_000).astype(np.float32)) * 3
    x = np.concatenate([x, -np.float32(2.0) ** -np.arange(1, 60, dtype=np.float32)])
    assert orc.expf(x).max() <= 1.0


def test_expf_preskip_bound(orc):
    """power < -5.6 can never reach alpha >= 1/255 with opacity <= 1 (pixel_alpha's pre-skip)."""
    x = np.float32(-5.6)
    assert float(orc.expf(np.array([x]))[0]) < 1.0 / 255.0
    xs = np.linspace(-87.0, -5.6, 200_001).astype(np.float32)
    assert orc.expf(xs).max() < 1.0 / 255.0


def test_blend_exp_q_against_libm(orc):
    """cugs_blend_exp_q(q) = exp(-q/2) for q in [0, 12] (base-2 range reduction + degree-5 polynomial, what the blend
    kernels and the oracle evaluate per (pixel, Gaussian) pair): within 1e-6 relative of the exact value, non-increasing
    up to its own error, exactly 1 at q <= 0, and clamped at exp(-6) beyond q = 12 - where opacity * it < 1/255 whatever
    the opacity, so the pair is skipped either way."""
    rng = np.random.default_rng(3)
    q = np.concatenate([rng.uniform(0.0, 12.0, 2_000_000), np.linspace(0.0, 12.0, 400_001),
                        np.float32(2.0) ** -np.arange(0, 40, dtype=np.float32)]).astype(np.float32)
    got = orc.blend_exp_q(q).astype(np.float64)
    want = np.exp(-0.5 * q.astype(np.float64))
    assert np.max(np.abs(got - want) / want) <= 1.0e-6
    qs = np.sort(q)
    e = orc.blend_exp_q(qs).astype(np.float64)
    assert np.all(np.diff(e) <= 4e-7 * e[1:])                 # non-increasing up to the polynomial's own error
    edge = orc.blend_exp_q(np.array([0.0, -0.0, -1.0, -1e-30, 12.0, 13.0, 1e9, np.inf], np.float32))
    assert edge[0] == 1.0 and edge[1] == 1.0 and edge[2] == 1.0 and edge[3] == 1.0
    assert edge[4] == edge[5] == edge[6] == edge[7] and abs(float(edge[4]) / np.exp(-6.0) - 1.0) <= 1e-6
    assert float(edge[4]) < 1.0 / 255.0
