"""Pins the oracle's SH restatement against REFERENCE-AUTHORED code: cugs::evaluate_sh_cpu
(src/core/sh.cpp:8-87), compiled in place from /root/reference into oracle/_ref/libref_sh.so by
oracle/Makefile, and against the reference's own SH tests (tests/test_sh.cpp:16-143)."""
import numpy as np
import pytest

C0 = 0.28209479177387814


@pytest.fixture(scope="module")
def ref(orc):
    if not orc.ref_sh_available():
        pytest.skip("oracle/_ref/libref_sh.so not built (needs /root/reference; built in the dev container)")
    return orc


@pytest.mark.parametrize("degree", [0, 1, 2, 3])
@pytest.mark.parametrize("n", [1, 200, 10_000])
def test_oracle_sh_equals_reference_cpu(ref, degree, n):
    rng = np.random.default_rng(100 * degree + n)
    sh = rng.standard_normal((n, 3, 16)).astype(np.float32)
    d = rng.standard_normal((n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rc, want = ref.ref_evaluate_sh_cpu(degree, sh, d)
    assert rc == 0
    got = ref.sh_forward(degree, sh, d)
    # the reference's own CUDA==CPU bar is rtol=atol=1e-4 (test_sh.cpp:161-216); the restatement
    # follows the same association so it is far tighter (g++ may contract differently: allow 1e-6)
    assert np.allclose(got, want, rtol=1e-6, atol=1e-6)


def test_reference_kats_hold_for_both(ref):
    """test_sh.cpp:16-125 known answers, evaluated on the reference build AND the oracle."""
    for evaluate in (lambda deg, s, d: ref.ref_evaluate_sh_cpu(deg, s, d)[1], ref.sh_forward):
        # DC only: colour = C0 * c0 + 0.5, exact to 1e-5 (test_sh.cpp:16-38)
        sh = np.zeros((1, 3, 16), np.float32)
        sh[0, :, 0] = [1.0, 0.5, -0.25]
        d = np.array([[0.0, 0.0, 1.0]], np.float32)
        out = evaluate(0, sh, d)
        assert np.allclose(out[0], C0 * sh[0, :, 0] + 0.5, atol=1e-5)
        # degree 0 is direction independent (:40-58)
        d2 = np.array([[1.0, 0.0, 0.0]], np.float32)
        assert np.array_equal(evaluate(0, sh, d2), out)
        # Y_1^-1 is antisymmetric in y (:60-84)
        sh1 = np.zeros((1, 3, 16), np.float32)
        sh1[0, :, 1] = 1.0
        up = evaluate(1, sh1, np.array([[0.0, 1.0, 0.0]], np.float32))
        dn = evaluate(1, sh1, np.array([[0.0, -1.0, 0.0]], np.float32))
        assert np.allclose(up - 0.5, -(dn - 0.5), atol=1e-6)
        # degree 3 with zero higher coefficients == degree 0 (:106-125)
        assert np.allclose(evaluate(3, sh, d), out, atol=1e-6)
        # batch == per item (:86-104)
        rng = np.random.default_rng(4)
        shb = rng.standard_normal((5, 3, 16)).astype(np.float32)
        db = rng.standard_normal((5, 3)).astype(np.float32)
        db /= np.linalg.norm(db, axis=1, keepdims=True)
        full = evaluate(3, shb, db)
        for i in range(5):
            assert np.array_equal(evaluate(3, shb[i:i + 1], db[i:i + 1])[0], full[i])


def test_reference_input_validation(ref):
    """test_sh.cpp:127-143: bad degree / too few coefficients throw c10::Error in the reference."""
    sh = np.zeros((2, 3, 4), np.float32)
    d = np.zeros((2, 3), np.float32)
    assert ref.ref_evaluate_sh_cpu(4, sh, d)[0] == 1
    assert ref.ref_evaluate_sh_cpu(2, sh, d)[0] == 1
    assert ref.ref_evaluate_sh_cpu(1, sh, d)[0] == 0
