"""reduce9t (cugs_raster_common.h): the 9-value wave64 transpose-reduce built on bank-masked DPP adds and
gfx950's v_permlane16_swap / v_permlane32_swap.  Exercised alone through the DEVELOPMENT build's test hook
(libcugs_hip_dev.so: cugsdbg_reduce9, not part of the public C ABI) with exact integer-valued data; and
v_rcp_f32(1.0f) == 1.0f, which the blend backward relies on for pixels a Gaussian does not touch."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev_lib(pkg):
    path = os.path.join(os.path.dirname(pkg.LIB_PATH), "libcugs_hip_dev.so")
    if not os.path.exists(path):
        pytest.skip("development library not built (make -C cuda-gaussian-splatting_amd/csrc)")
    return C.CDLL(path)


def test_reduce9_totals_land_in_their_lanes(pkg, dev):
    lib = _dev_lib(pkg)
    rng = np.random.default_rng(0)
    for trial in range(6):
        vals = rng.integers(-500, 500, size=(9, 64)).astype(np.float32)     # exact in fp32
        if trial == 0:
            vals = np.array([[1000.0 * (k + 1) + l for l in range(64)] for k in range(9)], np.float32)
        if trial == 1:                                                      # one hot lane per slot: catches lane mix-ups
            vals = np.zeros((9, 64), np.float32)
            for k in range(9):
                vals[k, (7 * k + 3) % 64] = float(k + 1)
        inp = torch.from_numpy(vals).to(dev)
        out = torch.zeros(64, device=dev)
        slots = torch.zeros(64, dtype=torch.int32, device=dev)
        rc = lib.cugsdbg_reduce9(C.c_void_p(inp.data_ptr()), C.c_void_p(out.data_ptr()),
                                 C.c_void_p(slots.data_ptr()), C.c_void_p(0))
        assert rc == 0
        torch.cuda.synchronize()
        o, s = out.cpu().numpy(), slots.cpu().numpy()
        want = vals.sum(1)
        assert sorted(s[s >= 0].tolist()) == list(range(9))            # every slot delivered by exactly one lane
        for lane in range(64):
            if s[lane] >= 0:
                assert o[lane] == want[s[lane]], (trial, lane, s[lane])


def test_reduce9r16_rows_are_independent(pkg, dev):
    """reduce9r16: the same reduction inside each 16-lane row (the blend backward's second phase: one Gaussian per
    row): every row delivers its own nine totals, nothing leaks between rows."""
    lib = _dev_lib(pkg)
    rng = np.random.default_rng(1)
    for trial in range(4):
        vals = rng.integers(-500, 500, size=(9, 64)).astype(np.float32)
        if trial == 0:
            vals = np.array([[1000.0 * (k + 1) + l for l in range(64)] for k in range(9)], np.float32)
        if trial == 1:
            vals[:, 16:32] = 0.0                                            # an empty row stays exactly zero
        inp = torch.from_numpy(vals).to(dev)
        out = torch.zeros(64, device=dev)
        slots = torch.zeros(64, dtype=torch.int32, device=dev)
        assert lib.cugsdbg_reduce9r16(C.c_void_p(inp.data_ptr()), C.c_void_p(out.data_ptr()),
                                      C.c_void_p(slots.data_ptr()), C.c_void_p(0)) == 0
        torch.cuda.synchronize()
        o, s = out.cpu().numpy(), slots.cpu().numpy()
        for row in range(4):
            sl = s[row * 16:(row + 1) * 16]
            assert sorted(sl[sl >= 0].tolist()) == list(range(9))
            want = vals[:, row * 16:(row + 1) * 16].sum(1)
            for r in range(16):
                if sl[r] >= 0:
                    assert o[row * 16 + r] == want[sl[r]], (trial, row, r, sl[r])


def test_rcp_of_one_is_exactly_one(pkg, dev):
    lib = _dev_lib(pkg)
    x = torch.tensor([1.0, 0.5, 2.0, 0.25, 0.01, 0.99, 1e-5], dtype=torch.float32, device=dev)
    out = torch.zeros_like(x)
    rc = lib.cugsdbg_rcp(C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), C.c_int(x.numel()), C.c_void_p(0))
    assert rc == 0
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    assert o[0] == 1.0 and o[1] == 2.0 and o[2] == 0.5 and o[3] == 4.0      # powers of two are exact
    assert np.allclose(o[4:], 1.0 / x.cpu().numpy()[4:], rtol=3e-7)          # 1 ulp elsewhere


def test_blend_exponential_has_the_oracles_bits(pkg, orc, dev):
    """cugs_blend_exp_q is the one transcendental the blend DECISIONS depend on: the device version (v_med3, v_fma,
    v_lshl_add_u32) must return the host version's bits for every input the kernels can feed it - the whole range of
    the quadratic form, values around the clamp at q = 12, tiny, negative, denormal and huge ones."""
    lib = _dev_lib(pkg)
    rng = np.random.default_rng(7)
    q = np.concatenate([rng.uniform(0.0, 12.5, 2_000_000), np.linspace(11.9, 12.1, 100_001),
                        np.float32(2.0) ** -np.arange(0, 150, dtype=np.float32), -np.float32(2.0) ** -np.arange(0, 150, dtype=np.float32),
                        [0.0, -0.0, 12.0, 1e9, 3e38, np.inf, 1e-45, -1e-45]]).astype(np.float32)
    x = torch.from_numpy(q).to(dev)
    out = torch.empty_like(x)
    rc = lib.cugsdbg_blend_exp_q(C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), C.c_int(x.numel()), C.c_void_p(0))
    assert rc == 0
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().view(np.uint32), orc.blend_exp_q(q).view(np.uint32))
