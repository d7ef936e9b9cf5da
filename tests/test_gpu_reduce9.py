"""reduce9 (cugs_raster_common.h): the 9-value wave64 transpose-reduce built on gfx950's
v_permlane32_swap / v_permlane16_swap + DPP.  Exercised alone through the library's test hook
(cugsdbg_reduce9, not part of the public C ABI) with exact integer-valued data."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_reduce9_totals_land_in_their_lanes(pkg, dev):
    lib = C.CDLL(pkg.LIB_PATH)
    rng = np.random.default_rng(0)
    for trial in range(4):
        vals = rng.integers(-500, 500, size=(9, 64)).astype(np.float32)     # exact in fp32
        if trial == 0:
            vals = np.array([[1000.0 * (k + 1) + l for l in range(64)] for k in range(9)], np.float32)
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
