#!/usr/bin/env python3
"""Times SURVEY §8(f) N3 at 1 M Gaussians / SH 3: the device-side record pack/unpack (csrc/ply.hip) and the
whole write/read, beside the oracle's numpy interleave on the host (the reference itself issues 62 ofstream
writes per Gaussian, utils/ply_io.cpp:156-190)."""
import os, sys, tempfile, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
from test_ply_oracle import load_ply_oracle, make_model
import ctypes as C
pkg = ge.load_package(); po = load_ply_oracle()
from cugs_amd._lib import lib
dev = torch.device("cuda:0"); n, c = 1_000_000, 16
ref = make_model(n, c)
model = pkg.GaussianModel(**{k: torch.from_numpy(v).to(dev) for k, v in ref.items()})
names = ("positions", "sh_coeffs", "opacities", "scales", "rotations")
arr = (C.c_void_p * 5)(*[getattr(model, k).data_ptr() for k in names])
verts = torch.empty((n, 62), device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def pack(): lib.cugs_ply_pack(n, c, arr, None, None, C.c_void_p(verts.data_ptr()), st)
for _ in range(3): pack()
torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); [pack() for _ in range(20)]; e1.record(); torch.cuda.synchronize()
t_pack = e0.elapsed_time(e1) / 20
t0 = time.perf_counter(); host = po.vertex_array(ref); t_np = (time.perf_counter() - t0) * 1e3
print("record pack, 1 M x 62 floats: HIP %.3f ms (%.0f GB/s of 121 x 4 B/Gaussian)   numpy interleave on the host %.0f ms" %
      (t_pack, n * 121 * 4 / t_pack / 1e6, t_np))
with tempfile.TemporaryDirectory(dir="/dev/shm") as d:
    p = os.path.join(d, "m.ply")
    torch.cuda.synchronize(); t0 = time.perf_counter(); pkg.write_gaussian_ply(p, model); t_w = time.perf_counter() - t0
    t0 = time.perf_counter(); back = pkg.read_gaussian_ply(p, device=dev); torch.cuda.synchronize(); t_r = time.perf_counter() - t0
    t0 = time.perf_counter(); po.write_gaussian_ply(os.path.join(d, "o.ply"), ref); t_ow = time.perf_counter() - t0
    t0 = time.perf_counter(); po.read_gaussian_ply(p); t_or = time.perf_counter() - t0
    assert open(p, "rb").read() == open(os.path.join(d, "o.ply"), "rb").read()
    print("whole file (248 MB, tmpfs): write %.0f ms, read to device %.0f ms   oracle (numpy, host): write %.0f ms, read %.0f ms"
          % (t_w * 1e3, t_r * 1e3, t_ow * 1e3, t_or * 1e3))
