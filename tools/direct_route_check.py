"""Development check of the sort's direct-binning route (k_bin_*): a sequence of scenes of changing size on ONE stream
(so that predicted capacities come from the previous scene: hits and misses), each rendered with the route on and off
(development library: cugsdbg_sort_direct_route) and compared bit for bit.
    CUGS_HIP_LIBRARY=.../libcugs_hip_dev.so python tools/direct_route_check.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

pkg = ge.load_package()
lib = C.CDLL(pkg.LIB_PATH)
dev = torch.device("cuda", 0)
cases = [(5000, 200, 150, -3.8), (5000, 200, 150, -3.8), (9687, 200, 150, -3.8), (9687, 200, 150, -3.8),
         (300, 64, 64, -2.0), (20000, 640, 360, -4.6), (20000, 640, 360, -4.6), (100000, 1920, 1080, -4.6),
         (100000, 1920, 1080, -3.0), (100000, 1920, 1080, -3.0), (1000, 1920, 1080, 0.5), (1000, 1920, 1080, 0.5)]
bad = 0
for n, w, h, mu in cases:
    arrays = pkg.scene.make_gaussians(n, w, h, sh_degree=1, seed=n + w, mu_s=mu)
    cam = pkg.scene.make_camera(w, h)
    model = pkg.scene.to_model(arrays, dev)
    st = pkg.RenderSettings(active_sh_degree=1)
    outs = []
    for on in (1, 0, 1):
        lib.cugsdbg_sort_direct_route(on)
        o = pkg.render(model, cam, st)
        torch.cuda.synchronize()
        outs.append(o)
    a, b, c = outs
    same = (a.total_pairs == b.total_pairs and torch.equal(a.gaussian_indices, b.gaussian_indices)
            and torch.equal(a.tile_ranges, b.tile_ranges) and torch.equal(a.color, b.color)
            and torch.equal(c.gaussian_indices, b.gaussian_indices) and torch.equal(c.tile_ranges, b.tile_ranges))
    print(f"n={n} {w}x{h} mu_s={mu}: pairs {a.total_pairs} {'same' if same else 'DIFFERENT'}", flush=True)
    bad += not same
print("bad", bad)
sys.exit(1 if bad else 0)
