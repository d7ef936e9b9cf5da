#!/usr/bin/env python3
"""Dev tool: what the colour_gate output costs the projection, and what it saves the projection backward
(config 3; same process, alternating).   python tools/bench_projection_gate.py"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); dev = torch.device("cuda:0")
from importlib import import_module
L = pkg._lib
wl = pkg.scene.CONFIGS["config3"]
arr = pkg.scene.make_gaussians(wl.n, wl.width, wl.height, 3, mu_s=wl.mu_s); cam = pkg.scene.make_camera(wl.width, wl.height)
m = pkg.scene.to_model(arr, dev); st = pkg.RenderSettings()
proj = pkg.rasterizer.project_gaussians(m.positions, m.rotations, m.scales, m.opacities, m.sh_coeffs, cam, 3, 1.0)
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
cam_abi = cam.to_abi(); n = wl.n
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def fwd(gate):
    L.check(L.lib.cugs_project_forward(n, 16, 3, P(m.positions), P(m.rotations), P(m.scales), P(m.opacities), P(m.sh_coeffs),
            C.byref(cam_abi), 1.0, P(proj.means_2d), P(proj.depths), P(proj.cov_2d_inv), P(proj.radii), P(proj.tiles_touched),
            P(proj.opacities_act), P(proj.rgb), P(proj.packed), P(proj.colour_gate) if gate else None, stream), "fwd")
out = pkg.render(m, cam, st)
g = torch.from_numpy(pkg.scene.make_dl_dcolor(wl.width, wl.height)).to(dev)
rb = pkg.rasterizer.rasterize_backward(g, out.means_2d, out.cov_2d_inv, out.rgb, out.opacities_act, out.tile_ranges, out.gaussian_indices,
                                       out.final_T, out.n_contrib, wl.width, wl.height, st.background, n, packed=out.packed, unpack=False)
dm = torch.empty((n, 2), device=dev)
def bwd(gate):
    pkg.rasterizer.project_backward(None, None, None, None, m.positions, m.rotations, m.scales, m.opacities, m.sh_coeffs, out.radii, cam, 3, 1.0,
                                    grad_accum=rb.grad_accum, colour_gate=out.colour_gate if gate else None, dL_dmeans_2d_out=dm)
def timeit(fn, arg, reps=200):
    for _ in range(20): fn(arg)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn(arg)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1000
for rnd in range(3):
    print("round %d: projection without gate bits %.1f us, with %.1f us | projection backward: gate recomputed from the coefficients %.1f us, from the bits %.1f us"
          % (rnd, timeit(fwd, False), timeit(fwd, True), timeit(bwd, False), timeit(bwd, True)), flush=True)
