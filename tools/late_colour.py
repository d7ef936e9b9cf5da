"""Development experiment: the projection's colour half (SH evaluation, HBM-bound, 73 us) launched on a side stream when
k_bin_scatter (instruction-bound, 79 us) STARTS - not at projection time, where it sits on top of the depth passes.
Needs the development library (cugsdbg_sort_mark_event: an event recorded right before k_bin_scatter).
    CUGS_HIP_LIBRARY=.../libcugs_hip_dev.so python tools/late_colour.py"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

pkg = ge.load_package()
R = pkg.rasterizer
lib = R.lib
dbg = C.CDLL(pkg.LIB_PATH)
dbg.cugsdbg_sort_mark_event.argtypes = [C.c_void_p]
dev = torch.device("cuda", 0)
wl = pkg.scene.CONFIGS["config3"]
arrays = pkg.scene.make_gaussians(wl.n, wl.width, wl.height, sh_degree=wl.sh_degree)
cam = pkg.scene.make_camera(wl.width, wl.height)
model = pkg.scene.to_model(arrays, dev)
settings = pkg.RenderSettings(active_sh_degree=wl.sh_degree)
g = torch.from_numpy(pkg.scene.make_dl_dcolor(wl.width, wl.height)).to(dev)
n, deg = wl.n, wl.sh_degree
ref = pkg.render(model, cam, settings)            # the pair prediction; the reference image
torch.cuda.synchronize()


def frame_late(mark, grid_cap=None):
    f = dict(dtype=torch.float32, device=dev)
    i = dict(dtype=torch.int32, device=dev)
    means_2d, depths, cov = torch.empty((n, 2), **f), torch.empty((n,), **f), torch.empty((n, 3), **f)
    radii, tiles = torch.empty((n,), **i), torch.empty((n,), **i)
    opa_act, rgb = torch.empty((n,), **f), torch.empty((n, 3), **f)
    packed = torch.empty((n, pkg._lib.PACKED_STRIDE), **f)
    gate = torch.empty((n,), dtype=torch.uint8, device=dev)
    camabi = cam.to_abi()
    ws = R._workspace(dev, lib.cugs_sort_workspace_bytes(n), "n")
    main, side = torch.cuda.current_stream(dev), R._side_stream(dev)
    R.check(lib.cugs_project_forward_geometry(n, R._ptr(model.positions), R._ptr(model.rotations), R._ptr(model.scales),
                                              R._ptr(model.opacities), C.byref(camabi), 1.0, R._ptr(means_2d), R._ptr(depths),
                                              R._ptr(cov), R._ptr(radii), R._ptr(tiles), R._ptr(opa_act), R._ptr(packed),
                                              R._ptr(ws), ws.numel(), R._stream(dev)), "geometry")
    pend = R.sort_gaussians_predicted(means_2d, depths, radii, tiles, wl.width, wl.height, keyed_workspace=ws)
    side.wait_event(mark)                         # recorded by the sort right before k_bin_scatter
    R.check(lib.cugs_project_forward_colour(n, int(model.sh_coeffs.shape[2]), deg, R._ptr(model.positions),
                                            R._ptr(model.sh_coeffs), C.byref(camabi), R._ptr(rgb), R._ptr(packed), R._ptr(gate),
                                            C.c_void_p(side.cuda_stream)), "colour")
    done = torch.cuda.Event()
    done.record(side)
    main.wait_event(done)
    accum = torch.empty((n, pkg._lib.GRAD_STRIDE), **f)
    fwd = R.rasterize_forward(means_2d, cov, rgb, opa_act, pend.tile_ranges, pend.gaussian_values_sorted, wl.width, wl.height,
                              settings.background, packed=packed, zero_buf=accum)
    srt, valid = pend.finish()
    assert valid
    out = R.RenderOutput(fwd.color, fwd.final_T, fwd.n_contrib, means_2d, depths, cov, radii, rgb, opa_act,
                         srt.gaussian_values_sorted, srt.tile_ranges, packed=packed, colour_gate=gate,
                         total_pairs=srt.total_pairs, zeroed_accum=accum)
    grads = pkg.render_backward(g, out, model, cam, settings)
    return out, grads


def frame_plain():
    out = pkg.render(model, cam, settings)
    return out, pkg.render_backward(g, out, model, cam, settings)


mark = torch.cuda.Event()
mark.record()
torch.cuda.synchronize()
dbg.cugsdbg_sort_mark_event(C.c_void_p(mark.cuda_event))
out, grads = frame_late(mark)
torch.cuda.synchronize()
assert torch.equal(out.color, ref.color) and torch.equal(out.gaussian_indices, ref.gaussian_indices), "late-colour frame differs"
print("late-colour frame: image and pairs bit-identical to render()", flush=True)
for rnd in range(3):
    for name, fn in (("plain", frame_plain), ("late ", lambda: frame_late(mark))):
        for _ in range(30):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"{name}: {e0.elapsed_time(e1) / 200:.4f} ms/step (host {1e3 * (time.perf_counter() - t0) / 200:.4f})", flush=True)
dbg.cugsdbg_sort_mark_event(C.c_void_p(0))
