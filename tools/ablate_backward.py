#!/usr/bin/env python3
"""Dev tool: timing and step counters of k_raster_backward (the counters live in the development build,
libcugs_hip_dev.so, behind cugsdbg_backward_stats).
Usage (on the GPU box): python tools/ablate_backward.py [--mu-s -4.6]"""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

ap = argparse.ArgumentParser(); ap.add_argument("--mu-s", type=float, default=-4.6); ap.add_argument("--rounds", type=int, default=8)
a = ap.parse_args()
pkg = ge.load_package(); dev = torch.device("cuda:0")
wl = pkg.scene.CONFIGS["config3"]
arr = pkg.scene.make_gaussians(wl.n, wl.width, wl.height, 3, mu_s=a.mu_s); cam = pkg.scene.make_camera(wl.width, wl.height)
model = pkg.scene.to_model(arr, dev); st = pkg.RenderSettings()
out = pkg.render(model, cam, st); g = torch.from_numpy(pkg.scene.make_dl_dcolor(wl.width, wl.height)).to(dev)
nc = out.n_contrib.float()
print("pairs", out.total_pairs, "n_contrib mean %.1f max %d  saturated %.3f" % (nc.mean().item(), int(nc.max().item()), (out.final_T < 1/255).float().mean().item()))
R = pkg.rasterizer
def run():
    return R.rasterize_backward(g, out.means_2d, out.cov_2d_inv, out.rgb, out.opacities_act, out.tile_ranges, out.gaussian_indices,
                                out.final_T, out.n_contrib, wl.width, wl.height, st.background, wl.n, packed=out.packed, unpack=False)
ts = []
for r in range(a.rounds):
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
print("backward median %.3f ms  min %.3f" % (float(np.median(ts)), min(ts)))
# forward for reference
ts = []
for r in range(a.rounds):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); R.rasterize_forward(out.means_2d, out.cov_2d_inv, out.rgb, out.opacities_act, out.tile_ranges, out.gaussian_indices, wl.width, wl.height, st.background, packed=out.packed); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
print("forward median %.3f ms" % float(np.median(ts)))

# ---- step counters: development build, accumulator with one extra row -> call its C ABI directly
import ctypes as C
dev_lib = C.CDLL(os.path.join(os.path.dirname(pkg.LIB_PATH), "libcugs_hip_dev.so"))
acc = torch.zeros((wl.n + 1, 16), dtype=torch.float32, device=dev)
dev_lib.cugsdbg_backward_stats(1)
bg = (C.c_float * 3)(0, 0, 0)
P = lambda t: C.c_void_p(t.data_ptr())
dev_lib.cugs_rasterize_backward.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_float)] + [C.c_void_p] * 10 + [C.c_int64] + [C.c_void_p] * 6
rc = dev_lib.cugs_rasterize_backward(wl.width, wl.height, bg, P(out.tile_ranges), P(out.gaussian_indices), P(out.means_2d), P(out.cov_2d_inv),
                                     P(out.rgb), P(out.opacities_act), P(out.packed), P(g), P(out.final_T), P(out.n_contrib), wl.n, P(acc),
                                     None, None, None, None, C.c_void_p(torch.cuda.current_stream().cuda_stream))
dev_lib.cugsdbg_backward_stats(0)
torch.cuda.synchronize()
st = acc[wl.n].cpu().numpy()
print("rc", rc, "wave-steps %.4g  contributing %.4g (%.0f%%)  lanes/contributing step %.1f  wave-batches walked %.4g of %.4g  records cull-tested %.4g  (pairs x4 waves = %.4g)"
      % (st[0], st[1], 100 * st[1] / max(st[0], 1), st[2] / max(st[1], 1), st[3], st[5], st[4], 4.0 * out.total_pairs))
print("not-finished lanes per contributing step: %.1f of 64" % (st[7] / max(st[1], 1)))
