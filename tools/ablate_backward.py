#!/usr/bin/env python3
"""Dev tool: A/B timing of k_raster_backward ablations in ONE process (cdna_hip_programming.md rule 24).
Usage (on the GPU box): python tools/ablate_backward.py [--mu-s -4.6]"""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

ap = argparse.ArgumentParser(); ap.add_argument("--mu-s", type=float, default=-4.6); ap.add_argument("--rounds", type=int, default=8)
ap.add_argument("--variants", default="0,1,2,3")
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
variants = [int(v) for v in a.variants.split(",")]
times = {v: [] for v in variants}
for r in range(a.rounds):
    for v in variants:
        os.environ["CUGS_BWD_ABLATE"] = str(v)
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        times[v].append(e0.elapsed_time(e1))
os.environ.pop("CUGS_BWD_ABLATE", None)
for v in variants:
    print("ABL", v, "median %.3f ms  min %.3f" % (float(np.median(times[v])), min(times[v])))
# forward for reference
ts = []
for r in range(a.rounds):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); R.rasterize_forward(out.means_2d, out.cov_2d_inv, out.rgb, out.opacities_act, out.tile_ranges, out.gaussian_indices, wl.width, wl.height, st.background, packed=out.packed); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
print("forward median %.3f ms" % float(np.median(ts)))

# ---- step counters (ABL 4): needs an accumulator with one extra row -> call the C ABI directly
import ctypes as C
from cugs_amd._lib import lib
acc = torch.zeros((wl.n + 1, 16), dtype=torch.float32, device=dev)
os.environ["CUGS_BWD_ABLATE"] = "4"
bg = (C.c_float * 3)(0, 0, 0)
P = lambda t: C.c_void_p(t.data_ptr())
rc = lib.cugs_rasterize_backward(wl.width, wl.height, bg, P(out.tile_ranges), P(out.gaussian_indices), P(out.means_2d), P(out.cov_2d_inv),
                                 P(out.rgb), P(out.opacities_act), P(out.packed), P(g), P(out.final_T), P(out.n_contrib), wl.n, P(acc),
                                 None, None, None, None, C.c_void_p(torch.cuda.current_stream().cuda_stream))
os.environ.pop("CUGS_BWD_ABLATE", None)
torch.cuda.synchronize()
st = acc[wl.n].cpu().numpy()
tiles = ((wl.width + 15) // 16) * ((wl.height + 15) // 16)
print("rc", rc, "wave-steps %.3g  contributing %.3g (%.0f%%)  lanes/contributing step %.1f  wave-batches walked %.3g of %.3g  records cull-tested %.3g  (pairs x4 waves = %.3g)"
      % (st[0], st[1], 100 * st[1] / max(st[0], 1), st[2] / max(st[1], 1), st[3], st[5], st[4], 4.0 * out.total_pairs))
print("4x4 sub-blocks with a contributing pixel per contributing step: %.2f of 4;  not-finished lanes per contributing step: %.1f of 64"
      % (st[6] / max(st[1], 1), st[7] / max(st[1], 1)))
