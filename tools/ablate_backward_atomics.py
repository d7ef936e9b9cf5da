import os, sys, numpy as np, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import __graft_entry__ as ge
pkg = ge.load_package(); dev = torch.device("cuda:0")
wl = pkg.scene.CONFIGS["config3"]
arr = pkg.scene.make_gaussians(wl.n, wl.width, wl.height, 3); cam = pkg.scene.make_camera(wl.width, wl.height)
model = pkg.scene.to_model(arr, dev); st = pkg.RenderSettings()
out = pkg.render(model, cam, st); g = torch.from_numpy(pkg.scene.make_dl_dcolor(wl.width, wl.height)).to(dev)
R = pkg.rasterizer
def run():
    return R.rasterize_backward(g, out.means_2d, out.cov_2d_inv, out.rgb, out.opacities_act, out.tile_ranges, out.gaussian_indices,
                                out.final_T, out.n_contrib, wl.width, wl.height, st.background, wl.n, packed=out.packed, unpack=False)
for rnd in range(3):
    for flag in ("0", "1"):
        os.environ["CUGS_BWD_NO_ATOMICS"] = flag
        ts = []
        for r in range(12):
            run(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print("no_atomics=%s: backward (memset + kernel) median %.4f ms min %.4f" % (flag, float(np.median(ts)), min(ts)), flush=True)
