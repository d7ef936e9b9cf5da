"""Development: time the keyed predicted sort (projection excluded) with parts of k_bin_scatter switched off.
CUGS_BIN_ABLATE bits: 1 no global store, 2 no advance phase, 4 no cover phase, 8 no batches at all.  With any bit set
the result is WRONG, so nothing downstream (no blend) is run here.
    CUGS_HIP_LIBRARY=.../libcugs_hip_dev.so python tools/ablate_bin.py [mu_s]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

pkg = ge.load_package()
R = pkg.rasterizer
dev = torch.device("cuda", 0)
wl = pkg.scene.CONFIGS[sys.argv[2] if len(sys.argv) > 2 else "config3"]
mu = float(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] != "-" else wl.mu_s
arrays = pkg.scene.make_gaussians(wl.n, wl.width, wl.height, sh_degree=wl.sh_degree, mu_s=mu)
cam = pkg.scene.make_camera(wl.width, wl.height)
model = pkg.scene.to_model(arrays, dev)
margs = (model.positions, model.rotations, model.scales, model.opacities, model.sh_coeffs, cam, wl.sh_degree)
out = pkg.render(model, cam, pkg.RenderSettings(active_sh_degree=wl.sh_degree))      # sets the pair prediction
print("pairs", out.total_pairs, flush=True)
import ctypes as C
_lib = C.CDLL(pkg.LIB_PATH)
for grp in ():              # Gaussians per table row (CUGS_BIN_GROUP; the workspace is sized for 4096: larger only)
    os.environ["CUGS_BIN_GROUP"] = str(grp)
    times = []
    for it in range(8):
        keyed = R.project_gaussians(*margs, key_sort=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        pend = R.sort_gaussians_predicted(keyed.means_2d, keyed.depths, keyed.radii, keyed.tiles_touched, wl.width, wl.height,
                                          keyed_workspace=keyed.sort_workspace)
        e1.record()
        pend.finish()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) * 1000.0)
    times.sort()
    print(f"group={grp}: sort {times[len(times) // 2]:.1f} us (min {times[0]:.1f})", flush=True)
os.environ.pop("CUGS_BIN_GROUP", None)
for ab in (0, 1, 8, 0, -1):                  # -1: the radix route (direct binning off)
    _lib.cugsdbg_sort_direct_route(0 if ab < 0 else 1)
    os.environ["CUGS_BIN_ABLATE"] = str(max(ab, 0))
    times = []
    for it in range(8):
        keyed = R.project_gaussians(*margs, key_sort=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        pend = R.sort_gaussians_predicted(keyed.means_2d, keyed.depths, keyed.radii, keyed.tiles_touched, wl.width, wl.height,
                                          keyed_workspace=keyed.sort_workspace)
        e1.record()
        pend.finish()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) * 1000.0)
    times.sort()
    print(f"ablate={ab}: sort {times[len(times) // 2]:.1f} us (min {times[0]:.1f})", flush=True)
os.environ["CUGS_BIN_ABLATE"] = "0"
