"""Development build only: time the sort stage of a workload on both pair routes (depth-ordered emission + two radix
passes / column-ordered emission + one pass).
    CUGS_HIP_LIBRARY=.../libcugs_hip_dev.so python tools/sort_routes.py [config3|config4] [mu_s]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge   # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "config3"
    pkg = ge.load_package()
    fn = pkg._lib.lib.cugsdbg_sort_column_ratio
    fn.restype, fn.argtypes = C.c_int, [C.c_int]
    dev = torch.device("cuda:0")
    wl = pkg.scene.CONFIGS[cfg]
    mu_s = float(sys.argv[2]) if len(sys.argv) > 2 else wl.mu_s
    arrays = pkg.scene.make_gaussians(wl.n, wl.width, wl.height, sh_degree=wl.sh_degree, mu_s=mu_s)
    model = pkg.scene.to_model(arrays, dev)
    cam = pkg.scene.make_camera(wl.width, wl.height)
    R = pkg.rasterizer
    proj = R.project_gaussians(model.positions, model.rotations, model.scales, model.opacities, model.sh_coeffs, cam, wl.sh_degree, 1.0)
    args = (proj.means_2d, proj.depths, proj.radii, proj.tiles_touched, wl.width, wl.height)
    res = {}
    for name, ratio in (("depth-ordered emission + 2 passes", 1 << 20), ("column-ordered emission + 1 pass", 0)):
        fn(ratio)
        for _ in range(5):
            srt = R.sort_gaussians_predicted(*args)
            if isinstance(srt, R.PendingSort):
                srt, _ = srt.finish()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 30
        e0.record()
        for _ in range(reps):
            p = R.sort_gaussians_predicted(*args)
        e1.record()
        torch.cuda.synchronize()
        srt, valid = p.finish()
        res[name] = (e0.elapsed_time(e1) / reps, srt.gaussian_values_sorted.clone())
        print(f"{cfg} mu_s={mu_s}: {name}: {res[name][0]:.4f} ms   pairs {srt.total_pairs} ({srt.total_pairs / wl.n:.1f} per Gaussian)", flush=True)
    a, b = res.values()
    print("same permutation:", bool(torch.equal(a[1], b[1])))


if __name__ == "__main__":
    main()
