"""Development build only: where k_col_emit (column-ordered pair emission) spends its time, per phase, from the
100 MHz tick sums the kernel accumulates.   CUGS_HIP_LIBRARY=.../libcugs_hip_dev.so python tools/emit_profile.py [mu_s]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge   # noqa: E402


def main():
    mu_s = float(sys.argv[1]) if len(sys.argv) > 1 else None
    pkg = ge.load_package()
    lib = pkg._lib.lib
    fn = lib.cugsdbg_emit_profile
    fn.restype, fn.argtypes = C.c_int, [C.POINTER(C.c_ulonglong)]
    force = lib.cugsdbg_sort_column_ratio
    force.restype, force.argtypes = C.c_int, [C.c_int]
    force(0)                                    # column-ordered emission whatever the pairs-per-Gaussian ratio
    dev = torch.device("cuda:0")
    wl = pkg.scene.CONFIGS["config3"]
    arrays = pkg.scene.make_gaussians(wl.n, wl.width, wl.height, sh_degree=wl.sh_degree, mu_s=mu_s if mu_s is not None else wl.mu_s)
    model = pkg.scene.to_model(arrays, dev)
    cam = pkg.scene.make_camera(wl.width, wl.height)
    settings = pkg.RenderSettings(active_sh_degree=wl.sh_degree)
    for _ in range(3):
        out = pkg.render(model, cam, settings, for_backward=False)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 16)()
    fn(buf)
    reps = 10
    for _ in range(reps):
        out = pkg.render(model, cam, settings, for_backward=False)
    torch.cuda.synchronize()
    fn(buf)
    nblk = (wl.n + 1023) // 1024
    names = ["load+scans", "clear+cover", "count+scan", "records", "slots+deltas", "stream (thread 0's wave)", "final barrier wait"]
    tot = sum(buf[:7])
    print(f"pairs {out.total_pairs}, {nblk} workgroups; per workgroup, microseconds (100 MHz ticks):")
    for k, nm in enumerate(names):
        print(f"  {nm:28s} {buf[k] / reps / nblk / 100.0:8.2f} us")
    print(f"  {'sum':28s} {tot / reps / nblk / 100.0:8.2f} us")


if __name__ == "__main__":
    main()
