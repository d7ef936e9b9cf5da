"""Where does the HOST time of one fwd+bwd step go?  Runs bench.py's timed_step on config 3 and prints, per step:
wall time, host time per section (enqueue only - nothing here waits except `wait`), the wait for the sort's pair
count, and a cProfile top list.  The step is GPU-bound when wall ~ GPU time and `wait` absorbs the slack; it is
host-bound when the sections other than `wait` add up to the wall time.
    python tools/host_profile.py [steps] [config3|config4] [adam]"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge   # noqa: E402
import bench                   # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    pkg = ge.load_package()
    dev = torch.device("cuda:0")
    wl = pkg.scene.CONFIGS[sys.argv[2] if len(sys.argv) > 2 else "config3"]
    use_adam = len(sys.argv) > 3 and sys.argv[3] == "adam"
    arrays = pkg.scene.make_gaussians(wl.n, wl.width, wl.height, sh_degree=wl.sh_degree, mu_s=wl.mu_s)
    model = pkg.scene.to_model(arrays, dev)
    cam = pkg.scene.make_camera(wl.width, wl.height)
    settings = pkg.RenderSettings(active_sh_degree=wl.sh_degree)
    g = torch.from_numpy(pkg.scene.make_dl_dcolor(wl.width, wl.height, seed=pkg.scene.GRAD_SEED)).to(dev)
    R = pkg.rasterizer
    opt = pkg.FusedAdam(model) if use_adam else None
    acc = {}
    orig = {}

    def wrap(obj, name, key):
        fn = getattr(obj, name)
        orig[(obj, name)] = fn

        def inner(*a, **k):
            t = time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                acc[key] = acc.get(key, 0.0) + time.perf_counter() - t
        setattr(obj, name, inner)

    def run(k):
        for _ in range(k):
            bench.timed_step(pkg, model, cam, settings, g, None, "compact", False, opt)

    run(30)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    plain = (time.perf_counter() - t0) / steps * 1e3
    wrap(R, "project_gaussians", "project_forward")
    wrap(R, "sort_gaussians_predicted", "sort")
    wrap(R, "rasterize_forward", "raster_forward")
    wrap(R.PendingSort, "finish", "wait")
    wrap(R, "rasterize_backward", "raster_backward")
    wrap(R, "project_backward", "project_backward")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e3
    print(f"wall {plain:.4f} ms/step (instrumented {wall:.4f})")
    tot = 0.0
    for k, v in acc.items():
        print(f"  host {k:18s} {v / steps * 1e3:.4f} ms")
        tot += v
    print(f"  host sum             {tot / steps * 1e3:.4f} ms   (outside these calls {wall - tot / steps * 1e3:.4f} ms)")
    for (obj, name), fn in orig.items():
        setattr(obj, name, fn)
    pr = cProfile.Profile()
    pr.enable()
    run(steps)
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr, stream=sys.stdout)
    st.sort_stats("tottime").print_stats(18)
    ms = torch.cuda.memory_stats(dev)
    print("allocator: segments", ms["segment.all.current"], "reserved MiB", ms["reserved_bytes.all.current"] >> 20,
          "device mallocs", ms["segment.all.allocated"], "frees", ms["segment.all.freed"], "retries", ms["num_alloc_retries"])


if __name__ == "__main__":
    main()
