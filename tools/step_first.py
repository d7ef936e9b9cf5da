"""Host wall time of each of the first steps of the config-3 fwd+bwd loop (no syncs inside), with the number of
device segments torch's caching allocator holds (a change = a hipMalloc, which synchronises the device)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge   # noqa: E402
import bench                   # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    pkg = ge.load_package()
    dev = torch.device("cuda:0")
    wl = pkg.scene.CONFIGS["config3"]
    arrays = pkg.scene.make_gaussians(wl.n, wl.width, wl.height, sh_degree=wl.sh_degree, mu_s=wl.mu_s)
    model = pkg.scene.to_model(arrays, dev)
    cam = pkg.scene.make_camera(wl.width, wl.height)
    settings = pkg.RenderSettings(active_sh_degree=wl.sh_degree)
    g = torch.from_numpy(pkg.scene.make_dl_dcolor(wl.width, wl.height, seed=pkg.scene.GRAD_SEED)).to(dev)
    torch.cuda.synchronize()
    rows = []
    t_prev = time.perf_counter()
    for k in range(steps):
        bench.timed_step(pkg, model, cam, settings, g, None, "compact", False, None)
        t = time.perf_counter()
        st = torch.cuda.memory_stats(dev)
        rows.append((k, (t - t_prev) * 1e3, st["segment.all.current"], st["reserved_bytes.all.current"] >> 20,
                     pkg.rasterizer._last_pairs.get(dev)))
        t_prev = time.perf_counter()
    torch.cuda.synchronize()
    for r in rows:
        print("step %3d host %.3f ms segments %d reserved %d MiB last_pairs %s" % r)


if __name__ == "__main__":
    main()
