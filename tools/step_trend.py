"""Wall time per step over a long run of the config-3 fwd+bwd step, in blocks of 50 steps (clock ramp / steady state),
with the GPU time of one sampled step per block beside it.
    python tools/step_trend.py [blocks] [idle_seconds_before]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge   # noqa: E402
import bench                   # noqa: E402


def main():
    blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    idle = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    pkg = ge.load_package()
    dev = torch.device("cuda:0")
    wl = pkg.scene.CONFIGS["config3"]
    arrays = pkg.scene.make_gaussians(wl.n, wl.width, wl.height, sh_degree=wl.sh_degree, mu_s=wl.mu_s)
    model = pkg.scene.to_model(arrays, dev)
    cam = pkg.scene.make_camera(wl.width, wl.height)
    settings = pkg.RenderSettings(active_sh_degree=wl.sh_degree)
    g = torch.from_numpy(pkg.scene.make_dl_dcolor(wl.width, wl.height, seed=pkg.scene.GRAD_SEED)).to(dev)
    torch.cuda.synchronize()
    time.sleep(idle)
    t_start = time.perf_counter()
    for b in range(blocks):
        events = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(50):
            bench.timed_step(pkg, model, cam, settings, g, events if k == 25 else None, "compact", False, None)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        ev = events[0]
        print(f"block {b:3d} t={t0 - t_start:7.3f}s wall {(t1 - t0) / 50 * 1e3:.4f} ms/step   sampled step GPU {ev[0].elapsed_time(ev[-1]):.4f} ms"
              f"  stages {[round(ev[i].elapsed_time(ev[i + 1]), 3) for i in range(5)]}", flush=True)


if __name__ == "__main__":
    main()
