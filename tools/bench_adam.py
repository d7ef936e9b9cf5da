"""The stand-alone optimizer step (cugs_fused_adam_groups through FusedAdam.step) on 1 M and 6 M Gaussians at SH degree 3:
ms per step and TB/s of its 28 x 59 bytes per Gaussian.  Learning rates zero (same launch, same bytes, parameters stay put).
    python tools/bench_adam.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge   # noqa: E402


def main():
    pkg = ge.load_package()
    dev = torch.device("cuda:0")
    for n in (1_000_000, 6_000_000):
        arrays = pkg.scene.make_gaussians(n, 1920, 1080, sh_degree=3, seed=5)
        model = pkg.scene.to_model(arrays, dev)
        opt = pkg.FusedAdam(model)
        opt.learning_rates_ = [0.0] * 5
        g = lambda t: torch.randn_like(t) * 1e-3
        grads = pkg.BackwardOutput(dL_dpositions=g(model.positions), dL_drotations=g(model.rotations), dL_dscales=g(model.scales),
                                   dL_dopacities=g(model.opacities), dL_dsh_coeffs=g(model.sh_coeffs),
                                   dL_dmeans_2d=torch.zeros((n, 2), device=dev))
        opt.apply_gradients(grads)
        for _ in range(30):
            opt.step()
        torch.cuda.synchronize()
        reps = 100 if n <= 1_000_000 else 30
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            opt.step()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print("FusedAdam.step, %d Gaussians: %.4f ms = %.2f TB/s of %d MB" % (n, ms, 28 * 59 * n / ms * 1e-9, 28 * 59 * n // 1000000),
              flush=True)
        del model, opt, grads
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
