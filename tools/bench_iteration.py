#!/usr/bin/env python3
"""One full training iteration as training/trainer.cpp:180-330 runs it, on this package's pieces only:
target from the view cache (N4) -> render -> combined_loss + dL/dcolor (N1) -> render_backward -> FusedAdam ->
densification statistics (N2).  Prints the time per iteration at BASELINE config 3 and, as an end-to-end
sanity check of the gradients, the loss trajectory of a small scene fitted to a target image."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); dev = torch.device("cuda:0")

def iteration(model, cam, settings, opt, ctrl, cache, view, step, fused=True, freeze=False):
    opt.update_lr(step)
    if freeze:                                     # timing runs: every launch as in training, the model stays put
        opt.learning_rates_ = [0.0] * opt.kNumGroups
    target = cache.target(view, cam.width, cam.height)
    # the loss is queued behind the forward blend BEFORE the host waits for the sort's pair count (render_backward
    # does that): the device has the loss kernels to run while the host wakes up and queues the backward
    out = pkg.render(model, cam, settings, defer_count=True)
    loss, dl = pkg.combined_loss_and_grad(out.color, target, 0.2)
    if fused:       # single GPU: the optimizer step rides in the projection backward (same bits, 472 B/Gaussian less)
        grads = pkg.render_backward(dl, out, model, cam, settings, fused_adam=opt)
    else:
        grads = pkg.render_backward(dl, out, model, cam, settings)
        opt.apply_gradients(grads)
        opt.step()
    ctrl.accumulate_gradients(grads.dL_dmeans_2d, out.radii)
    return loss

# ---- 1. time per iteration, 1 M Gaussians / 1080p / SH 3
wl = pkg.scene.CONFIGS["config3"]
model = pkg.scene.to_model(pkg.scene.make_gaussians(wl.n, wl.width, wl.height, 3), dev)
cam = pkg.scene.make_camera(wl.width, wl.height); settings = pkg.RenderSettings(active_sh_degree=3)
cache = pkg.ViewCache(dev)
# The timing runs execute every launch of a training iteration with all learning rates at zero: Adam with eps = 1e-15
# takes lr-sized steps whatever the gradient, so any target - even the scene's own render - sends the splats on a
# random walk (+-15 % pairs within 300 iterations) and the 350 iterations below would not time ONE workload.
cache.add(np.random.default_rng(0).integers(0, 256, (wl.height, wl.width, 3), dtype=np.uint8))
ctrl = pkg.DensificationController(pkg.DensificationConfig(), 6.0)
K = 200             # after 150 untimed iterations: clocks up, the runtime's one-time stall behind (profiles/README.md)
for fused in (False, True):
    model = pkg.scene.to_model(pkg.scene.make_gaussians(wl.n, wl.width, wl.height, 3), dev)
    opt = pkg.FusedAdam(model)
    for s in range(150): iteration(model, cam, settings, opt, ctrl, cache, 0, s, fused, freeze=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in range(K): iteration(model, cam, settings, opt, ctrl, cache, 0, 150 + s, fused, freeze=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    print("full training iteration, 1 M / 1920x1080 / SH 3, %s: %.3f ms = %.0f it/s (target + render + loss + backward + Adam + densify stats)"
          % ("Adam fused into the projection backward" if fused else "backward, then FusedAdam.step", dt * 1e3, 1 / dt))

# ---- 2. the gradients point downhill: fit 4000 Gaussians to the render of a perturbed copy
w, h, n = 320, 240, 4000
arr = pkg.scene.make_gaussians(n, w, h, 3, seed=3, mu_s=-3.2)
cam = pkg.scene.make_camera(w, h); settings = pkg.RenderSettings(active_sh_degree=3)
truth = pkg.scene.to_model(arr, dev)
target = pkg.render(truth, cam, settings).color.clone()
rng = np.random.default_rng(1)
pert = {k: v.copy() for k, v in arr.items()}
pert["positions"] += rng.normal(0, 0.02, pert["positions"].shape).astype(np.float32)
pert["sh_coeffs"] += rng.normal(0, 0.1, pert["sh_coeffs"].shape).astype(np.float32)
pert["opacities"] += rng.normal(0, 0.3, pert["opacities"].shape).astype(np.float32)
model = pkg.scene.to_model(pert, dev); opt = pkg.FusedAdam(model)
losses = []
for s in range(200):
    opt.update_lr(s)
    out = pkg.render(model, cam, settings)
    loss, dl = pkg.combined_loss_and_grad(out.color, target, 0.2)
    pkg.render_backward(dl, out, model, cam, settings, fused_adam=opt)
    if s % 40 == 0 or s == 199: losses.append(float(loss))
print("fit of a perturbed 4000-Gaussian scene to its target, loss at steps 0/40/80/120/160/199: " + " ".join("%.5f" % l for l in losses))
assert losses[-1] < 0.5 * losses[0], "the optimisation did not reduce the loss"
