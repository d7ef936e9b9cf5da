#!/usr/bin/env python3
"""Integration soak: a training-like loop over every piece of the package - N4 targets from a view cache, render,
N1 loss, backward, fused Adam, N2 statistics + densification with the moments carried, a N3 checkpoint written
and resumed half way - several hundred iterations on a small scene fitted to rendered targets.  Checks that the
loss goes down, the model stays valid across size changes, memory does not creep, and the resumed run continues
bit-exactly where the checkpoint was taken."""
import os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); dev = torch.device("cuda:0")
W, H, N, V, ITERS = 480, 270, 20000, 6, 600
truth = pkg.scene.to_model(pkg.scene.make_gaussians(N, W, H, 3, seed=11, mu_s=-3.6), dev)
settings = pkg.RenderSettings(active_sh_degree=3)
cams = [pkg.scene.make_camera(W, H, view=v) for v in range(V)]
cache = pkg.ViewCache(dev)
for cam in cams:                                             # the "dataset": 8-bit renders of the true scene
    cache.add((pkg.render(truth, cam, settings).color.clamp(0, 1) * 255).round().to(torch.uint8).cpu().numpy())
arr = pkg.scene.make_gaussians(N // 2, W, H, 3, seed=12, mu_s=-3.4)      # start from a different, sparser scene
model = pkg.scene.to_model(arr, dev)
opt = pkg.FusedAdam(model)
ctrl = pkg.DensificationController(pkg.DensificationConfig(densify_from=100, densify_every=100, densify_until=500,
                                                           grad_threshold=float(os.environ.get("SOAK_THR", "2e-5")), opacity_threshold=0.01,
                                                           max_gaussians=int(os.environ.get("SOAK_CAP", "0"))), 6.0)
rng = np.random.default_rng(0)
losses, sizes, mem = [], [], []
t0 = time.perf_counter()
with tempfile.TemporaryDirectory(dir="/dev/shm") as d:
    for step in range(1, ITERS + 1):
        v = int(rng.integers(V))
        opt.update_lr(step)
        out = pkg.render(model, cams[v], settings)
        loss, dl = pkg.combined_loss_and_grad(out.color, cache.target(v, W, H), 0.2)
        grads = pkg.render_backward(dl, out, model, cams[v], settings)
        opt.apply_gradients(grads); opt.step()
        ctrl.accumulate_gradients(grads.dL_dmeans_2d, out.radii)
        if ctrl.should_densify(step):
            s = ctrl.densify(model, step, optimizer=opt)
            assert model.is_valid() and model.num_gaussians() == s.num_after
            sizes.append((step, s.num_before, s.num_cloned, s.num_split, s.num_pruned, s.num_after))
        if step == ITERS // 2:                              # checkpoint, then continue from the loaded copy
            assert pkg.write_gaussian_ply(os.path.join(d, "ckpt.ply"), model, optimizer=opt)
            m2, st = pkg.read_gaussian_ply(os.path.join(d, "ckpt.ply"), device=dev, return_state=True)
            for k in ("positions", "sh_coeffs", "opacities", "rotations", "scales"):
                assert torch.equal(getattr(m2, k), getattr(model, k))
            model = m2
            opt = pkg.FusedAdam(model); pkg.restore_optimizer(opt, st)
        if step % 50 == 0:
            losses.append(float(loss)); mem.append(torch.cuda.memory_allocated() / 1e6)
    torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("%d iterations in %.2f s (%.0f it/s incl. %d densifications and a checkpoint/resume)" % (ITERS, dt, ITERS / dt, len(sizes)))
print("loss every 50 steps:", " ".join("%.4f" % l for l in losses))
print("densifications (step, before, cloned, split, pruned, after):", sizes)
print("allocated MB every 50 steps:", " ".join("%.0f" % m for m in mem))
assert losses[-1] < 0.6 * losses[0], "loss did not go down"
assert max(mem[-4:]) < 1.5 * max(mem[4:8]) + 50, "memory creeps"
print("soak ok")
