#!/usr/bin/env python3
"""Times SURVEY §8(f) N2 at 1 M Gaussians / SH 3 against what it replaces: the reference's libtorch op
sequence (optimizer/densification.cpp, restated in oracle/densify_oracle.py) run on the same GPU."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import __graft_entry__ as ge
from test_densify_oracle import load_densify_oracle
pkg = ge.load_package()
do = load_densify_oracle()
dev = torch.device("cuda:0")
n, C = 1_000_000, 16
g = torch.Generator().manual_seed(0)
rot = torch.randn((n, 4), generator=g); rot = rot / rot.norm(2, 1, True)
t = dict(positions=torch.randn((n, 3), generator=g) * 2, sh_coeffs=torch.randn((n, 3, C), generator=g) * 0.1,
         opacities=torch.randn((n, 1), generator=g) * 3, rotations=rot, scales=torch.randn((n, 3), generator=g) * 1.5 - 2.5)
names = ("positions", "sh_coeffs", "opacities", "rotations", "scales")
grads = (torch.randn((n, 2), generator=g) * 0.0004).to(dev)
radii = torch.randint(0, 40, (n,), generator=g, dtype=torch.int32).to(dev)
noise = torch.randn((2, n, 3), generator=g).to(dev)
cfg = dict(densify_from=500, densify_every=100, grad_threshold=0.0002, opacity_threshold=0.05)

def wall(fn, reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3

with torch.device(dev):        # the oracle's factory calls land on the GPU, as in the reference
    ctrl_ref = do.DensificationController(do.DensificationConfig(**cfg), 8.0)
    ref_acc = lambda: ctrl_ref.accumulate_gradients(grads, radii)
    for _ in range(3): ref_acc()
    t_ref_acc = wall(ref_acc, 20)
ctrl = pkg.DensificationController(pkg.DensificationConfig(**cfg), 8.0)
our_acc = lambda: ctrl.accumulate_gradients(grads, radii)
for _ in range(3): our_acc()
t_our_acc = wall(our_acc, 20)
print("accumulate_gradients, 1 M: libtorch ops %.3f ms   HIP %.3f ms   (%.0fx)" % (t_ref_acc, t_our_acc, t_ref_acc / t_our_acc))

def run_ref():
    with torch.device(dev):
        c = do.DensificationController(do.DensificationConfig(**cfg), 8.0)
        c.grad_accum_, c.grad_count_, c.max_radii_2d_ = ctrl_ref.grad_accum_.clone(), ctrl_ref.grad_count_.clone(), ctrl_ref.max_radii_2d_.clone()
        m = do.Model(*(t[k].to(dev) for k in names))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s = c.densify(m, 600, noise)
        torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3, s
def run_ours(with_opt):
    c = pkg.DensificationController(pkg.DensificationConfig(**cfg), 8.0)
    c.grad_accum_, c.grad_count_, c.max_radii_2d_ = ctrl.grad_accum_.clone(), ctrl.grad_count_.clone(), ctrl.max_radii_2d_.clone()
    m = pkg.GaussianModel(**{k: t[k].to(dev) for k in names})
    opt = pkg.FusedAdam(m) if with_opt else None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s = c.densify(m, 600, noise, optimizer=opt)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3, s
for _ in range(2): run_ref(); run_ours(False); run_ours(True)
r = min(run_ref()[0] for _ in range(5)); sr = run_ref()[1]
o = min(run_ours(False)[0] for _ in range(5)); so = run_ours(False)[1]
om = min(run_ours(True)[0] for _ in range(5))
assert (sr.num_cloned, sr.num_split, sr.num_after) == (so.num_cloned, so.num_split, so.num_after)
moved = so.num_after * (3 + 3 * C + 1 + 4 + 3) * 4 * 2
print("densify, 1 M -> %d (%d cloned, %d split, %d pruned): libtorch ops %.2f ms   HIP %.2f ms (%.1fx; %.0f GB/s of the rows moved)   HIP with Adam moments carried %.2f ms"
      % (so.num_after, so.num_cloned, so.num_split, so.num_pruned, r, o, r / o, moved / o / 1e6, om))
