#!/usr/bin/env python3
"""Times the fused combined_loss + dL/dcolor (csrc/loss.hip) at 1920x1080 against what it replaces: the
reference's libtorch op sequence (training/loss.cpp) + autograd + the two clones (trainer.cpp:214-217),
run on the same GPU (grouped conv2d -> MIOpen)."""
import importlib.util, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
spec = importlib.util.spec_from_file_location("lo", os.path.join(ge.ROOT, "oracle", "loss_oracle.py")); lo = importlib.util.module_from_spec(spec); spec.loader.exec_module(lo)
dev = torch.device("cuda:0"); h, w = 1080, 1920
g = torch.Generator().manual_seed(0)
t = torch.rand((h, w, 3), generator=g).to(dev); r = (t + 0.1 * torch.randn((h, w, 3), generator=g).to(dev)).clamp(0, 1)
kern = lo.gaussian_kernel(11).to(dev)
lo.gaussian_kernel = lambda ws: kern           # the reference caches the kernel per device (loss.cpp:47-60)

def ref_step():
    rendered = r.clone().detach().requires_grad_(True)
    loss = lo.combined_loss(rendered, t, 0.2)
    loss.backward()
    return loss, rendered.grad.clone()

def ours():
    return pkg.combined_loss_and_grad(r, t, 0.2)

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

l_ref, g_ref = ref_step(); l_our, g_our = ours()
print("loss ref %.7f ours %.7f  grad max-err/max %.2e" % (float(l_ref), float(l_our), float((g_ref - g_our).abs().max() / g_ref.abs().max())))
t_ref, t_our = timeit(ref_step), timeit(ours)
bytes_alg = h * w * 3 * 4 * 11      # read x,y twice (2+2), write+read 3 maps (6), write grad (1)
print("libtorch ops + autograd on GPU: %.3f ms   fused HIP: %.3f ms   (%.1fx);  fused = %.0f GB/s of 11 x 12 B/pixel" % (t_ref, t_our, t_ref / t_our, bytes_alg / t_our / 1e6))
