"""How sparse is one view's gradient?  Fraction of Gaussians with a non-zero gradient row at config 3, for three of the
eight data-parallel views - what a sparse (index + row) gradient exchange could save on the wire.
Measured on the final tree of round 2: 0.58-0.63 of the rows are non-zero (every Gaussian is in view, mean n_contrib 45)."""
import sys, torch, numpy as np
sys.path.insert(0, ".")
import __graft_entry__ as ge
pkg = ge.load_package()
dev = torch.device("cuda:0")
for name in ("config3",):
    wl = pkg.scene.CONFIGS[name]
    arrays = pkg.scene.make_gaussians(wl.n, wl.width, wl.height, sh_degree=wl.sh_degree, mu_s=wl.mu_s)
    model = pkg.scene.to_model(arrays, dev)
    st = pkg.RenderSettings(active_sh_degree=wl.sh_degree)
    for view in (0, 3, 7):
        cam = pkg.scene.make_camera(wl.width, wl.height, view=view)
        g = torch.from_numpy(pkg.scene.make_dl_dcolor(wl.width, wl.height, seed=pkg.scene.GRAD_SEED + view)).to(dev)
        out = pkg.render(model, cam, st)
        gr = pkg.render_backward(g, out, model, cam, st)
        nzp = (gr.dL_dpositions != 0).any(1)
        nzs = (gr.dL_dsh_coeffs.reshape(wl.n, -1) != 0).any(1)
        nzo = gr.dL_dopacities.reshape(wl.n, -1).ne(0).any(1)
        vis = out.radii > 0
        print(name, "view", view, "visible", float(vis.float().mean()), "nz pos", float(nzp.float().mean()), "nz sh", float(nzs.float().mean()),
              "nz opa", float(nzo.float().mean()), "any", float((nzp | nzs | nzo).float().mean()), "mean n_contrib", float(out.n_contrib.float().mean()))
