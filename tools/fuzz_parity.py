"""Randomised parity sweep (not part of the test suite): many small random scenes - sizes, image shapes, SH degrees,
splat scales from sub-pixel to screen-filling, views, backgrounds, scale modifiers - rendered and differentiated on
the GPU and by the oracle.  Integers, order, tile ranges, n_contrib and the image must be bit-equal, gradients within
1e-4 of each tensor's scale.    python tools/fuzz_parity.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge   # noqa: E402
from util import max_err_over_max, np_, oracle_backward, oracle_forward   # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    dense = len(sys.argv) > 3 and sys.argv[3] == "dense"      # larger, denser scenes: the sort's column route, long lists
    pkg, orc = ge.load_package(), ge.load_oracle()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    worst = 0.0
    for k in range(cases):
        n = int(rng.choice([1, 7, 100, 900, 3000, 8000]))
        w = int(rng.choice([1, 15, 16, 17, 64, 130, 333, 640]))
        h = int(rng.choice([1, 9, 16, 33, 96, 217, 360]))
        deg = int(rng.integers(0, 4))
        mu_s = float(rng.choice([-6.0, -4.6, -3.8, -3.0, -2.0, -0.5]))
        if dense:
            n = int(rng.choice([5000, 12000, 25000]))
            w, h = [(640, 360), (1000, 400), (1280, 720), (517, 389)][int(rng.integers(0, 4))]
            mu_s = float(rng.choice([-3.6, -3.2, -2.8]))
        view = int(rng.integers(0, 6))
        bg = tuple(float(x) for x in rng.random(3))
        scale_mod = float(rng.choice([1.0, 1.0, 0.5, 2.0]))
        arrays = pkg.scene.make_gaussians(n, max(w, 8), max(h, 8), sh_degree=deg, seed=int(rng.integers(1 << 30)), mu_s=mu_s)
        if rng.random() < 0.3:
            arrays["opacities"] += np.float32(rng.choice([-4.0, 3.0]))          # faint / opaque (clamp gate, saturation)
        if rng.random() < 0.2:
            arrays["positions"][:, :2] *= np.float32(1.8)                       # many splats off screen (Q7, Q12)
        cam = pkg.scene.make_camera(w, h, view=view)
        model = pkg.scene.to_model(arrays, dev)
        st = pkg.RenderSettings(background=list(bg), active_sh_degree=deg, scale_modifier=scale_mod)
        out = pkg.render(model, cam, st)
        ref = oracle_forward(orc, arrays, cam, bg=bg, degree=deg, scale_mod=scale_mod)
        tag = f"case {k}: n={n} {w}x{h} deg={deg} mu_s={mu_s} view={view} mod={scale_mod} pairs={ref['total_pairs']}"
        assert out.total_pairs == ref["total_pairs"], tag
        assert np.array_equal(np_(out.radii), ref["radii"]), tag
        assert np.array_equal(np_(out.gaussian_indices), ref["values"]), tag
        assert np.array_equal(np_(out.tile_ranges), ref["tile_ranges"]), tag
        assert np.array_equal(np_(out.n_contrib), ref["n_contrib"]), tag
        assert np.array_equal(np_(out.color).view(np.uint32), ref["color"].view(np.uint32)), tag
        g = (pkg.scene.make_dl_dcolor(w, h, seed=k) * np.float32(rng.choice([1.0, 1000.0]))).astype(np.float32)
        grads = pkg.render_backward(torch.from_numpy(g).to(dev), out, model, cam, st)
        refb = oracle_backward(orc, g, ref, arrays, cam, bg=bg, scale_mod=scale_mod)
        for name in ("dL_dpositions", "dL_drotations", "dL_dscales", "dL_dopacities", "dL_dsh_coeffs"):
            got = np_(getattr(grads, name)).reshape(refb[name].shape)
            if not np.any(refb[name]):
                assert not np.any(got), (tag, name)
                continue
            err = max_err_over_max(got, refb[name])
            worst = max(worst, err)
            assert err <= 1e-4, (tag, name, err)
        if k % (2 if dense else 20) == 0:
            print(tag, "ok; worst gradient error so far %.2e" % worst, flush=True)
    print(f"{cases} cases ok, worst gradient error {worst:.2e}")


if __name__ == "__main__":
    main()
