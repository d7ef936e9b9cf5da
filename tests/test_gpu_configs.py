"""BASELINE.json configs 1 and 4 through the HIP path (configs 2 and 3: test_gpu_parity.py, test_gpu_fullsize.py).

config 4 - 6 M Gaussians, 1600x1063 (6700 tiles), SH 3, fwd + bwd + fused Adam: ~40 M pairs, int32 pair
indexing, 16-bit tile ids, a grown workspace, Adam over 354 M floats (fused_adam.cu:140-164).  The oracle
projects and sorts all 6 M Gaussians (seconds) and blends a quarter-frame band (272 rows, host threads); everything that is an integer is
compared over the WHOLE frame (radii, all pairs in order, tile ranges), the image and the gradients on the band,
the Adam update bit for bit on sampled ranges of every parameter group.
config 1 - 10 k Gaussians, 256x256, SH 0, projection + 2-D covariance (projection.cu:55-189): every output of
project_gaussians against the oracle, bit for bit.
"""
import numpy as np
import pytest
import torch

from util import max_err_over_max, np_

pytestmark = pytest.mark.gpu


def test_config1_projection_matches_oracle(pkg, orc, dev):
    wl = pkg.scene.CONFIGS["config1"]
    arrays = pkg.scene.make_gaussians(wl.n, wl.width, wl.height, sh_degree=wl.sh_degree)
    cam = pkg.scene.make_camera(wl.width, wl.height)
    model = pkg.scene.to_model(arrays, dev)
    K = cam.intrinsics
    proj = pkg.project_gaussians(model.positions, model.rotations, model.scales, model.opacities, model.sh_coeffs,
                                 cam, wl.sh_degree)
    ref = orc.project_forward(arrays["positions"], arrays["rotations"], arrays["scales"], arrays["opacities"],
                              orc.view_matrix(cam.rotation, cam.translation), K.fx, K.fy, K.cx, K.cy, wl.width,
                              wl.height)
    assert int((ref["radii"] > 0).sum()) > 8000                         # the scene is on screen
    assert np.array_equal(np_(proj.radii), ref["radii"])
    assert np.array_equal(np_(proj.tiles_touched), ref["tiles_touched"])
    vis = ref["radii"] > 0
    for name in ("means_2d", "depths", "cov_2d_inv", "opacities_act"):  # floats bit for bit (2-D covariance inverse incl.)
        got, want = np_(getattr(proj, name)), ref[name]
        assert np.array_equal(got[vis].view(np.uint32), want[vis].view(np.uint32)), name
    cc = (-(cam.rotation.T @ cam.translation)).astype(np.float32)
    rgb = orc.clamp_min0(orc.sh_forward(0, arrays["sh_coeffs"], orc.directions(arrays["positions"], cc)))
    assert np.array_equal(np_(proj.rgb).view(np.uint32), rgb.view(np.uint32))
    # and the whole forward of the same scene (not part of config 1, but free at this size)
    out = pkg.render(model, cam, pkg.RenderSettings(active_sh_degree=0))
    full = orc.render(arrays, cam.rotation, cam.translation, K.fx, K.fy, K.cx, K.cy, wl.width, wl.height,
                      active_degree=0)
    assert np.array_equal(np_(out.gaussian_indices), full["values"])
    assert np.array_equal(np_(out.tile_ranges), full["tile_ranges"])
    assert np.array_equal(np_(out.n_contrib), full["n_contrib"])
    assert np.array_equal(np_(out.color).view(np.uint32), full["color"].view(np.uint32))


@pytest.fixture(scope="module")
def c4(pkg, dev):
    wl = pkg.scene.CONFIGS["config4"]
    arrays = pkg.scene.make_gaussians(wl.n, wl.width, wl.height, sh_degree=wl.sh_degree)
    cam = pkg.scene.make_camera(wl.width, wl.height)
    model = pkg.scene.to_model(arrays, dev)
    settings = pkg.RenderSettings(active_sh_degree=wl.sh_degree)
    out = pkg.render(model, cam, settings)
    tiles = pkg.project_gaussians(model.positions, model.rotations, model.scales, model.opacities, model.sh_coeffs,
                                  cam, wl.sh_degree).tiles_touched
    yield wl, arrays, cam, model, settings, out, tiles
    del model, out, tiles
    torch.cuda.empty_cache()


def test_config4_properties(pkg, dev, c4):
    wl, arrays, cam, model, settings, out, tiles = c4
    P = out.total_pairs
    assert 30_000_000 < P < 2 ** 31
    assert (wl.width + 15) // 16 * ((wl.height + 15) // 16) == 6700
    srt = pkg.sort_gaussians(out.means_2d, out.depths, out.radii, tiles, wl.width, wl.height)
    keys = srt.gaussian_keys_sorted
    assert srt.total_pairs == P
    assert bool((keys[1:] >= keys[:-1]).all())                        # sortedness of the full 64-bit key
    vals = srt.gaussian_values_sorted.long()
    same = keys[1:] == keys[:-1]                                      # stability: ties in ascending index
    assert bool((vals[1:][same] > vals[:-1][same]).all())
    del same
    tr = srt.tile_ranges.long()
    touched = tr[:, 1] > tr[:, 0]
    starts, ends = tr[touched, 0], tr[touched, 1]
    assert int(starts[0]) == 0 and int(ends[-1]) == P and bool((starts[1:] == ends[:-1]).all())   # partition of [0,P)
    assert bool(((keys >> 32)[starts] == torch.nonzero(touched).squeeze(1)).all())
    depth_bits = out.depths.view(torch.int32).long() & 0xFFFFFFFF
    assert bool(((keys & 0xFFFFFFFF) == depth_bits[vals]).all())
    assert torch.equal(srt.gaussian_values_sorted, out.gaussian_indices)            # exact path == predicted path
    assert int(vals.sum()) == int((torch.arange(wl.n, device=dev) * tiles.long()).sum())
    assert bool(torch.isfinite(out.color).all())
    assert float(out.final_T.min()) >= 0.0 and float(out.final_T.max()) <= 1.0


def test_config4_band_matches_oracle_and_adam_is_bit_exact(pkg, orc, dev, c4):
    wl, arrays, cam, model, settings, out, tiles = c4
    K = cam.intrinsics
    r0, r1 = 400, 672                                                 # a quarter of the frame (17 tile rows)
    th = orc.host_threads()
    ref = orc.render(arrays, cam.rotation, cam.translation, K.fx, K.fy, K.cx, K.cy, wl.width, wl.height,
                     active_degree=3, rows=(r0, r1), threads=th)
    assert np.array_equal(np_(out.radii), ref["radii"])
    assert np.array_equal(np_(tiles), ref["tiles_touched"])
    assert out.total_pairs == ref["total_pairs"]
    assert np.array_equal(np_(out.gaussian_indices), ref["values"])   # all ~40 M pairs in the oracle's order
    assert np.array_equal(np_(out.tile_ranges), ref["tile_ranges"])
    assert np.array_equal(np_(out.n_contrib)[r0:r1], ref["n_contrib"][r0:r1])
    assert np.array_equal(np_(out.color)[r0:r1].view(np.uint32), ref["color"][r0:r1].view(np.uint32))

    g = np.zeros((wl.height, wl.width, 3), np.float32)                # backward restricted to the band
    g[r0:r1] = pkg.scene.make_dl_dcolor(wl.width, wl.height)[r0:r1]
    grads = pkg.render_backward(torch.from_numpy(g).to(dev), out, model, cam, settings)
    ref["final_T"][r0:r1] = np_(out.final_T)[r0:r1]
    refb = orc.render_backward(g, ref, arrays, K.fx, K.fy, K.cx, K.cy, wl.width, wl.height, rows=(r0, r1),
                               threads=min(th, 8))                    # 432 MB of fp64 sums per thread at 6 M Gaussians
    names = ("dL_dpositions", "dL_dsh_coeffs", "dL_dopacities", "dL_dscales", "dL_drotations")   # ParamGroup order
    for name in names:
        got = np_(getattr(grads, name)).reshape(refb[name].shape)
        assert max_err_over_max(got, refb[name]) <= 1e-4, name
    del ref, refb

    # fused Adam on the 6 M model: two steps, every group, compared bit for bit with the oracle's k_fused_adam
    # on three 60 000-row ranges (start, middle - where the kernel's flat float4 index crosses group boundaries
    # is covered by every group's first and last rows - and end) of each tensor
    params = ("positions", "sh_coeffs", "opacities", "scales", "rotations")
    n = wl.n
    spans = [(0, 60_000), (n // 2 - 30_000, n // 2 + 30_000), (n - 60_000, n)]
    cfg = pkg.AdamConfig()
    opt = pkg.FusedAdam(model, cfg)
    lrs = [opt.get_lr(pkg.ParamGroup(i)) for i in range(5)]
    host = {}
    for pn, gn in zip(params, names):
        p, gr = getattr(model, pn), getattr(grads, gn)
        host[pn] = [(np_(p[a:b]).copy().reshape(-1), np_(gr[a:b]).copy().reshape(-1)) for a, b in spans]
    state = {pn: [(np.zeros_like(pp), np.zeros_like(pp)) for pp, _ in host[pn]] for pn in params}
    opt.apply_gradients(grads)
    for step in (1, 2):
        opt.step()
        bc1, bc2 = orc.adam_bias_correction(cfg.beta1, cfg.beta2, step)
        for gi, pn in enumerate(params):
            for si, (a, b) in enumerate(spans):
                pp, gg = host[pn][si]
                m, v = state[pn][si]
                orc.fused_adam(pp, gg, m, v, lrs[gi], cfg.beta1, cfg.beta2, cfg.eps, bc1, bc2)
                got = np_(getattr(model, pn)[a:b]).reshape(-1)
                assert np.array_equal(got.view(np.uint32), pp.view(np.uint32)), (pn, step, si)
                assert np.array_equal(np_(opt.m_[gi][a:b]).reshape(-1).view(np.uint32), m.view(np.uint32)), (pn, step)
                assert np.array_equal(np_(opt.v_[gi][a:b]).reshape(-1).view(np.uint32), v.view(np.uint32)), (pn, step)
    # the step changed the model (a gradient reached the band's Gaussians) and left it finite
    assert bool(torch.isfinite(model.positions).all()) and bool(torch.isfinite(model.sh_coeffs).all())


def test_config4_fused_adam_training_steps_equal_the_unfused_path(pkg, dev, monkeypatch):
    """BASELINE.json configs[3] as a TRAINING step: render -> render_backward(fused_adam=opt) on the 6 M-Gaussian
    model with the default, non-zero learning rates, two steps (moments carried, bias corrections advancing), against
    render_backward -> apply_gradients -> step on a twin model - every parameter and moment bit for bit
    (fused_adam.cu:44-76, 140-164; trainer.cpp:228-242).  The blend backward's atomics may sum in a different order
    from launch to launch, so the twin's backward is handed the SAME accumulator rows (a copy): everything after the
    accumulator - the chain rule, the SH gradient tile, the optimizer arithmetic - is what the two paths differ in."""
    wl = pkg.scene.CONFIGS["config4"]
    arrays = pkg.scene.make_gaussians(wl.n, wl.width, wl.height, sh_degree=wl.sh_degree)
    cam = pkg.scene.make_camera(wl.width, wl.height)
    settings = pkg.RenderSettings(active_sh_degree=wl.sh_degree)
    g = torch.from_numpy(pkg.scene.make_dl_dcolor(wl.width, wl.height)).to(dev)
    ma, mb = pkg.scene.to_model(arrays, dev), pkg.scene.to_model(arrays, dev)
    oa, ob = pkg.FusedAdam(ma), pkg.FusedAdam(mb)
    assert all(oa.get_lr(pkg.ParamGroup(i)) > 0 for i in range(5))
    R = pkg.rasterizer
    real = R.rasterize_backward
    shared = {}

    def first(*a, **k):                       # path A: the real blend backward; keep a copy of its accumulator rows
        rb = real(*a, **k)
        shared["rows"] = rb.grad_accum.clone()
        return rb

    def second(*a, **k):                      # path B: the same rows instead of a second (re-ordered) atomic sum
        return pkg.RasterizeBackwardOutput(None, None, None, None, shared.pop("rows"))

    names = ("positions", "sh_coeffs", "opacities", "scales", "rotations")
    start = ma.positions.clone()
    for step in (1, 2):
        out_a = pkg.render(ma, cam, settings)
        monkeypatch.setattr(R, "rasterize_backward", first)
        grads = pkg.render_backward(g, out_a, ma, cam, settings)
        oa.apply_gradients(grads)
        oa.step()
        out_b = pkg.render(mb, cam, settings)
        assert torch.equal(out_a.gaussian_indices, out_b.gaussian_indices)       # twins: the same frame
        monkeypatch.setattr(R, "rasterize_backward", second)
        res = pkg.render_backward(g, out_b, mb, cam, settings, fused_adam=ob)
        monkeypatch.setattr(R, "rasterize_backward", real)
        assert res.dL_dpositions is None and res.dL_dsh_coeffs is None
        assert torch.equal(res.dL_dmeans_2d, grads.dL_dmeans_2d)
        del grads, res, out_a, out_b
        for i, k in enumerate(names):
            assert torch.equal(getattr(ma, k), getattr(mb, k)), (step, k)
            assert torch.equal(oa.m_[i], ob.m_[i]) and torch.equal(oa.v_[i], ob.v_[i]), (step, k)
        assert oa.step_count_ == ob.step_count_ == step
    moved = int((ma.positions != start).any(dim=1).sum())
    assert moved > wl.n // 10                                                    # a real step: the visible part of the model moved
    assert bool(torch.isfinite(ma.positions).all()) and bool(torch.isfinite(ma.sh_coeffs).all())
    del ma, mb, oa, ob
    torch.cuda.empty_cache()
