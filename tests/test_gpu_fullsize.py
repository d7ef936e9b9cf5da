"""Parity at BASELINE.json's FULL sizes (config 3: 1 M Gaussians, 1920x1080, SH 3): size-independent properties of
every stage, and the WHOLE headline frame against the oracle - all 1080 rows of the forward bit for bit and all five
gradients (the oracle's blend runs on the host's cores, OpenMP over rows with a thread-count-independent summation
order: oracle/cugs_oracle.c orc_rasterize_*_rows_mt).  The dense variant is compared on a quarter-frame band.  Also the
extreme-input cases the reference's tests only touch qualitatively."""
import numpy as np
import pytest
import torch

from util import GRAD_NAMES, blend_stage_report, load_parity, max_err_over_max, np_, oracle_forward

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full(pkg, dev):
    wl = pkg.scene.CONFIGS["config3"]
    arrays = pkg.scene.make_gaussians(wl.n, wl.width, wl.height, sh_degree=wl.sh_degree)
    cam = pkg.scene.make_camera(wl.width, wl.height)
    model = pkg.scene.to_model(arrays, dev)
    settings = pkg.RenderSettings(active_sh_degree=wl.sh_degree)
    out = pkg.render(model, cam, settings)
    return wl, arrays, cam, model, settings, out


def test_fullsize_sort_properties(pkg, dev, full):
    wl, arrays, cam, model, settings, out = full
    srt = pkg.sort_gaussians(out.means_2d, out.depths, out.radii,
                             pkg.project_gaussians(model.positions, model.rotations, model.scales, model.opacities,
                                                   model.sh_coeffs, cam, 3).tiles_touched, wl.width, wl.height)
    keys = srt.gaussian_keys_sorted                                   # int64 holding uint64 < 2^45: order-preserving
    P = srt.total_pairs
    assert P == out.total_pairs and P > 4_000_000
    assert bool((keys[1:] >= keys[:-1]).all())                        # sortedness of the full 64-bit key
    vals = srt.gaussian_values_sorted.long()
    same = keys[1:] == keys[:-1]                                      # stability: ties in ascending index
    assert bool((vals[1:][same] > vals[:-1][same]).all())
    tr = srt.tile_ranges.long()
    touched = tr[:, 1] > tr[:, 0]
    starts, ends = tr[touched, 0], tr[touched, 1]
    assert int(starts[0]) == 0 and int(ends[-1]) == P and bool((starts[1:] == ends[:-1]).all())   # partition of [0,P)
    tile_of = (keys >> 32)
    assert bool((tile_of[starts] == torch.nonzero(touched).squeeze(1)).all())
    # every pair's depth half equals its Gaussian's depth bits; idempotence: sorting again changes nothing
    depth_bits = out.depths.view(torch.int32).long() & 0xFFFFFFFF
    assert bool(((keys & 0xFFFFFFFF) == depth_bits[vals]).all())
    assert torch.equal(srt.gaussian_values_sorted, out.gaussian_indices)
    # checksum of checksums: the multiset of (tile, index) pairs equals the one implied by the tile rectangles
    assert int(vals.sum()) == int((torch.arange(wl.n, device=dev) * pkg.project_gaussians(
        model.positions, model.rotations, model.scales, model.opacities, model.sh_coeffs, cam, 3).tiles_touched.long()).sum())


def test_fullsize_forward_properties_and_determinism(pkg, dev, full):
    wl, arrays, cam, model, settings, out = full
    assert bool(torch.isfinite(out.color).all())
    assert float(out.final_T.min()) >= 0.0 and float(out.final_T.max()) <= 1.0
    tr = out.tile_ranges.long()
    ntx = (wl.width + 15) // 16
    lens = (tr[:, 1] - tr[:, 0]).view(-1, ntx)
    per_pixel_len = lens.repeat_interleave(16, 0).repeat_interleave(16, 1)[:wl.height, :wl.width]
    assert bool((out.n_contrib.long() <= per_pixel_len).all())
    again = pkg.render(model, cam, settings)                          # the forward is deterministic bit for bit
    assert torch.equal(again.color, out.color) and torch.equal(again.n_contrib, out.n_contrib)
    assert torch.equal(again.gaussian_indices, out.gaussian_indices)


def test_fullsize_frame_matches_oracle(pkg, orc, dev, full):
    """The headline frame, whole: forward bit-exact on all 1080 rows, all five gradients within 1e-4 of their scale,
    SURVEY 8d's element-wise figure printed beside it, and every accumulator element over 1e-4 element-wise shown
    to be a cancelling sum (|diff| within the fp32 bound of the magnitudes of its own terms).
    Reference kernels: rasterizer/forward.cu:48-174, backward.cu:31-233; bar of tests/test_backward.cpp:266-336."""
    wl, arrays, cam, model, settings, out = full
    K = cam.intrinsics
    par = load_parity()
    th = orc.host_threads()
    ref = orc.render(arrays, cam.rotation, cam.translation, K.fx, K.fy, K.cx, K.cy, wl.width, wl.height,
                     active_degree=3, threads=th)
    assert np.array_equal(np_(out.radii), ref["radii"])
    assert np.array_equal(np_(out.gaussian_indices), ref["values"])   # all 8.4 M pairs in the same order
    assert np.array_equal(np_(out.tile_ranges), ref["tile_ranges"])
    assert np.array_equal(np_(out.n_contrib), ref["n_contrib"])       # every blend decision of 2 M pixels
    assert np.array_equal(np_(out.color).view(np.uint32), ref["color"].view(np.uint32))
    assert np.array_equal(np_(out.final_T).view(np.uint32), ref["final_T"].view(np.uint32))
    g = pkg.scene.make_dl_dcolor(wl.width, wl.height)
    grads = pkg.render_backward(torch.from_numpy(g).to(dev), out, model, cam, settings)
    refb = orc.render_backward(g, ref, arrays, K.fx, K.fy, K.cx, K.cy, wl.width, wl.height, threads=th)
    rep = par.gradient_report({k: np_(getattr(grads, k)) for k in GRAD_NAMES}, refb, GRAD_NAMES)
    print(par.format_report(rep, "config 3, full frame: gradients vs oracle (SURVEY 8d element-wise | of the tensor's scale)"))
    for name, v in rep["per_tensor"].items():
        assert v["over_scale"] <= 1e-4, (name, v)
    out2 = pkg.render(model, cam, settings)                            # a fresh accumulator for the stage-level pass
    stage = blend_stage_report(pkg, orc, dev, out2, ref, g, (0.0, 0.0, 0.0), wl.n, wl.width, wl.height, threads=th)
    print(par.format_report(stage, "config 3, full frame: blend-backward accumulators with the magnitude of their terms"))
    for name, v in stage.items():
        assert v["over_scale"] <= 1e-4, (name, v)
        assert v["over_bar_beyond_term_bound"] == 0, (name, v)
    again = pkg.render_backward(torch.from_numpy(g).to(dev), out2, model, cam, settings)   # atomics: order only
    assert max_err_over_max(np_(again.dL_dsh_coeffs), np_(grads.dL_dsh_coeffs)) <= 1e-5


@pytest.mark.parametrize("w,h,mu_s,n", [(1, 1, -3.0, 50), (17, 3, -1.0, 200), (640, 360, 0.5, 300), (96, 64, -9.0, 500)])
def test_extreme_inputs_match_oracle(pkg, orc, dev, w, h, mu_s, n):
    """1-pixel images, slivers, screen-filling splats that hit the radius clamp min(r, max(W,H))
    (projection.cu:165-167; thousands of tiles per Gaussian), sub-pixel splats (the 0.3 low-pass dominates)."""
    arrays = pkg.scene.make_gaussians(n, max(w, 8), max(h, 8), sh_degree=2, seed=w * 31 + h, mu_s=mu_s)
    cam = pkg.scene.make_camera(w, h, view=1)
    model = pkg.scene.to_model(arrays, dev)
    st = pkg.RenderSettings(background=[0.3, 0.2, 0.1], active_sh_degree=2)
    out = pkg.render(model, cam, st)
    ref = oracle_forward(orc, arrays, cam, bg=(0.3, 0.2, 0.1), degree=2)
    assert np.array_equal(np_(out.radii), ref["radii"])
    if mu_s > 0:
        assert ref["radii"].max() == max(w, h)                        # the clamp is exercised
    assert out.total_pairs == ref["total_pairs"]
    assert np.array_equal(np_(out.gaussian_indices), ref["values"])
    assert np.array_equal(np_(out.tile_ranges), ref["tile_ranges"])
    assert np.array_equal(np_(out.n_contrib), ref["n_contrib"])
    assert np.array_equal(np_(out.color).view(np.uint32), ref["color"].view(np.uint32))


def test_fullsize_dense_view_matches_oracle(pkg, orc, dev):
    """The dense variant of config 3 (mu_s = -3.5: 45 M pairs, 45 per Gaussian - the column-ordered pair emission with
    one radix pass, workgroups that stage their items in several batches): all pairs in the oracle's order, tile
    ranges, and a quarter-frame band (272 rows) of the image and of the gradients against the oracle."""
    wl = pkg.scene.CONFIGS["config3"]
    arrays = pkg.scene.make_gaussians(wl.n, wl.width, wl.height, sh_degree=wl.sh_degree, mu_s=-3.5)
    cam = pkg.scene.make_camera(wl.width, wl.height)
    model = pkg.scene.to_model(arrays, dev)
    settings = pkg.RenderSettings(active_sh_degree=wl.sh_degree)
    out = pkg.render(model, cam, settings)
    K = cam.intrinsics
    r0, r1 = 400, 672
    th = orc.host_threads()
    ref = orc.render(arrays, cam.rotation, cam.translation, K.fx, K.fy, K.cx, K.cy, wl.width, wl.height,
                     active_degree=3, rows=(r0, r1), threads=th)
    assert out.total_pairs == ref["total_pairs"] and out.total_pairs >= 13 * wl.n
    assert np.array_equal(np_(out.gaussian_indices), ref["values"])
    assert np.array_equal(np_(out.tile_ranges), ref["tile_ranges"])
    assert np.array_equal(np_(out.n_contrib)[r0:r1], ref["n_contrib"][r0:r1])
    assert np.array_equal(np_(out.color)[r0:r1].view(np.uint32), ref["color"][r0:r1].view(np.uint32))
    g = np.zeros((wl.height, wl.width, 3), np.float32)
    g[r0:r1] = pkg.scene.make_dl_dcolor(wl.width, wl.height)[r0:r1]
    grads = pkg.render_backward(torch.from_numpy(g).to(dev), out, model, cam, settings)
    ref["final_T"][r0:r1] = np_(out.final_T)[r0:r1]
    refb = orc.render_backward(g, ref, arrays, K.fx, K.fy, K.cx, K.cy, wl.width, wl.height, rows=(r0, r1), threads=th)
    par = load_parity()
    rep = par.gradient_report({k: np_(getattr(grads, k)) for k in GRAD_NAMES}, refb, GRAD_NAMES)
    print(par.format_report(rep, "dense variant, rows 400..672: gradients vs oracle"))
    for name, v in rep["per_tensor"].items():
        assert v["over_scale"] <= 1e-4, (name, v)
