"""Parity at BASELINE.json's FULL sizes (config 3: 1 M Gaussians, 1920x1080, SH 3), where the oracle cannot
render the whole frame in test time: size-independent properties of every stage, plus exact comparison with
the oracle on a BAND of image rows of the same full-size scene (the oracle projects and sorts all 1 M
Gaussians - seconds - and blends only the band; the GPU's backward is restricted to the band by zeroing
dL_dcolor elsewhere).  Also the extreme-input cases the reference's tests only touch qualitatively."""
import numpy as np
import pytest
import torch

from util import max_err_over_max, np_, oracle_forward

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


def test_fullsize_band_matches_oracle(pkg, orc, dev, full):
    wl, arrays, cam, model, settings, out = full
    K = cam.intrinsics
    r0, r1 = 512, 544                                                 # two tile rows in the middle of the frame
    ref = orc.render(arrays, cam.rotation, cam.translation, K.fx, K.fy, K.cx, K.cy, wl.width, wl.height,
                     active_degree=3, rows=(r0, r1))
    assert np.array_equal(np_(out.radii), ref["radii"])
    assert np.array_equal(np_(out.gaussian_indices), ref["values"])   # all 8.4 M pairs in the same order
    assert np.array_equal(np_(out.tile_ranges), ref["tile_ranges"])
    assert np.array_equal(np_(out.n_contrib)[r0:r1], ref["n_contrib"][r0:r1])
    assert np.array_equal(np_(out.color)[r0:r1].view(np.uint32), ref["color"][r0:r1].view(np.uint32))
    # backward restricted to the band
    g = np.zeros((wl.height, wl.width, 3), np.float32)
    g[r0:r1] = pkg.scene.make_dl_dcolor(wl.width, wl.height)[r0:r1]
    grads = pkg.render_backward(torch.from_numpy(g).to(dev), out, model, cam, settings)
    ref["final_T"][r0:r1] = np_(out.final_T)[r0:r1]
    refb = orc.render_backward(g, ref, arrays, K.fx, K.fy, K.cx, K.cy, wl.width, wl.height, rows=(r0, r1))
    for name in ("dL_dpositions", "dL_drotations", "dL_dscales", "dL_dopacities", "dL_dsh_coeffs"):
        got = np_(getattr(grads, name)).reshape(refb[name].shape)
        assert max_err_over_max(got, refb[name]) <= 1e-4, name
    again = pkg.render_backward(torch.from_numpy(g).to(dev), out, model, cam, settings)   # atomics: order only
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
    ranges, and a band of the image and of the gradients against the oracle."""
    wl = pkg.scene.CONFIGS["config3"]
    arrays = pkg.scene.make_gaussians(wl.n, wl.width, wl.height, sh_degree=wl.sh_degree, mu_s=-3.5)
    cam = pkg.scene.make_camera(wl.width, wl.height)
    model = pkg.scene.to_model(arrays, dev)
    settings = pkg.RenderSettings(active_sh_degree=wl.sh_degree)
    out = pkg.render(model, cam, settings)
    K = cam.intrinsics
    r0, r1 = 512, 528
    ref = orc.render(arrays, cam.rotation, cam.translation, K.fx, K.fy, K.cx, K.cy, wl.width, wl.height,
                     active_degree=3, rows=(r0, r1))
    assert out.total_pairs == ref["total_pairs"] and out.total_pairs >= 13 * wl.n
    assert np.array_equal(np_(out.gaussian_indices), ref["values"])
    assert np.array_equal(np_(out.tile_ranges), ref["tile_ranges"])
    assert np.array_equal(np_(out.n_contrib)[r0:r1], ref["n_contrib"][r0:r1])
    assert np.array_equal(np_(out.color)[r0:r1].view(np.uint32), ref["color"][r0:r1].view(np.uint32))
    g = np.zeros((wl.height, wl.width, 3), np.float32)
    g[r0:r1] = pkg.scene.make_dl_dcolor(wl.width, wl.height)[r0:r1]
    grads = pkg.render_backward(torch.from_numpy(g).to(dev), out, model, cam, settings)
    ref["final_T"][r0:r1] = np_(out.final_T)[r0:r1]
    refb = orc.render_backward(g, ref, arrays, K.fx, K.fy, K.cx, K.cy, wl.width, wl.height, rows=(r0, r1))
    for name in ("dL_dpositions", "dL_drotations", "dL_dscales", "dL_dopacities", "dL_dsh_coeffs"):
        got = np_(getattr(grads, name)).reshape(refb[name].shape)
        assert max_err_over_max(got, refb[name]) <= 1e-4, name
