"""The blend kernels hand their workgroups out in the order of an optional [tiles] tensor (not in the reference): the tiles
sorted by the length of their lists, longest first - computed by the keyed predicted sort in passing
(cugs_sort_pairs_predicted_keyed_ordered) or from any valid tile_ranges (cugs_tile_order).  ANY permutation of the tiles
is a correct order: the forward's outputs must not depend on it by a bit, the backward's sums only through the order of
their atomic adds."""
import numpy as np
import pytest
import torch

from util import GRAD_NAMES, max_err_over_max, np_

pytestmark = pytest.mark.gpu


def _bucket(length):
    """order_bucket of csrc/sort.hip: 32 x 16 length classes (4 bits below the leading one), empty tiles last."""
    if length == 0:
        return 512
    e = int(length).bit_length() - 1
    m = (length >> (e - 4)) & 15 if e >= 4 else (length << (4 - e)) & 15
    return 511 - (e * 16 + m)


@pytest.mark.parametrize("n,w,h,mu_s,cluster", [
    (60000, 1280, 720, -4.2, (0.8, 0.1)),       # clustered: a few tiles with lists many times the mean
    (3000, 200, 120, -3.0, None),
    (50, 1920, 1080, -4.6, None),               # nearly every tile empty
    (20000, 333, 211, -3.5, (0.5, 0.02)),
])
def test_tile_order_is_a_permutation_longest_first(pkg, dev, monkeypatch, n, w, h, mu_s, cluster):
    R = pkg.rasterizer
    monkeypatch.setattr(R, "TILE_ORDER_MIN_PAIRS", 0)              # render() makes the order for large frames only
    arrays = pkg.scene.make_gaussians(n, w, h, sh_degree=1, seed=n, mu_s=mu_s, cluster=cluster)
    cam = pkg.scene.make_camera(w, h)
    model = pkg.scene.to_model(arrays, dev)
    st = pkg.RenderSettings(active_sh_degree=1)
    outs = [pkg.render(model, cam, st) for _ in range(3)]          # exact sort, then twice the keyed predicted one
    tiles = outs[0].tile_ranges.shape[0]
    for out in outs:
        lens = np_(out.tile_ranges[:, 1] - out.tile_ranges[:, 0]).astype(np.int64)
        tr = np_(out.tile_ranges).astype(np.int64)
        for rec in (np_(out.tile_order), np_(R.tile_order_of(out.tile_ranges, w, h))):
            order = rec[:, 0]
            assert rec.shape == (tiles, 4) and sorted(order.tolist()) == list(range(tiles))
            b = np.array([_bucket(int(x)) for x in lens[order]])
            assert np.all(np.diff(b) >= 0)                         # class by class, longest first
            full = lens[order] > 0                                 # each record carries its tile's range ({t, 0, 0} if empty)
            assert np.array_equal(rec[full, 1], tr[order[full], 0]) and np.array_equal(rec[full, 2], tr[order[full], 1])
            assert not rec[~full, 1:3].any() and not rec[:, 3].any()
    assert torch.equal(outs[1].color, outs[0].color) and torch.equal(outs[2].n_contrib, outs[0].n_contrib)


def test_blends_do_not_depend_on_the_order(pkg, dev, monkeypatch):
    R = pkg.rasterizer
    monkeypatch.setattr(R, "TILE_ORDER_MIN_PAIRS", 0)
    n, w, h = 30000, 640, 360
    arrays = pkg.scene.make_gaussians(n, w, h, sh_degree=2, seed=5, mu_s=-3.8, cluster=(0.7, 0.05))
    cam = pkg.scene.make_camera(w, h)
    model = pkg.scene.to_model(arrays, dev)
    st = pkg.RenderSettings(background=[0.2, 0.1, 0.3], active_sh_degree=2)
    out = pkg.render(model, cam, st)
    tiles = out.tile_ranges.shape[0]
    g = torch.from_numpy(pkg.scene.make_dl_dcolor(w, h)).to(dev)
    gen = torch.Generator().manual_seed(3)
    def records(perm):                                             # {tile, first pair, one past the last pair, 0}
        perm = perm.to(dev).long()
        return torch.cat([perm[:, None].int(), out.tile_ranges[perm], torch.zeros((tiles, 1), dtype=torch.int32, device=dev)],
                         dim=1).contiguous()
    orders = [None, out.tile_order, records(torch.randperm(tiles, generator=gen)), records(torch.arange(tiles - 1, -1, -1))]
    ref_f = ref_b = None
    for order in orders:
        accum = torch.empty((n, pkg._lib.GRAD_STRIDE), dtype=torch.float32, device=dev)
        f = R.rasterize_forward(out.means_2d, out.cov_2d_inv, out.rgb, out.opacities_act, out.tile_ranges, out.gaussian_indices,
                                w, h, st.background, packed=out.packed, zero_buf=accum, tile_order=order)
        assert not bool(accum.any())                               # the accumulator fill rides along in every order
        b = R.rasterize_backward(g, out.means_2d, out.cov_2d_inv, out.rgb, out.opacities_act, out.tile_ranges,
                                 out.gaussian_indices, f.final_T, f.n_contrib, w, h, st.background, n, packed=out.packed,
                                 zeroed_accum=accum, tile_order=order)
        if ref_f is None:
            ref_f, ref_b = f, b
            continue
        assert torch.equal(f.color, ref_f.color) and torch.equal(f.final_T, ref_f.final_T) and torch.equal(f.n_contrib, ref_f.n_contrib)
        for name in ("dL_drgb", "dL_dopacity_act", "dL_dmeans_2d", "dL_dcov_2d_inv"):
            assert max_err_over_max(np_(getattr(b, name)), np_(getattr(ref_b, name))) <= 1e-5, name
    with pytest.raises(Exception):
        R.rasterize_forward(out.means_2d, out.cov_2d_inv, out.rgb, out.opacities_act, out.tile_ranges, out.gaussian_indices,
                            w, h, st.background, packed=out.packed, tile_order=orders[2][:-1].contiguous())   # a record short


def test_render_is_the_same_with_and_without_the_order(pkg, orc, dev, monkeypatch):
    from util import oracle_backward, oracle_forward
    R = pkg.rasterizer
    n, w, h = 20000, 640, 360
    arrays = pkg.scene.make_gaussians(n, w, h, sh_degree=3, seed=77, mu_s=-4.0, cluster=(0.8, 0.1))
    cam = pkg.scene.make_camera(w, h)
    model = pkg.scene.to_model(arrays, dev)
    st = pkg.RenderSettings(active_sh_degree=3)
    g = pkg.scene.make_dl_dcolor(w, h)
    ref = oracle_forward(orc, arrays, cam, degree=3)
    refb = oracle_backward(orc, g, ref, arrays, cam)
    monkeypatch.setattr(R, "TILE_ORDER_MIN_PAIRS", 0)
    for on in (True, False):
        monkeypatch.setattr(R, "TILE_ORDER", on)
        for _ in range(2):
            out = pkg.render(model, cam, st)
        assert (out.tile_order is not None) == on
        assert np.array_equal(np_(out.color).view(np.uint32), ref["color"].view(np.uint32))
        assert np.array_equal(np_(out.n_contrib), ref["n_contrib"])
        grads = pkg.render_backward(torch.from_numpy(g).to(dev), out, model, cam, st)
        for name in GRAD_NAMES:
            assert max_err_over_max(np_(getattr(grads, name)), refb[name]) <= 1e-4, (on, name)


@pytest.mark.parametrize("cluster", [(0.5, 0.02), (0.9, 0.01)])
def test_clustered_view_through_the_window_major_scatter(pkg, orc, dev, cluster):
    """Half (or nine tenths) of the splats on a fiftieth (a hundredth) of a 1080p screen: one of the eighteen windows of the
    binned sort's scatter holds many times the mean, its grid is read window-major with the heaviest window first - and
    the pairs, their order, the ranges and the image are still the oracle's, frame after frame."""
    n, w, h = 40000, 1920, 1080
    arrays = pkg.scene.make_gaussians(n, w, h, sh_degree=0, seed=31, mu_s=-4.2, cluster=cluster)
    cam = pkg.scene.make_camera(w, h)
    model = pkg.scene.to_model(arrays, dev)
    st = pkg.RenderSettings(active_sh_degree=0)
    K = cam.intrinsics
    ref = orc.render(arrays, cam.rotation, cam.translation, K.fx, K.fy, K.cx, K.cy, w, h, active_degree=0,
                     threads=orc.host_threads())
    lens = (ref["tile_ranges"][:, 1] - ref["tile_ranges"][:, 0]).reshape(68, 120)
    windows = [lens[8 * by:8 * by + 8, 64 * gx:64 * gx + 64].sum() for by in range(9) for gx in range(2)]
    assert max(windows) * len(windows) > 2 * sum(windows)          # the rule of k_bin_scatter is met
    for attempt in range(3):                                       # exact sort, then twice the keyed, binned one
        out = pkg.render(model, cam, st)
        assert out.total_pairs == ref["total_pairs"], attempt
        assert np.array_equal(np_(out.gaussian_indices), ref["values"]), attempt
        assert np.array_equal(np_(out.tile_ranges), ref["tile_ranges"]), attempt
        assert np.array_equal(np_(out.n_contrib), ref["n_contrib"]), attempt
        assert np.array_equal(np_(out.color).view(np.uint32), ref["color"].view(np.uint32)), attempt
