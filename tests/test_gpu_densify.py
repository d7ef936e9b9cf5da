"""SURVEY §8(f) N2 on the GPU: statistics and clone / split / prune (csrc/densify.hip through the C ABI and
the DensificationController mirror) against the oracle = the reference's libtorch op sequence on CPU
(oracle/densify_oracle.py), same split noise on both sides.  Copied rows must be bit-identical; the
children's positions involve exp() and are held to 1e-6 of the scene scale."""
import numpy as np
import pytest
import torch

from test_densify_oracle import load_densify_oracle
from util import np_

pytestmark = pytest.mark.gpu
NAMES = ("positions", "sh_coeffs", "opacities", "rotations", "scales")


@pytest.fixture(scope="module")
def do():
    return load_densify_oracle()


def _models(pkg, do, dev, n, seed, coeffs=4):
    g = torch.Generator().manual_seed(seed)
    rot = torch.randn((n, 4), generator=g)
    rot = rot / rot.norm(2, 1, True).clamp_min(1e-8)
    t = dict(positions=torch.randn((n, 3), generator=g) * 2.0, sh_coeffs=torch.randn((n, 3, coeffs), generator=g) * 0.1,
             opacities=torch.randn((n, 1), generator=g) * 3.0, rotations=rot,
             scales=torch.randn((n, 3), generator=g) * 1.5 - 2.5)
    ref = do.Model(*(t[k].clone() for k in NAMES))
    ours = pkg.GaussianModel(**{k: t[k].clone().to(dev) for k in NAMES})
    return ref, ours


def _accumulate_both(ctrl_ref, ctrl, dev, n, views, seed):
    g = torch.Generator().manual_seed(seed)
    for _ in range(views):
        grads = torch.randn((n, 2), generator=g) * 0.0004
        radii = torch.randint(-1, 40, (n,), generator=g, dtype=torch.int32)
        radii[torch.rand(n, generator=g) < 0.3] = 0
        ctrl_ref.accumulate_gradients(grads, radii)
        ctrl.accumulate_gradients(grads.to(dev), radii.to(dev))


def _compare_models(ref, ours, n_children_start=None):
    assert ours.num_gaussians() == ref.num_gaussians() and ours.is_valid()
    for k in NAMES:
        a, b = np_(getattr(ours, k)), getattr(ref, k).numpy()
        if k == "positions" and n_children_start is not None:
            assert np.array_equal(a[:n_children_start], b[:n_children_start])
            assert np.max(np.abs(a[n_children_start:] - b[n_children_start:]), initial=0.0) <= 1e-6 * max(1.0, np.abs(b).max())
        elif k == "scales":
            assert np.allclose(a, b, rtol=0, atol=1e-6)
        else:
            assert np.array_equal(a, b), k


@pytest.mark.parametrize("n,step,kw", [
    (5000, 600, dict()),                                           # before the first opacity reset: no size pruning
    (5000, 3100, dict()),                                          # size pruning on (step > opacity_reset_every)
    (3001, 3100, dict(max_screen_size=0)),                         # screen-size test disabled, ragged n
    (257, 700, dict(grad_threshold=1000.0)),                       # nothing cloned or split: prune only
])
def test_statistics_and_densify_parity(pkg, do, dev, n, step, kw):
    cfg_kw = dict(densify_from=500, densify_every=100, grad_threshold=0.0002, opacity_threshold=0.05, percent_dense=0.01)
    cfg_kw.update(kw)
    ctrl_ref = do.DensificationController(do.DensificationConfig(**cfg_kw), 8.0)
    ctrl = pkg.DensificationController(pkg.DensificationConfig(**cfg_kw), 8.0)
    ref, ours = _models(pkg, do, dev, n, seed=n + step)
    _accumulate_both(ctrl_ref, ctrl, dev, n, views=7, seed=step)
    assert np.allclose(np_(ctrl.grad_accum_), ctrl_ref.grad_accum_.numpy(), rtol=1e-6, atol=0)
    assert np.array_equal(np_(ctrl.grad_count_), ctrl_ref.grad_count_.numpy())
    assert np.array_equal(np_(ctrl.max_radii_2d_), ctrl_ref.max_radii_2d_.numpy())
    noise = torch.randn((2, n, 3), generator=torch.Generator().manual_seed(9))
    s_ref = ctrl_ref.densify(ref, step, noise)
    s = ctrl.densify(ours, step, noise.to(dev))
    assert (s.num_before, s.num_cloned, s.num_split, s.num_pruned, s.num_after) == \
        (s_ref.num_before, s_ref.num_cloned, s_ref.num_split, s_ref.num_pruned, s_ref.num_after)
    if "grad_threshold" not in kw:
        assert s.num_cloned > 0 and s.num_split > 0 and s.num_pruned > s.num_split      # every path exercised
    _compare_models(ref, ours, n_children_start=s.num_after - 2 * s.num_split)
    assert ctrl.grad_accum_.shape[0] == s.num_after and float(ctrl.grad_accum_.abs().sum()) == 0.0


def test_max_gaussians_budget(pkg, do, dev):
    n = 2000
    for cap in (2300, 2050, 1900):                                 # clone budget only / clone + split budget / none
        cfg_kw = dict(densify_from=0, densify_every=5, grad_threshold=0.0002, opacity_threshold=0.0001, max_gaussians=cap)
        ctrl_ref = do.DensificationController(do.DensificationConfig(**cfg_kw), 8.0)
        ctrl = pkg.DensificationController(pkg.DensificationConfig(**cfg_kw), 8.0)
        ref, ours = _models(pkg, do, dev, n, seed=cap)
        _accumulate_both(ctrl_ref, ctrl, dev, n, views=3, seed=cap)
        noise = torch.randn((2, n, 3), generator=torch.Generator().manual_seed(3))
        s_ref = ctrl_ref.densify(ref, 5, noise)
        s = ctrl.densify(ours, 5, noise.to(dev))
        assert (s.num_cloned, s.num_split, s.num_after) == (s_ref.num_cloned, s_ref.num_split, s_ref.num_after)
        assert n + s.num_cloned + 2 * s.num_split <= max(cap, n)
        _compare_models(ref, ours, n_children_start=s.num_after - 2 * s.num_split)


def test_moments_ride_along_and_edges(pkg, do, dev):
    n = 1500
    cfg_kw = dict(densify_from=0, densify_every=5, grad_threshold=0.0002, opacity_threshold=0.05)
    ctrl = pkg.DensificationController(pkg.DensificationConfig(**cfg_kw), 8.0)
    _, ours = _models(pkg, do, dev, n, seed=5)
    opt = pkg.FusedAdam(ours)
    for i in range(5):
        opt.m_[i] = torch.randn_like(opt.m_[i])
        opt.v_[i] = torch.rand_like(opt.v_[i])
    m_before = [t.clone() for t in opt.m_]
    g = torch.Generator().manual_seed(1)
    ctrl.accumulate_gradients((torch.randn((n, 2), generator=g) * 0.0005).to(dev), torch.ones(n, dtype=torch.int32, device=dev))
    flags, _ = ctrl._classify(ours, 5)
    keep = ((flags >> 2) & 1).bool() & ~((flags >> 1) & 1).bool()
    s = ctrl.densify(ours, 5, optimizer=opt)
    kept = int(keep.sum())
    assert s.num_after == kept + s.num_cloned + 2 * s.num_split and s.num_cloned > 0 and s.num_split > 0
    for i in range(5):
        assert opt.m_[i].shape == getattr(ours, opt._names[i]).shape == opt.v_[i].shape
        assert torch.equal(opt.m_[i][:kept], m_before[i][keep])                  # survivors keep their moments
        assert float(opt.m_[i][kept:].abs().sum()) == 0.0 and float(opt.v_[i][kept:].abs().sum()) == 0.0
    # everything pruned -> an empty, still valid model
    ctrl = pkg.DensificationController(pkg.DensificationConfig(densify_from=0, densify_every=5, opacity_threshold=0.999999,
                                                               grad_threshold=1000.0), 8.0)
    _, ours = _models(pkg, do, dev, 100, seed=6)
    s = ctrl.densify(ours, 5)
    assert s.num_after == 0 and s.num_pruned == 100 and ours.num_gaussians() == 0 and ours.is_valid()
    # empty model
    empty = pkg.GaussianModel(torch.zeros((0, 3), device=dev), torch.zeros((0, 3, 1), device=dev), torch.zeros((0, 1), device=dev),
                              torch.zeros((0, 4), device=dev), torch.zeros((0, 3), device=dev))
    assert ctrl.densify(empty, 5).num_after == 0
    # reset_opacity (densification.cpp:331-334)
    _, ours = _models(pkg, do, dev, 10, seed=7)
    ctrl.reset_opacity(ours)
    assert np.allclose(np_(ours.opacities), -4.59511985013459)
