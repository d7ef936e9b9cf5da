"""SURVEY §8(f) N2 — the densification oracle (oracle/densify_oracle.py: the reference's libtorch op
sequence on CPU) against the reference's own known answers, tests/test_densification.cpp:48-361 (all
CUDA-only there), and the host mirror's schedule logic (pure Python, no GPU)."""
import importlib.util
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_densify_oracle():
    spec = importlib.util.spec_from_file_location("cugs_densify_oracle", os.path.join(ROOT, "oracle", "densify_oracle.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["cugs_densify_oracle"] = mod          # dataclasses resolve their module through sys.modules
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def do():
    return load_densify_oracle()


def make_test_model(do, n, scale_val=-2.0, opacity_val=2.0, seed=0):
    """make_test_model (test_densification.cpp:27-42)"""
    g = torch.Generator().manual_seed(seed)
    rot = torch.randn((n, 4), generator=g)
    rot = rot / rot.norm(2, 1, True).clamp_min(1e-8)
    return do.Model(torch.randn((n, 3), generator=g) * 0.5, torch.randn((n, 3, 1), generator=g) * 0.1,
                    torch.full((n, 1), opacity_val), rot, torch.full((n, 3), scale_val))


def is_valid(m):
    n = m.positions.shape[0]
    return (m.positions.shape == (n, 3) and m.sh_coeffs.shape[:2] == (n, 3) and m.opacities.shape == (n, 1)
            and m.rotations.shape == (n, 4) and m.scales.shape == (n, 3))


def _cfg(do, **kw):
    base = dict(densify_from=0, densify_until=1000, densify_every=5)
    base.update(kw)
    return do.DensificationConfig(**base)


def _noise(n, seed=1):
    return torch.randn((2, n, 3), generator=torch.Generator().manual_seed(seed))


@pytest.mark.parametrize("which", ["oracle", "mirror"])
def test_schedule(do, which):
    if which == "mirror":
        import __graft_entry__ as ge
        ns = ge.load_package()
    else:
        ns = do
    ctrl = ns.DensificationController(ns.DensificationConfig(densify_from=500, densify_until=15000, densify_every=100), 10.0)
    for s in (0, 100, 400, 499, 501, 550, 999, 15100, 20000):                     # :58-79
        assert not ctrl.should_densify(s)
    for s in (500, 600, 1000, 14900, 15000):
        assert ctrl.should_densify(s)
    ctrl = ns.DensificationController(ns.DensificationConfig(densify_from=500, opacity_reset_every=3000), 10.0)
    assert not ctrl.should_reset_opacity(0)                                       # :91-100
    assert all(ctrl.should_reset_opacity(s) for s in (3000, 6000, 9000))
    assert not ctrl.should_reset_opacity(3001) and not ctrl.should_reset_opacity(4000)


def test_accumulate_and_invisible(do):
    n = 10
    ctrl = do.DensificationController(_cfg(do, grad_threshold=0.0001), 10.0)
    model = make_test_model(do, n, -5.0)
    for _ in range(5):                                                            # :134-161
        ctrl.accumulate_gradients(torch.ones(n, 2) * 10.0, torch.zeros(n, dtype=torch.int32))
    assert float(ctrl.grad_accum_.abs().sum()) == 0.0 and float(ctrl.grad_count_.sum()) == 0.0
    stats = ctrl.densify(model, 5, _noise(n))
    assert stats.num_cloned == 0 and stats.num_split == 0 and stats.num_before == n
    # visible ones accumulate the 2-norm and the count; max radius tracks every Gaussian
    ctrl = do.DensificationController(_cfg(do), 10.0)
    g = torch.tensor([[3.0, 4.0], [1.0, 0.0], [0.0, 0.0]])
    ctrl.accumulate_gradients(g, torch.tensor([2, 0, 7], dtype=torch.int32))
    ctrl.accumulate_gradients(g, torch.tensor([5, 3, 0], dtype=torch.int32))
    assert ctrl.grad_accum_.tolist() == [10.0, 1.0, 0.0] and ctrl.grad_count_.tolist() == [2.0, 1.0, 1.0]
    assert ctrl.max_radii_2d_.tolist() == [5.0, 3.0, 7.0]


def test_clone_split_prune_known_answers(do):
    n = 10
    hi = lambda c: [c.accumulate_gradients(torch.ones(n, 2), torch.ones(n, dtype=torch.int32)) for _ in range(5)]
    # :167-198 high gradient + small scale -> cloned, order = [originals | clones]
    ctrl = do.DensificationController(_cfg(do, grad_threshold=0.0001, percent_dense=0.01), 10.0)
    model = make_test_model(do, n, -5.0, 2.0)
    before = model.positions.clone()
    hi(ctrl)
    stats = ctrl.densify(model, 5, _noise(n))
    assert (stats.num_cloned, stats.num_split, stats.num_after) == (10, 0, 20) and is_valid(model)
    assert torch.equal(model.positions[:n], before) and torch.equal(model.positions[n:], before)
    assert ctrl.grad_accum_.shape[0] == 20 and float(ctrl.grad_accum_.sum()) == 0.0        # reset to the new size
    # :200-229 high gradient + large scale -> split into two children, originals removed
    ctrl = do.DensificationController(_cfg(do, grad_threshold=0.0001, percent_dense=0.01), 10.0)
    model = make_test_model(do, n, 0.0, 2.0)
    parents, noise = model.positions.clone(), _noise(n)
    hi(ctrl)
    stats = ctrl.densify(model, 5, noise)
    assert (stats.num_cloned, stats.num_split, stats.num_pruned, stats.num_after) == (0, 10, 10, 20) and is_valid(model)
    new_scale = torch.full((n, 3), 0.0) - float(torch.log(torch.tensor(1.6)))
    assert torch.allclose(model.scales, torch.cat([new_scale, new_scale]))
    assert torch.allclose(model.positions[:n], parents + noise[0] * torch.exp(new_scale), atol=1e-6)
    assert torch.allclose(model.positions[n:], parents + noise[1] * torch.exp(new_scale), atol=1e-6)
    # :231-266 low opacity -> pruned
    ctrl = do.DensificationController(_cfg(do, opacity_threshold=0.5, grad_threshold=1000.0), 10.0)
    model = make_test_model(do, n)
    model.opacities[:5] = 5.0
    model.opacities[5:] = -5.0
    keep_pos = model.positions[:5].clone()
    ctrl.accumulate_gradients(torch.zeros(n, 2), torch.ones(n, dtype=torch.int32))
    stats = ctrl.densify(model, 5, _noise(n))
    assert (stats.num_pruned, stats.num_after) == (5, 5) and is_valid(model)
    assert torch.equal(model.positions, keep_pos)


def test_opacity_reset_full_cycle_and_cap(do):
    model = make_test_model(do, 20, -2.0, 3.0)                                    # :268-285
    ctrl = do.DensificationController(do.DensificationConfig(), 10.0)
    ctrl.reset_opacity(model)
    assert torch.allclose(model.opacities, torch.full((20, 1), -4.595), atol=0.01)
    n = 20                                                                        # :287-329
    ctrl = do.DensificationController(_cfg(do, grad_threshold=0.0001, percent_dense=0.01, opacity_threshold=0.5), 10.0)
    model = make_test_model(do, n)
    model.scales[:10] = -5.0
    model.scales[10:] = 0.0
    model.opacities[:5] = -5.0
    model.opacities[5:] = 3.0
    for _ in range(5):
        ctrl.accumulate_gradients(torch.ones(n, 2), torch.ones(n, dtype=torch.int32))
    stats = ctrl.densify(model, 5, _noise(n))
    # 10 clones (5 of their originals pruned for opacity), 10 splits (originals removed)
    assert (stats.num_cloned, stats.num_split, stats.num_pruned, stats.num_after) == (10, 10, 15, 35) and is_valid(model)
    n = 10                                                                        # :331-361
    ctrl = do.DensificationController(_cfg(do, grad_threshold=0.0001, percent_dense=0.01, max_gaussians=15), 10.0)
    model = make_test_model(do, n, -5.0, 2.0)
    for k in range(5):
        ctrl.accumulate_gradients(torch.arange(1, n + 1, dtype=torch.float32).reshape(n, 1).repeat(1, 2),
                                  torch.ones(n, dtype=torch.int32))
    top = model.positions[5:].clone()
    stats = ctrl.densify(model, 5, _noise(n))
    assert stats.num_after == 15 and model.num_gaussians() == 15 and is_valid(model)
    assert torch.equal(model.positions[n:], top)                                  # the five highest-gradient ones
