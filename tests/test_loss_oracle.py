"""SURVEY §8(f) N1 — the loss oracle (oracle/loss_oracle.py: the reference's libtorch op sequence on
CPU) against the reference's own known answers, tests/test_loss.cpp:30-136 (all CUDA-only there)."""
import importlib.util
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lo():
    spec = importlib.util.spec_from_file_location("cugs_loss_oracle", os.path.join(ROOT, "oracle", "loss_oracle.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _rand(h, w, seed):
    return torch.rand((h, w, 3), generator=torch.Generator().manual_seed(seed))


def _uniform(h, w, v):
    return torch.full((h, w, 3), v, dtype=torch.float32)


def test_l1(lo):
    img = _rand(64, 64, 0)
    assert abs(float(lo.l1_loss(img, img))) <= 1e-6                                       # :30-37
    assert abs(float(lo.l1_loss(_uniform(32, 32, 0.8), _uniform(32, 32, 0.3))) - 0.5) <= 1e-5   # :39-48
    assert float(lo.l1_loss(_rand(64, 64, 1), _rand(64, 64, 2))) >= 0.0                    # :50-58


def test_ssim(lo):
    img = _rand(64, 64, 3)
    assert abs(float(lo.ssim(img, img).mean()) - 1.0) <= 1e-4                              # :64-72
    assert float(lo.ssim(_uniform(64, 64, 0.0), _uniform(64, 64, 1.0)).mean()) < 0.1       # :74-83
    a, b = _rand(64, 64, 4), _rand(64, 64, 5)
    assert abs(float(lo.ssim(a, b).mean()) - float(lo.ssim(b, a).mean())) <= 1e-5          # :85-95
    m = lo.ssim(a, b)
    assert float(m.min()) >= -1.0 - 1e-5 and float(m.max()) <= 1.0 + 1e-5                  # :97-107
    assert m.shape == (64, 64)


def test_combined(lo):
    img = _rand(64, 64, 6)
    assert abs(float(lo.combined_loss(img, img))) <= 1e-4                                  # :113-121
    target = _rand(64, 64, 7)
    g = torch.Generator().manual_seed(8)
    close = (target + 0.05 * torch.randn(target.shape, generator=g)).clamp(0, 1)
    far = (target + 0.5 * torch.randn(target.shape, generator=g)).clamp(0, 1)
    assert float(lo.combined_loss(close, target)) < float(lo.combined_loss(far, target))   # :123-137


def test_gaussian_kernel_is_normalised_and_separable(lo):
    k = lo.gaussian_kernel(11)
    assert k.shape == (3, 1, 11, 11) and abs(float(k[0, 0].sum()) - 1.0) < 1e-6
    u, s, _ = np.linalg.svd(k[0, 0].numpy().astype(np.float64))
    assert s[1] / s[0] < 1e-6          # rank one: the HIP kernels use its separable factor


def test_grad_is_finite_differences(lo):
    r, t = _rand(12, 10, 9).numpy(), _rand(12, 10, 10).numpy()
    loss, grad, _, _ = lo.combined_loss_and_grad(r, t, 0.2)
    f = lambda a: float(lo.combined_loss(torch.from_numpy(a), torch.from_numpy(t), 0.2))
    eps = 1e-2
    for (i, j, c) in [(0, 0, 0), (5, 4, 1), (11, 9, 2), (6, 0, 0)]:
        p, m = r.copy(), r.copy()
        p[i, j, c] += eps
        m[i, j, c] -= eps
        num = (f(p) - f(m)) / (2 * eps)
        assert abs(num - grad[i, j, c]) <= 2e-4 + 0.05 * abs(num)
