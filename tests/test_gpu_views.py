"""SURVEY §8(f) N4 on the GPU: targets from the device-resident view cache (csrc/views.hip through the C ABI)
against the oracle (oracle/views_oracle.py): bit-identical floats, with and without the resize."""
import numpy as np
import pytest
import torch

from test_views_oracle import load_views_oracle, write_ppm
from util import np_

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vo():
    return load_views_oracle()


@pytest.mark.parametrize("sw,sh,tw,th", [(64, 48, 64, 48), (64, 48, 32, 24), (37, 53, 80, 61), (1920, 1080, 960, 540),
                                         (5, 7, 1, 1), (3, 2, 9, 8), (100, 100, 99, 101)])
def test_target_is_bit_identical(pkg, vo, dev, sw, sh, tw, th):
    raw = np.random.default_rng(sw * 1000 + tw).integers(0, 256, (sh, sw, 3), dtype=np.uint8)
    cache = pkg.ViewCache(dev)
    i = cache.add(raw)
    got = cache.target(i, tw, th)
    assert got.shape == (th, tw, 3) and got.dtype == torch.float32 and got.is_cuda
    assert np.array_equal(np_(got).view(np.uint32), vo.target(raw, tw, th).view(np.uint32))


def test_cache_and_file_helpers(pkg, vo, dev, tmp_path):
    cache = pkg.ViewCache(dev)
    for k, (w, h, rgb) in enumerate([(8, 6, (100, 100, 100)), (4, 3, (255, 0, 128))]):
        write_ppm(tmp_path / f"v{k}.ppm", w, h, rgb)
        assert cache.add_file(tmp_path / f"v{k}.ppm") == k
    assert len(cache) == 2 and cache.size(0) == (8, 6) and cache.bytes() == 8 * 6 * 3 + 4 * 3 * 3
    t = np_(cache.target(1, 4, 3))
    assert t[0, 0, 0] == 1.0 and t[0, 0, 1] == 0.0 and abs(t[0, 0, 2] - 128.0 / 255.0) <= 1e-6
    half = pkg.load_image_resized(tmp_path / "v0.ppm", 2, dev)              # test_dataset.cpp:345-355
    assert tuple(half.shape) == (3, 4, 3) and np.allclose(np_(half), 100.0 / 255.0, atol=1e-6)
    full = pkg.load_image_resized(tmp_path / "v0.ppm", 1, dev)
    assert tuple(full.shape) == (6, 8, 3)
    with pytest.raises(RuntimeError, match="Invalid target dimensions"):
        cache.target(0, 0, 5)


def test_prefetched_targets_equal_cached_targets(pkg, dev):
    """StreamedViewCache (views in pinned host memory, uploaded one iteration ahead on a side stream into a
    two-slot device ring): every target is bit-identical to the device-resident ViewCache's, whatever the order,
    with prefetch hits, on-demand misses and slot reuse (8 views through 2 slots, different sizes)."""
    rng = np.random.default_rng(3)
    sizes = [(64, 48), (640, 360), (37, 53), (1920, 1080), (5, 7), (320, 200), (64, 48), (800, 600)]
    views = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for w, h in sizes]
    resident, streamed = pkg.ViewCache(dev), pkg.StreamedViewCache(dev, slots=2)
    for v in views:
        assert resident.add(v) == streamed.add(v)
    assert len(streamed) == 8 and streamed.host_bytes() == resident.bytes() and streamed.size(3) == (1920, 1080)
    tw, th = 480, 270
    order = [0, 1, 2, 3, 4, 5, 6, 7, 3, 3, 1, 6, 0]
    streamed.prefetch(order[0])
    for k, idx in enumerate(order):
        if k + 1 < len(order):
            streamed.prefetch(order[k + 1])                     # next view travels while this one is consumed
        got = streamed.target(idx, tw, th)
        # work on the compute stream between iterations, as a render would be
        busy = torch.ones((512, 512), device=dev) @ torch.ones((512, 512), device=dev)
        want = resident.target(idx, tw, th)
        assert torch.equal(got, want), (k, idx)
        del busy
    assert streamed.misses == 0 and streamed.uploads >= 8
    # on-demand path: no prefetch at all
    cold = pkg.StreamedViewCache(dev, slots=2)
    for v in views[:3]:
        cold.add(v)
    for idx in (2, 0, 1, 0):
        assert torch.equal(cold.target(idx, 100, 80), resident.target(idx, 100, 80))
    assert cold.misses == 3                                      # the last access finds view 0 still in its slot
    # same-size native-resolution target (no resize) through the ring
    assert torch.equal(streamed.target(3, 1920, 1080), resident.target(3, 1920, 1080))
