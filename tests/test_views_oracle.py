"""SURVEY §8(f) N4 — the view oracle (oracle/views_oracle.py) against the reference's own image tests
(tests/test_dataset.cpp:300-355) and the host mirror's PPM loader (pure Python, no GPU)."""
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_views_oracle():
    spec = importlib.util.spec_from_file_location("cugs_views_oracle", os.path.join(ROOT, "oracle", "views_oracle.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def vo():
    return load_views_oracle()


def write_ppm(path, w, h, rgb):
    """create_dummy_ppm (test_dataset.cpp): a P6 file filled with one colour."""
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (w, h))
        f.write(bytes(rgb) * (w * h))


def test_reference_image_tests(vo, tmp_path):
    import __graft_entry__ as ge
    pkg = ge.load_package()
    p = tmp_path / "test.ppm"
    write_ppm(p, 4, 3, (255, 0, 128))                                            # :300-315 LoadPPM
    raw = pkg.load_image_u8(p)
    assert raw.shape == (3, 4, 3)
    img = vo.convert_u8(raw)
    assert img.size == 4 * 3 * 3 and abs(img[0, 0, 0] - 1.0) <= 1e-3 and img[0, 0, 1] == 0.0
    assert abs(img[0, 0, 2] - 128.0 / 255.0) <= 1e-3
    src = np.full((4, 4, 3), 0.5, np.float32)                                    # :317-332 ResizeImage
    dst = vo.resize_image(src, 2, 2)
    assert dst.shape == (2, 2, 3) and np.all(np.abs(dst - 0.5) <= 1e-3)
    write_ppm(p, 8, 6, (100, 100, 100))                                          # :334-355 sizes of load_image_resized
    raw = pkg.load_image_u8(p)
    assert vo.target(raw, 8, 6).shape == (6, 8, 3)
    assert vo.target(raw, max(1, 8 // 2), max(1, 6 // 2)).shape == (3, 4, 3)
    with pytest.raises(RuntimeError, match="Failed to load image"):              # :357-361
        pkg.load_image_u8(tmp_path / "nonexistent_file.png")
    (tmp_path / "bad.ppm").write_bytes(b"P5\n2 2\n255\n....")
    with pytest.raises(RuntimeError, match="Failed to load image"):
        pkg.load_image_u8(tmp_path / "bad.ppm")
    with pytest.raises(RuntimeError, match="Invalid target dimensions"):
        vo.resize_image(src, 0, 2)


def test_resize_against_a_scalar_restatement(vo):
    """The vectorised oracle equals the reference's loops written out one pixel at a time."""
    rng = np.random.default_rng(0)
    src = rng.random((7, 5, 3), dtype=np.float32)
    F = np.float32
    for tw, th in ((3, 4), (10, 14), (5, 7), (1, 1)):
        want = np.zeros((th, tw, 3), F)
        xs, ys = F(5) / F(tw), F(7) / F(th)
        for y in range(th):
            sy = (F(y) + F(0.5)) * ys - F(0.5)
            y0 = max(0, int(np.floor(sy))); y1 = min(6, y0 + 1); fy = F(sy - F(y0))
            for x in range(tw):
                sx = (F(x) + F(0.5)) * xs - F(0.5)
                x0 = max(0, int(np.floor(sx))); x1 = min(4, x0 + 1); fx = F(sx - F(x0))
                for c in range(3):
                    top = F(src[y0, x0, c] + F(F(src[y0, x1, c] - src[y0, x0, c]) * fx))
                    bot = F(src[y1, x0, c] + F(F(src[y1, x1, c] - src[y1, x0, c]) * fx))
                    want[y, x, c] = F(top + F(F(bot - top) * fy))
        assert np.array_equal(vo.resize_image(src, tw, th), want)
