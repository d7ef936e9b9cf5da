"""SURVEY §8(f) N3 on the GPU: write_gaussian_ply / read_gaussian_ply (csrc/ply.hip through the C ABI) against
the oracle (oracle/ply_oracle.py): byte-identical files, exact round trips, name-based loading, the
optimizer-state extension, the reference's error behaviour."""
import numpy as np
import pytest
import torch

from test_ply_oracle import load_ply_oracle, make_model
from util import np_

pytestmark = pytest.mark.gpu
NAMES = ("positions", "sh_coeffs", "opacities", "scales", "rotations")


@pytest.fixture(scope="module")
def po():
    return load_ply_oracle()


def _to_model(pkg, d, dev):
    return pkg.GaussianModel(**{k: torch.from_numpy(d[k]).to(dev) for k in NAMES})


@pytest.mark.parametrize("n,c", [(1, 1), (257, 4), (5000, 16), (33, 9)])
def test_file_is_byte_identical_and_round_trips(pkg, po, dev, tmp_path, n, c):
    ref = make_model(n, c, seed=n)
    ours, theirs = tmp_path / "ours.ply", tmp_path / "oracle.ply"
    assert pkg.write_gaussian_ply(ours, _to_model(pkg, ref, dev))
    po.write_gaussian_ply(theirs, ref)
    assert ours.read_bytes() == theirs.read_bytes()
    back = pkg.read_gaussian_ply(theirs)                                  # on the CPU, like the reference
    assert not back.positions.is_cuda and back.is_valid()
    assert all(np.array_equal(np_(getattr(back, k)), ref[k]) for k in NAMES)
    on_dev = pkg.read_gaussian_ply(ours, device=dev)
    assert on_dev.positions.is_cuda and all(np.array_equal(np_(getattr(on_dev, k)), ref[k]) for k in NAMES)


def test_optimizer_state_resumes_bit_exactly(pkg, po, dev, tmp_path):
    n, c = 3000, 16
    model = _to_model(pkg, make_model(n, c, 1), dev)
    opt = pkg.FusedAdam(model)
    g = torch.Generator().manual_seed(2)
    mk = lambda t: torch.randn(t.shape, generator=g).to(dev)
    grads = lambda: pkg.BackwardOutput(mk(model.positions), mk(model.rotations), mk(model.scales), mk(model.opacities),
                                       mk(model.sh_coeffs), None)
    for _ in range(3):
        opt.apply_gradients(grads())
        opt.step()
    path = tmp_path / "ckpt.ply"
    assert pkg.write_gaussian_ply(path, model, optimizer=opt)
    # the oracle reads the same state out of our file
    ref_model, ref_state = po.read_gaussian_ply(path)
    assert ref_state["step"] == 3
    for i, k in enumerate(opt._names):
        assert np.array_equal(ref_state["m"][k], np_(opt.m_[i])) and np.array_equal(ref_state["v"][k], np_(opt.v_[i]))
    # resume: a fresh optimizer on the loaded model continues exactly like the original
    model2, state = pkg.read_gaussian_ply(path, device=dev, return_state=True)
    opt2 = pkg.FusedAdam(model2)
    pkg.restore_optimizer(opt2, state)
    nxt = grads()
    opt.apply_gradients(nxt); opt.step()
    opt2.apply_gradients(nxt); opt2.step()
    assert opt2.step_count_ == 4
    for k in NAMES:
        assert torch.equal(getattr(model, k), getattr(model2, k))
    # a file with state still loads as a plain model (what the reference's reader would do with it)
    plain = pkg.read_gaussian_ply(path)
    assert all(np.array_equal(np_(getattr(plain, k)), ref_model[k]) for k in NAMES)
    model3, none_state = pkg.read_gaussian_ply(tmp_path / "ckpt.ply", return_state=False), None
    assert model3.is_valid() and none_state is None


def test_name_based_loading_and_errors(pkg, po, dev, tmp_path):
    ref = make_model(40, 4, 9)
    path = tmp_path / "m.ply"
    po.write_gaussian_ply(path, ref)
    n, names, off, _ = po.parse_header(path.read_bytes())
    data = np.frombuffer(path.read_bytes(), "<f4", n * len(names), off).reshape(n, len(names))
    perm = np.random.default_rng(4).permutation(len(names))
    p2 = tmp_path / "perm.ply"
    p2.write_bytes(("ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n +
                    "".join("property float %s\n" % names[i] for i in perm) + "property float extra\nend_header\n").encode()
                   + np.concatenate([data[:, perm], np.zeros((n, 1), np.float32)], 1).astype("<f4").tobytes())
    back = pkg.read_gaussian_ply(p2)
    assert all(np.array_equal(np_(getattr(back, k)), ref[k]) for k in NAMES)
    with pytest.raises(RuntimeError, match="Failed to open PLY file"):
        pkg.read_gaussian_ply(tmp_path / "nope.ply")
    (tmp_path / "short.ply").write_bytes(path.read_bytes()[:-4])
    with pytest.raises(RuntimeError, match="Failed to read PLY binary data"):
        pkg.read_gaussian_ply(tmp_path / "short.ply")
    (tmp_path / "ascii.ply").write_bytes(b"ply\nformat ascii 1.0\nend_header\n")
    with pytest.raises(RuntimeError, match="binary_little_endian"):
        pkg.read_gaussian_ply(tmp_path / "ascii.ply")
    # the writer returns False, like the reference, for an invalid model or an unwritable path
    assert not pkg.write_gaussian_ply(tmp_path / "x.ply", pkg.GaussianModel())
    assert not pkg.write_gaussian_ply(tmp_path / "no_such_dir" / "x.ply", _to_model(pkg, ref, dev))


@pytest.mark.parametrize("n,degree", [(50, 3), (20, 0), (30, 2)])
def test_reference_roundtrip_degrees_on_the_product(pkg, dev, tmp_path, n, degree):
    """tests/test_gaussian_model.cpp:97-141 (RoundtripDegree3 / 0 / 2) through the HIP record pack/unpack."""
    ref = make_model(n, (degree + 1) ** 2, seed=100 + degree)
    original = _to_model(pkg, ref, dev)
    path = tmp_path / ("test_d%d.ply" % degree)
    assert original.save_ply(path) if hasattr(original, "save_ply") else pkg.write_gaussian_ply(path, original)
    assert path.exists()
    loaded = pkg.read_gaussian_ply(path)
    assert loaded.is_valid() and loaded.num_gaussians() == n and loaded.max_sh_degree() == degree
    for k in NAMES:
        assert torch.allclose(getattr(original, k).cpu(), getattr(loaded, k), rtol=1e-5, atol=1e-5), k


def test_reference_empty_model_on_the_product(pkg, dev, tmp_path):
    """tests/test_gaussian_model.cpp:143-157 (EmptyModel): an empty model saves and loads back empty."""
    m = pkg.GaussianModel(torch.zeros((0, 3), device=dev), torch.zeros((0, 3, 16), device=dev),
                          torch.zeros((0, 1), device=dev), torch.zeros((0, 4), device=dev), torch.zeros((0, 3), device=dev))
    path = tmp_path / "empty.ply"
    assert pkg.write_gaussian_ply(path, m)
    loaded = pkg.read_gaussian_ply(path)
    assert loaded.num_gaussians() == 0
