"""SURVEY §8(f) N3 — the PLY checkpoint oracle (oracle/ply_oracle.py) against the byte layout the reference's
source states (utils/ply_io.cpp:123-190: header text, per-vertex float order, (coefficient, channel) interleave of
f_rest) and its reader's rules (:211-351), and pinned by the reference's own Gaussian-PLY tests
(tests/test_gaussian_model.cpp:97-160: RoundtripDegree3 / Degree0 / Degree2, EmptyModel), restated below with the
reference's model sizes and its allclose(1e-5, 1e-5) bar (the round trips here are in fact exact)."""
import importlib.util
import os
import struct

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_ply_oracle():
    spec = importlib.util.spec_from_file_location("cugs_ply_oracle", os.path.join(ROOT, "oracle", "ply_oracle.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def po():
    return load_ply_oracle()


def make_model(n, c, seed=0):
    rng = np.random.default_rng(seed)
    f = lambda *s: rng.standard_normal(s).astype(np.float32)
    return dict(positions=f(n, 3), sh_coeffs=f(n, 3, c), opacities=f(n, 1), scales=f(n, 3), rotations=f(n, 4))


@pytest.mark.parametrize("c", [1, 4, 16])
def test_layout_matches_the_reference_source(po, tmp_path, c):
    n = 7
    model = make_model(n, c)
    path = tmp_path / "m.ply"
    assert po.write_gaussian_ply(path, model)
    buf = path.read_bytes()
    head, _, body = buf.partition(b"end_header\n")
    lines = head.decode().split("\n")[:-1]
    assert lines[0] == "ply" and lines[1] == "format binary_little_endian 1.0" and lines[2] == "element vertex 7"
    props = [l.split()[-1] for l in lines[3:]]
    assert all(l.startswith("property float ") for l in lines[3:])
    assert props[:9] == ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"]
    assert props[9:9 + 3 * (c - 1)] == ["f_rest_%d" % i for i in range(3 * (c - 1))]
    assert props[-8:] == ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    row = 14 + 3 * c
    assert len(props) == row and len(body) == n * row * 4
    rec = struct.unpack("<%df" % row, body[3 * row * 4:4 * row * 4])            # vertex 3, field by field
    assert rec[0:3] == tuple(model["positions"][3]) and rec[3:6] == (0.0, 0.0, 0.0)
    assert rec[6:9] == tuple(model["sh_coeffs"][3, :, 0])
    for k in range(1, c):
        for ch in range(3):
            assert rec[9 + (k - 1) * 3 + ch] == model["sh_coeffs"][3, ch, k]     # ply_io.cpp:170-174
    tail = rec[9 + 3 * (c - 1):]
    assert tail[0] == model["opacities"][3, 0] and tail[1:4] == tuple(model["scales"][3])
    assert tail[4:8] == tuple(model["rotations"][3])


def test_round_trip_and_reader_rules(po, tmp_path):
    model = make_model(50, 16, seed=1)
    path = tmp_path / "m.ply"
    po.write_gaussian_ply(path, model)
    back, state = po.read_gaussian_ply(path)
    assert state is None and all(np.array_equal(back[k], model[k]) for k in model)
    # properties are found by NAME: a file with its columns permuted and a foreign one added loads the same
    n, names, off, _ = po.parse_header(path.read_bytes())
    data = np.frombuffer(path.read_bytes(), "<f4", n * len(names), off).reshape(n, len(names))
    perm = np.random.default_rng(2).permutation(len(names))
    names2 = [names[i] for i in perm] + ["confidence"]
    data2 = np.concatenate([data[:, perm], np.ones((n, 1), np.float32)], 1).astype("<f4")
    p2 = tmp_path / "perm.ply"
    p2.write_bytes(("ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n +
                    "".join("property float %s\n" % s for s in names2) + "end_header\n").encode() + data2.tobytes())
    back2, _ = po.read_gaussian_ply(p2)
    assert all(np.array_equal(back2[k], model[k]) for k in model)
    # errors (ply_io.cpp:221-227, 297-299, 290-294)
    (tmp_path / "bad1.ply").write_bytes(b"plx\nformat binary_little_endian 1.0\nend_header\n")
    (tmp_path / "bad2.ply").write_bytes(b"ply\nformat ascii 1.0\nend_header\n")
    for name in ("bad1.ply", "bad2.ply"):
        with pytest.raises(RuntimeError):
            po.read_gaussian_ply(tmp_path / name)
    (tmp_path / "short.ply").write_bytes(path.read_bytes()[:-4])
    with pytest.raises(RuntimeError, match="Failed to read PLY binary data"):
        po.read_gaussian_ply(tmp_path / "short.ply")
    drop = [i for i, s in enumerate(names) if s != "opacity"]
    (tmp_path / "noopa.ply").write_bytes(("ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n +
                                          "".join("property float %s\n" % names[i] for i in drop) +
                                          "end_header\n").encode() + np.ascontiguousarray(data[:, drop]).tobytes())
    with pytest.raises(RuntimeError, match="Missing PLY property: opacity"):
        po.read_gaussian_ply(tmp_path / "noopa.ply")


def test_state_extension_round_trip_and_stays_loadable_without_it(po, tmp_path):
    model, m, v = make_model(20, 4, 3), make_model(20, 4, 4), make_model(20, 4, 5)
    path = tmp_path / "s.ply"
    po.write_gaussian_ply(path, model, m, v, step=1234)
    back, state = po.read_gaussian_ply(path)
    assert all(np.array_equal(back[k], model[k]) for k in model) and state["step"] == 1234
    assert all(np.array_equal(state["m"][k], m[k]) and np.array_equal(state["v"][k], v[k]) for k in model)
    head = path.read_bytes().partition(b"end_header\n")[0].decode().split("\n")
    assert head[2] == "comment cugs_adam_step 1234" and head[3] == "element vertex 20"
    assert len([l for l in head if l.startswith("property")]) == (14 + 12) + 2 * (11 + 12)


# ---- the reference's own tests: tests/test_gaussian_model.cpp:97-160 (GaussianPlyTest) ---------------------
def _max_sh_degree(sh):                                   # GaussianModel::max_sh_degree (gaussian.hpp)
    return int(round(np.sqrt(sh.shape[2]))) - 1


@pytest.mark.parametrize("n,degree", [(50, 3), (20, 0), (30, 2)])    # RoundtripDegree3, RoundtripDegree0, RoundtripDegree2
def test_reference_roundtrip_degrees(po, tmp_path, n, degree):
    original = make_model(n, (degree + 1) ** 2, seed=100 + degree)   # make_test_model(n, degree): randn tensors
    path = tmp_path / ("test_d%d.ply" % degree)
    assert po.write_gaussian_ply(path, original) and path.exists()   # ASSERT_TRUE(save_ply), exists
    loaded, _ = po.read_gaussian_ply(path)
    assert loaded["positions"].shape == (n, 3) and _max_sh_degree(loaded["sh_coeffs"]) == degree
    assert loaded["sh_coeffs"].shape == (n, 3, (degree + 1) ** 2) and loaded["opacities"].shape == (n, 1)
    for k in original:                                               # EXPECT_TRUE(allclose(..., 1e-5, 1e-5))
        assert np.allclose(original[k], loaded[k], rtol=1e-5, atol=1e-5), k
        assert np.array_equal(original[k], loaded[k]), k             # and in fact exact


def test_reference_empty_model(po, tmp_path):                        # GaussianPlyTest.EmptyModel
    empty = dict(positions=np.zeros((0, 3), np.float32), sh_coeffs=np.zeros((0, 3, 16), np.float32),
                 opacities=np.zeros((0, 1), np.float32), rotations=np.zeros((0, 4), np.float32),
                 scales=np.zeros((0, 3), np.float32))
    path = tmp_path / "empty.ply"
    assert po.write_gaussian_ply(path, empty)
    loaded, _ = po.read_gaussian_ply(path)
    assert loaded["positions"].shape[0] == 0
