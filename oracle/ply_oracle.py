"""CPU oracle for SURVEY §8(f) N3: the Gaussian-model PLY checkpoint.  TEST INFRASTRUCTURE.

A numpy restatement of write_gaussian_ply / read_gaussian_ply (src/utils/ply_io.cpp:98-196, 258-351): the
same header text, the same per-vertex float order, properties looked up by name on the way back.  The
reference's tests (tests/test_ply_io.cpp) cover only its point-cloud and camera writers, so this format has
no golden vector of the reference's own: PARITY UNPINNED beyond the byte layout stated in its source, which
tests/test_ply_oracle.py checks field by field.  The optimizer-state extension (properties m_*, v_* and the
header comment carrying the step count) is not in the reference; a file that has it still loads in the
reference's reader, which ignores comments and unknown properties.
Only tests/ and tools/ import this module.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np

GROUPS = ("positions", "sh_coeffs", "opacities", "scales", "rotations")


def property_names(num_coeffs: int, with_state: bool = False) -> List[str]:
    """ply_io.cpp:123-151 (+ the state extension)."""
    base = ["x", "y", "z"]
    model = ["f_dc_%d" % i for i in range(3)] + ["f_rest_%d" % i for i in range(3 * (num_coeffs - 1))]
    model += ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    names = base + ["nx", "ny", "nz"] + model
    if with_state:
        names += ["m_" + p for p in base + model] + ["v_" + p for p in base + model]
    return names


def header_bytes(n: int, num_coeffs: int, step: Optional[int] = None) -> bytes:
    lines = ["ply", "format binary_little_endian 1.0"]
    if step is not None:
        lines.append("comment cugs_adam_step %d" % step)
    lines.append("element vertex %d" % n)
    lines += ["property float " + p for p in property_names(num_coeffs, step is not None)]
    lines.append("end_header")
    return ("\n".join(lines) + "\n").encode("ascii")


def _block(t: Dict[str, np.ndarray], normals: bool) -> np.ndarray:
    """One [x .. rot_3] block per Gaussian in the reference's order (ply_io.cpp:156-190)."""
    n, _, c = t["sh_coeffs"].shape
    cols = [t["positions"].reshape(n, 3)]
    if normals:
        cols.append(np.zeros((n, 3), np.float32))
    sh = t["sh_coeffs"]
    cols.append(sh[:, :, 0])                                              # f_dc_0..2
    cols.append(sh[:, :, 1:].transpose(0, 2, 1).reshape(n, 3 * (c - 1)))  # for k: for ch: sh[ch][k]
    cols += [t["opacities"].reshape(n, 1), t["scales"].reshape(n, 3), t["rotations"].reshape(n, 4)]
    return np.concatenate([np.asarray(x, np.float32) for x in cols], axis=1)


def vertex_array(model: Dict[str, np.ndarray], m=None, v=None) -> np.ndarray:
    blocks = [_block(model, True)]
    if m is not None:
        blocks += [_block(m, False), _block(v, False)]
    return np.ascontiguousarray(np.concatenate(blocks, axis=1), dtype="<f4")


def write_gaussian_ply(path, model: Dict[str, np.ndarray], m=None, v=None, step: Optional[int] = None) -> bool:
    n, _, c = model["sh_coeffs"].shape
    with open(path, "wb") as f:
        f.write(header_bytes(n, c, step if m is not None else None))
        f.write(vertex_array(model, m, v).tobytes())
    return True


def parse_header(buf: bytes) -> Tuple[int, List[str], int, Optional[int]]:
    """parse_ply_header (ply_io.cpp:211-250): vertex count, property names in order, data offset (+ step)."""
    pos, lines = 0, []
    while True:
        end = buf.index(b"\n", pos)
        line = buf[pos:end].decode("ascii", "replace").rstrip("\r")
        pos = end + 1
        lines.append(line)
        if line == "end_header":
            break
    if "ply" not in lines[0]:
        raise RuntimeError("Not a PLY file")
    if "binary_little_endian" not in lines[1]:
        raise RuntimeError("Only binary_little_endian PLY is supported")
    count, names, step = 0, [], None
    for line in lines[2:]:
        tok = line.split()
        if not tok:
            continue
        if tok[0] == "element" and tok[1] == "vertex":
            count = int(tok[2])
        elif tok[0] == "property":
            names.append(tok[2])
        elif tok[0] == "comment" and len(tok) == 3 and tok[1] == "cugs_adam_step":
            step = int(tok[2])
    return count, names, pos, step


def read_gaussian_ply(path):
    """read_gaussian_ply (ply_io.cpp:258-351) -> (model dict, state dict or None)."""
    buf = open(path, "rb").read()
    n, names, off, step = parse_header(buf)
    index = {nm: i for i, nm in enumerate(names)}
    num_rest = 0
    while "f_rest_%d" % num_rest in index:
        num_rest += 1
    c = 1 + num_rest // 3
    need = n * len(names) * 4
    if len(buf) - off < need:
        raise RuntimeError("Failed to read PLY binary data")
    data = np.frombuffer(buf, "<f4", n * len(names), off).reshape(n, len(names))

    def get(name):
        if name not in index:
            raise RuntimeError("Missing PLY property: " + name)
        return data[:, index[name]]

    def gather(prefix):
        sh = np.zeros((n, 3, c), np.float32)
        for ch in range(3):
            sh[:, ch, 0] = get(prefix + "f_dc_%d" % ch)
        for k in range(1, c):
            for ch in range(3):
                sh[:, ch, k] = get(prefix + "f_rest_%d" % ((k - 1) * 3 + ch))
        return dict(positions=np.stack([get(prefix + a) for a in "xyz"], 1).astype(np.float32), sh_coeffs=sh,
                    opacities=get(prefix + "opacity").reshape(n, 1).astype(np.float32),
                    scales=np.stack([get(prefix + "scale_%d" % i) for i in range(3)], 1).astype(np.float32),
                    rotations=np.stack([get(prefix + "rot_%d" % i) for i in range(4)], 1).astype(np.float32))

    model = gather("")
    state = None
    if "m_x" in index and "v_x" in index:
        state = dict(m=gather("m_"), v=gather("v_"), step=step if step is not None else 0)
    return model, state
