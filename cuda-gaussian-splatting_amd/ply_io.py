"""Host-side mirror of the reference's Gaussian checkpoint I/O (utils/ply_io.hpp:53-64) over csrc/ply.hip
(SURVEY §8f N3): write_gaussian_ply / read_gaussian_ply in the reference's binary PLY layout, with the vertex
records packed and unpacked on the device (one copy and one file operation each way), plus what the
reference lacks for a true resume: the Adam moments and step count ride in the same file as extra properties
(m_*, v_*) and a header comment, which the reference's own reader skips."""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Tuple

import numpy as np
import torch

from ._lib import check, lib
from .rasterizer import _ptr, _stream, _torch_check
from .types import GaussianModel

_GROUPS = ("positions", "sh_coeffs", "opacities", "scales", "rotations")      # ParamGroup order


def _property_names(num_coeffs: int, with_state: bool) -> List[str]:
    """ply_io.cpp:123-151 (+ m_*, v_*)."""
    base = ["x", "y", "z"]
    model = ["f_dc_%d" % i for i in range(3)] + ["f_rest_%d" % i for i in range(3 * (num_coeffs - 1))]
    model += ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    names = base + ["nx", "ny", "nz"] + model
    if with_state:
        names += ["m_" + p for p in base + model] + ["v_" + p for p in base + model]
    return names


def _canonical_names(num_coeffs: int, with_state: bool) -> List[str]:
    """Model floats in record order without the normals; then m_*, then v_* (cugs_ply_unpack's col_of order)."""
    names = [p for p in _property_names(num_coeffs, False) if p not in ("nx", "ny", "nz")]
    return names + (["m_" + p for p in names] + ["v_" + p for p in names] if with_state else [])


def _ptr_array(tensors) -> "C.Array":
    arr = (C.c_void_p * 5)()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


def write_gaussian_ply(path, model: GaussianModel, optimizer=None) -> bool:
    """ply_io.cpp:98-196.  Returns False (as the reference does) for an invalid model or an unwritable path.
    `optimizer` (FusedAdam): its moments and step count are stored too."""
    if not model.is_valid():
        return False
    dev = model.positions.device if model.positions.is_cuda else torch.device("cuda", torch.cuda.current_device())
    f32 = lambda t: t.to(dev).contiguous().to(torch.float32)
    params = [f32(getattr(model, g)) for g in _GROUPS]
    n, c = int(params[0].shape[0]), int(params[1].shape[2])
    state = optimizer is not None
    m = [f32(t) for t in optimizer.m_] if state else None
    v = [f32(t) for t in optimizer.v_] if state else None
    row = lib.cugs_ply_vertex_floats(c, int(state))
    verts = torch.empty((n, row), dtype=torch.float32, device=dev)
    check(lib.cugs_ply_pack(n, c, _ptr_array(params), _ptr_array(m) if state else None,
                            _ptr_array(v) if state else None, _ptr(verts), _stream(dev)), "cugs_ply_pack")
    lines = ["ply", "format binary_little_endian 1.0"]
    if state:
        lines.append("comment cugs_adam_step %d" % int(optimizer.step_count_))
    lines.append("element vertex %d" % n)
    lines += ["property float " + p for p in _property_names(c, state)]
    lines.append("end_header")
    host = verts.cpu().numpy()                      # one device-to-host copy of the finished records
    try:
        with open(os.fspath(path), "wb") as f:
            f.write(("\n".join(lines) + "\n").encode("ascii"))
            f.write(host.astype("<f4", copy=False).tobytes())
    except OSError:
        return False
    return True


def _parse_header(buf: bytes):
    """parse_ply_header (ply_io.cpp:211-250)."""
    pos, lines = 0, []
    while True:
        end = buf.find(b"\n", pos)
        if end < 0:
            raise RuntimeError("Not a PLY file")
        line = buf[pos:end].decode("ascii", "replace").rstrip("\r")
        pos = end + 1
        lines.append(line)
        if line == "end_header":
            break
    if "ply" not in lines[0]:
        raise RuntimeError("Not a PLY file")
    if len(lines) < 2 or "binary_little_endian" not in lines[1]:
        raise RuntimeError("Only binary_little_endian PLY is supported")
    count, names, step = 0, [], None
    for line in lines[2:]:
        tok = line.split()
        if not tok:
            continue
        if tok[0] == "element" and len(tok) >= 3 and tok[1] == "vertex":
            count = int(tok[2])
        elif tok[0] == "property" and len(tok) >= 3:
            names.append(tok[2])
        elif tok[0] == "comment" and len(tok) == 3 and tok[1] == "cugs_adam_step":
            step = int(tok[2])
    return count, names, pos, step


def read_gaussian_ply(path, device=None, return_state: bool = False):
    """ply_io.cpp:258-351.  Raises RuntimeError like the reference (missing file, bad header, short data,
    missing property).  device=None returns the model on the CPU as the reference does; pass a CUDA device
    to keep it where it was unpacked.  return_state: also return {'m': [5 tensors], 'v': [...], 'step': k}
    (or None when the file carries no optimizer state)."""
    try:
        buf = open(os.fspath(path), "rb").read()
    except OSError:
        raise RuntimeError("Failed to open PLY file: " + str(path))
    n, names, off, step = _parse_header(buf)
    index = {nm: i for i, nm in enumerate(names)}
    num_rest = 0
    while "f_rest_%d" % num_rest in index:
        num_rest += 1
    c = 1 + num_rest // 3                                          # :283
    have_state = "m_x" in index and "v_x" in index
    want = _canonical_names(c, have_state and return_state)
    for nm in want:
        if nm not in index:
            raise RuntimeError("Missing PLY property: " + nm)
    num_props = len(names)
    if len(buf) - off < n * num_props * 4:
        raise RuntimeError("Failed to read PLY binary data")
    work = torch.device(device) if device is not None and torch.device(device).type == "cuda" else \
        torch.device("cuda", torch.cuda.current_device())
    data = torch.from_numpy(np.frombuffer(buf, "<f4", n * num_props, off).copy()).to(work)
    col_of = torch.tensor([index[nm] for nm in want], dtype=torch.int32, device=work)
    shapes = ((n, 3), (n, 3, c), (n, 1), (n, 3), (n, 4))
    mk = lambda: [torch.empty(s, dtype=torch.float32, device=work) for s in shapes]
    params = mk()
    with_state = have_state and return_state
    m, v = (mk(), mk()) if with_state else (None, None)
    check(lib.cugs_ply_unpack(n, c, num_props, _ptr(data), _ptr(col_of), _ptr_array(params),
                              _ptr_array(m) if with_state else None, _ptr_array(v) if with_state else None,
                              _stream(work)), "cugs_ply_unpack")
    out_dev = torch.device("cpu") if device is None else torch.device(device)
    model = GaussianModel(**{g: t.to(out_dev) for g, t in zip(_GROUPS, params)})
    if not return_state:
        return model
    state = dict(m=[t.to(out_dev) for t in m], v=[t.to(out_dev) for t in v], step=step or 0) if with_state else None
    return model, state


def restore_optimizer(optimizer, state) -> None:
    """Put a checkpoint's moments and step count back into a FusedAdam built on the loaded model."""
    _torch_check(state is not None, "the checkpoint carries no optimizer state")
    for i in range(5):
        _torch_check(tuple(state["m"][i].shape) == tuple(optimizer.m_[i].shape), "optimizer state shape mismatch")
        optimizer.m_[i] = state["m"][i].to(optimizer.m_[i].device).contiguous()
        optimizer.v_[i] = state["v"][i].to(optimizer.v_[i].device).contiguous()
    optimizer.step_count_ = int(state["step"])
