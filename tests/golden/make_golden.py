#!/usr/bin/env python3
"""Generates tests/golden/*.npz: small input/expected-output vectors for the whole hot path.

Provenance: the reference (Artemarius/cuda-gaussian-splatting) holds NO numeric fixtures for this
path and its kernels cannot run without NVIDIA hardware, so these vectors are
produced by THIS repository's CPU oracle (oracle/cugs_oracle.c) after it was pinned as DESIGN.md §7
describes.  They are data only (inputs and expected outputs); they guard the oracle against
regressions and give the GPU tests a target that does not depend on the oracle library at run time.
Regenerate with:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

CASES = {
    # name: (n, w, h, sh_degree, mu_s, view, bg, seed)
    "small_sh3": (300, 96, 64, 3, -3.2, 0, (0.1, 0.2, 0.3), 101),
    "dense_sh1_rotated": (400, 80, 56, 1, -2.6, 3, (0.0, 0.0, 0.0), 202),     # saturated pixels (Q1), Q12 zero pairs
    "ragged_sh0": (150, 50, 35, 0, -3.0, 0, (1.0, 1.0, 1.0), 303),            # image not a multiple of 16
}


def build(name):
    pkg, orc = ge.load_package(), ge.load_oracle()
    n, w, h, deg, mu_s, view, bg, seed = CASES[name]
    arrays = pkg.scene.make_gaussians(n, w, h, sh_degree=deg, seed=seed, mu_s=mu_s)
    arrays["positions"][:5, 2] = -1.0                          # culled rows
    if name.startswith("dense"):
        arrays["positions"][:, :2] *= 1.7                      # many splats off screen in both axes -> quirk Q12
        arrays["opacities"] += 2.5                             # opaque -> saturated pixels -> quirk Q1
        arrays["scales"] += 1.0
    cam = pkg.scene.make_camera(w, h, view=view)
    K = cam.intrinsics
    g = pkg.scene.make_dl_dcolor(w, h, seed=seed + 1)
    fwd = orc.render(arrays, cam.rotation, cam.translation, K.fx, K.fy, K.cx, K.cy, w, h, bg=bg, active_degree=deg)
    bwd = orc.render_backward(g, fwd, arrays, K.fx, K.fy, K.cx, K.cy, w, h, bg=bg)
    out = {f"in_{k}": v for k, v in arrays.items()}
    out.update(in_dl_dcolor=g, in_rotation=cam.rotation, in_translation=cam.translation,
               in_intrinsics=np.array([K.fx, K.fy, K.cx, K.cy], np.float32), in_size=np.array([w, h, deg], np.int32),
               in_background=np.array(bg, np.float32), in_view=np.int32(view))
    for k in ("means_2d", "depths", "cov_2d_inv", "radii", "tiles_touched", "opacities_act", "rgb", "keys", "values",
              "tile_ranges", "color", "final_T", "n_contrib"):
        out[f"fwd_{k}"] = fwd[k]
    for k in ("dL_dpositions", "dL_drotations", "dL_dscales", "dL_dopacities", "dL_dsh_coeffs", "dL_dmeans_2d"):
        out[f"bwd_{k}"] = bwd[k]
    return out


if __name__ == "__main__":
    for name in CASES:
        data = build(name)
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), f"{name}.npz")
        np.savez_compressed(path, **data)
        print(name, os.path.getsize(path) // 1024, "KiB", "pairs", data["fwd_values"].size,
              "zero-key pairs", int((data["fwd_keys"] == 0).sum()), "saturated px", int((data["fwd_final_T"] < 1 / 255).sum()))
