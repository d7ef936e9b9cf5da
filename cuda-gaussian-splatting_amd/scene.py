"""Synthetic scenes for parity tests and bench.py.

Counter-based RNG (numpy's Philox keyed by the seed), no torch involved, so the oracle and the
GPU path consume byte-identical inputs.  Camera conventions are the reference tests'
(tests/test_projection.cpp:25-36): pinhole, principal point at the image centre, identity pose
for view 0.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict

import numpy as np

from .types import CameraInfo, CameraIntrinsics, GaussianModel, sh_coeff_count

SCENE_SEED = 1234
GRAD_SEED = 5678
FOCAL_RATIO = 0.78          # fx = fy = 0.78 * W  (500/640, test_projection.cpp:27-32)


def make_camera(width: int, height: int, view: int = 0) -> CameraInfo:
    """View 0 is the identity pose of every reference test; view k > 0 orbits the scene centre
    (0, 0, 6) by 4 degrees per view about the y axis (distinct training views for the
    data-parallel configuration)."""
    fx = FOCAL_RATIO * width
    cam = CameraInfo(width=width, height=height,
                     intrinsics=CameraIntrinsics(fx=fx, fy=fx, cx=width / 2.0, cy=height / 2.0))
    if view:
        ang = math.radians(4.0 * view)
        c, s = math.cos(ang), math.sin(ang)
        R = np.array([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]], dtype=np.float64)   # world-to-camera
        centre = np.array([0.0, 0.0, 6.0])
        cam.rotation = R.astype(np.float32)
        cam.translation = (centre - R @ centre).astype(np.float32)    # keeps the scene centre fixed
    return cam


def make_gaussians(n: int, width: int, height: int, sh_degree: int = 3, seed: int = SCENE_SEED,
                   mu_s: float = -4.6, z_range=(2.0, 10.0), cluster=None) -> Dict[str, np.ndarray]:
    """Arrays in the reference's layouts (core/gaussian.hpp:36-40), float32.
    `cluster` = (fraction, area): a SKEWED variant of the scene - `fraction` of the Gaussians are drawn inside a
    centred window that covers `area` of the screen (cluster=(0.8, 0.1): 80 % of the splats on 10 % of the image, the
    tile lists there ~36 times as long as elsewhere), the rest uniformly as before.  Real captures look like this; the
    uniform scene of SURVEY 8d does not exercise what one workgroup per tile costs then."""
    rng = np.random.Generator(np.random.Philox(key=seed))
    fx = FOCAL_RATIO * width
    z = rng.uniform(z_range[0], z_range[1], n)
    u = rng.uniform(-1.0, 1.0, n)
    v = rng.uniform(-1.0, 1.0, n)
    if cluster is not None:
        frac, area = float(cluster[0]), float(cluster[1])
        half = math.sqrt(area)                       # window [-half, half]^2 of the [-1, 1]^2 screen square
        inside = rng.uniform(0.0, 1.0, n) < frac
        u = np.where(inside, u * half, u)
        v = np.where(inside, v * half, v)
    x = u * z * (width / 2.0) / fx * 1.05            # ~9% land off screen (exercises SURVEY Q7)
    y = v * z * (height / 2.0) / fx * 1.05
    positions = np.stack([x, y, z], axis=1).astype(np.float32)
    scales = (mu_s + 0.5 * rng.standard_normal((n, 3))).astype(np.float32)
    q = rng.standard_normal((n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    rotations = q.astype(np.float32)
    opacities = rng.standard_normal((n, 1)).astype(np.float32)
    c = sh_coeff_count(sh_degree)
    sh = (0.5 * rng.standard_normal((n, 3, c), dtype=np.float32)).astype(np.float32)
    return dict(positions=positions, sh_coeffs=sh, opacities=opacities, rotations=rotations, scales=scales)


def make_dl_dcolor(width: int, height: int, seed: int = GRAD_SEED) -> np.ndarray:
    rng = np.random.Generator(np.random.Philox(key=seed))
    g = rng.standard_normal((height, width, 3), dtype=np.float32)
    return (g / np.float32(width * height)).astype(np.float32)


def to_model(arrays: Dict[str, np.ndarray], device) -> GaussianModel:
    import torch
    return GaussianModel(**{k: torch.from_numpy(np.ascontiguousarray(v)).to(device) for k, v in arrays.items()})


@dataclass
class Workload:
    """One of BASELINE.json's configs."""
    name: str
    n: int
    width: int
    height: int
    sh_degree: int
    mu_s: float = -4.6


CONFIGS = {
    "config1": Workload("10k/256x256/SH0", 10_000, 256, 256, 0),
    "config2": Workload("100k/1920x1080/SH0", 100_000, 1920, 1080, 0),
    "config3": Workload("1M/1920x1080/SH3", 1_000_000, 1920, 1080, 3),
    "config4": Workload("6M/1600x1063/SH3", 6_000_000, 1600, 1063, 3),
}
