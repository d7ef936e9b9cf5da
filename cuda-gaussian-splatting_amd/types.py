"""Boundary types of the hot path, mirroring the reference's structs field for field.

GaussianModel  <- src/core/gaussian.hpp:34-102
CameraInfo     <- src/core/types.hpp:78-109 (+ CameraIntrinsics)
RenderSettings <- src/rasterizer/rasterizer.hpp:17-21
Output structs <- rasterizer.hpp:27-46,65-72; projection.hpp, sorting.hpp:18-24, forward.hpp, backward.hpp
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import torch

from . import _lib

K_MAX_SH_DEGREE = 3          # gaussian.hpp:13
K_TILE_SIZE = _lib.TILE      # sorting.hpp:16


def sh_coeff_count(degree: int) -> int:
    """gaussian.hpp:16-18"""
    return (degree + 1) * (degree + 1)


@dataclass
class GaussianModel:
    """SoA parameter store (gaussian.hpp:34-102): positions [N,3], sh_coeffs [N,3,C]
    (channel-major), opacities [N,1] (logit), rotations [N,4] (wxyz), scales [N,3] (log)."""
    positions: Optional[torch.Tensor] = None
    sh_coeffs: Optional[torch.Tensor] = None
    opacities: Optional[torch.Tensor] = None
    rotations: Optional[torch.Tensor] = None
    scales: Optional[torch.Tensor] = None

    def num_gaussians(self) -> int:
        return int(self.positions.shape[0]) if self.positions is not None else 0

    def max_sh_degree(self) -> int:
        """gaussian.hpp:47-54: D = int(sqrt(float(C))) - 1"""
        if self.sh_coeffs is None:
            return 0
        return int(math.sqrt(float(self.sh_coeffs.shape[2]))) - 1

    def to_device(self, device) -> None:
        for name in ("positions", "sh_coeffs", "opacities", "rotations", "scales"):
            t = getattr(self, name)
            if t is not None:
                setattr(self, name, t.to(device))

    def is_valid(self) -> bool:
        """gaussian.hpp:68-88"""
        p = self.positions
        if p is None or p.dim() != 2 or p.shape[1] != 3:
            return False
        n = p.shape[0]
        s, o, r, c = self.sh_coeffs, self.opacities, self.rotations, self.scales
        if s is None or s.dim() != 3 or s.shape[0] != n or s.shape[1] != 3:
            return False
        if o is None or o.dim() != 2 or o.shape[0] != n or o.shape[1] != 1:
            return False
        if r is None or r.dim() != 2 or r.shape[0] != n or r.shape[1] != 4:
            return False
        if c is None or c.dim() != 2 or c.shape[0] != n or c.shape[1] != 3:
            return False
        return all(t.device == p.device for t in (s, o, r, c))


@dataclass
class CameraIntrinsics:
    fx: float = 0.0
    fy: float = 0.0
    cx: float = 0.0
    cy: float = 0.0


@dataclass
class CameraInfo:
    """types.hpp:78-109.  rotation/translation are world-to-camera, float32."""
    width: int = 0
    height: int = 0
    intrinsics: CameraIntrinsics = field(default_factory=CameraIntrinsics)
    rotation: np.ndarray = field(default_factory=lambda: np.eye(3, dtype=np.float32))
    translation: np.ndarray = field(default_factory=lambda: np.zeros(3, dtype=np.float32))
    image_id: int = 0
    camera_id: int = 0

    def camera_center(self) -> np.ndarray:
        """C = -R^T t (types.hpp:98-100), evaluated in float32 like Eigen's Matrix3f."""
        R = np.asarray(self.rotation, dtype=np.float32)
        t = np.asarray(self.translation, dtype=np.float32)
        return (-(R.T @ t)).astype(np.float32)

    def world_to_camera(self) -> np.ndarray:
        """types.hpp:103-108"""
        m = np.eye(4, dtype=np.float32)
        m[:3, :3] = np.asarray(self.rotation, dtype=np.float32)
        m[:3, 3] = np.asarray(self.translation, dtype=np.float32)
        return m

    def to_abi(self) -> "_lib.Camera":
        """The POD the adapter hands to the C ABI: row-major view[16] (projection.cu:228-233),
        pinhole intrinsics, image size, camera centre (projection.cu:273-275)."""
        cam = _lib.Camera()
        w2c = self.world_to_camera().reshape(-1)
        for i in range(16):
            cam.view[i] = float(w2c[i])
        cam.fx, cam.fy = float(self.intrinsics.fx), float(self.intrinsics.fy)
        cam.cx, cam.cy = float(self.intrinsics.cx), float(self.intrinsics.cy)
        cam.width, cam.height = int(self.width), int(self.height)
        cc = self.camera_center()
        for i in range(3):
            cam.cam_center[i] = float(cc[i])
        return cam


@dataclass
class RenderSettings:
    """rasterizer.hpp:17-21"""
    background: List[float] = field(default_factory=lambda: [0.0, 0.0, 0.0])
    active_sh_degree: int = 3
    scale_modifier: float = 1.0


@dataclass
class ProjectionOutput:
    means_2d: torch.Tensor
    depths: torch.Tensor
    cov_2d_inv: torch.Tensor
    radii: torch.Tensor
    tiles_touched: torch.Tensor
    rgb: torch.Tensor
    opacities_act: torch.Tensor
    packed: Optional[torch.Tensor] = None      # [N,12] scratch for the blend kernels (not in the reference)
    colour_gate: Optional[torch.Tensor] = None # [N] uint8: the SH backward's ReLU gate bits (not in the reference)
    sort_workspace: Optional[torch.Tensor] = None  # project_gaussians(key_sort=True): the sort workspace it has keyed
    # project_gaussians(colour_on_side_stream=True): rgb, colour_gate and the colour words of `packed` are being written
    # on a side stream; whoever reads them must first make its stream wait for this event (wait_colour does)
    colour_ready: Optional[object] = None

    def wait_colour(self) -> "ProjectionOutput":
        """Make the CURRENT stream wait for the colour half (no-op for a projection made in one launch)."""
        if self.colour_ready is not None:
            torch.cuda.current_stream(self.means_2d.device).wait_event(self.colour_ready)
            self.colour_ready = None
        return self


@dataclass
class SortingOutput:
    """sorting.hpp:18-24"""
    gaussian_keys_sorted: torch.Tensor      # [P] int64 holding the uint64 keys
    gaussian_values_sorted: torch.Tensor    # [P] int32
    tile_ranges: torch.Tensor               # [tiles,2] int32
    total_pairs: int
    tile_order: Optional[torch.Tensor] = None   # [tiles,4] int32 (not in the reference): {tile, first, end, 0}, longest list first


@dataclass
class ForwardOutput:
    color: torch.Tensor
    final_T: torch.Tensor
    n_contrib: torch.Tensor


@dataclass
class RasterizeBackwardOutput:
    dL_drgb: torch.Tensor
    dL_dopacity_act: torch.Tensor
    dL_dmeans_2d: torch.Tensor
    dL_dcov_2d_inv: torch.Tensor
    grad_accum: Optional[torch.Tensor] = None   # [N,16] packed rows (not in the reference)


@dataclass
class ProjectionBackwardOutput:
    dL_dpositions: torch.Tensor
    dL_drotations: torch.Tensor
    dL_dscales: torch.Tensor
    dL_dopacities: torch.Tensor
    dL_dsh_coeffs: torch.Tensor


@dataclass
class RenderOutput:
    """rasterizer.hpp:27-46"""
    color: torch.Tensor
    final_T: torch.Tensor
    n_contrib: torch.Tensor
    means_2d: torch.Tensor
    depths: torch.Tensor
    cov_2d_inv: torch.Tensor
    radii: torch.Tensor
    rgb: torch.Tensor
    opacities_act: torch.Tensor
    gaussian_indices: torch.Tensor
    tile_ranges: torch.Tensor
    packed: Optional[torch.Tensor] = None       # scratch kept alive for render_backward
    colour_gate: Optional[torch.Tensor] = None  # [N] uint8 ReLU gate bits of the SH backward, from the projection
    total_pairs: int = 0
    zeroed_accum: Optional[torch.Tensor] = None # [N,16] accumulator already cleared by the forward blend (one backward)
    pending: Optional[object] = None            # render(..., defer_count=True): the sort's pair count has not been read yet
    tile_order: Optional[torch.Tensor] = None   # [tiles,4] int32: the order the blend kernels' workgroups take the tiles in

    def wait(self) -> "RenderOutput":
        """Completes a render(..., defer_count=True): waits for the sort's pair count, trims `gaussian_indices` to it
        and fills in `total_pairs`.  Raises PredictionMiss if the predicted capacity was too small - everything
        computed from this output (the image included) is then invalid and the view must be rendered again."""
        if self.pending is not None:
            pending, self.pending = self.pending, None
            srt, valid = pending.finish()
            if not valid:
                if getattr(pending, "wide_depth_found", False):
                    raise PredictionMiss("render(defer_count=True): a splat's depth lies outside the range of the fast "
                                         "depth ordering; render this view again - the sorts on this stream now take the "
                                         "general route and the pair count is known, so the next render completes")
                raise PredictionMiss("render(defer_count=True): the predicted pair capacity was too small; "
                                     "render this view again (the next prediction is already corrected)")
            self.gaussian_indices, self.total_pairs = srt.gaussian_values_sorted, srt.total_pairs
        return self


class PredictionMiss(RuntimeError):
    """A deferred render ran on a pair capacity that turned out too small (first frames of a new view set, right
    after a densification): its outputs are invalid.  Render again without defer_count (or simply again)."""


@dataclass
class BackwardOutput:
    """rasterizer.hpp:65-72"""
    dL_dpositions: torch.Tensor
    dL_drotations: torch.Tensor
    dL_dscales: torch.Tensor
    dL_dopacities: torch.Tensor
    dL_dsh_coeffs: torch.Tensor
    dL_dmeans_2d: torch.Tensor
    geom_flat: Optional[torch.Tensor] = None    # [11N] buffer the four geometry gradients are views of (DP exchange)
