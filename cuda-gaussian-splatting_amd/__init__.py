"""MI355X-native differentiable Gaussian-splat rasterizer + fused Adam.

A drop-in for the hot path of Artemarius/cuda-gaussian-splatting: hand-written HIP for gfx950
behind a C ABI (include/cugs_hip.h, built to libcugs_hip.so), with this package as the thin
host-side mirror of the reference's operator surface (namespace cugs).  Importing it loads the
shared library and raises if it is missing: there is no fallback path.

The directory is named after the reference repository (cuda-gaussian-splatting_amd), which is
not a Python identifier; load it under the module name `cugs_amd` via __graft_entry__.load_package().
"""
from ._lib import CugsError, LIB_PATH, version  # noqa: F401  (loads libcugs_hip.so or raises)
from .types import (BackwardOutput, CameraInfo, CameraIntrinsics, ForwardOutput, GaussianModel,  # noqa: F401
                    PredictionMiss, ProjectionBackwardOutput, ProjectionOutput, RasterizeBackwardOutput, RenderOutput,
                    RenderSettings, SortingOutput, K_MAX_SH_DEGREE, K_TILE_SIZE, sh_coeff_count)
from .rasterizer import (evaluate_sh_backward_cuda, evaluate_sh_cuda, project_backward,  # noqa: F401
                         project_gaussians, rasterize_backward, rasterize_forward, render, render_backward,
                         sh_backward_views, sort_gaussians)
from .fused_adam import (AdamConfig, FusedAdam, ParamGroup, PositionLRConfig,  # noqa: F401
                         active_sh_degree_for_step, lr_defaults, position_lr)
from .loss import combined_loss, combined_loss_and_grad, l1_loss, ssim, ssim_loss  # noqa: F401
from .densification import DensificationConfig, DensificationController, DensificationStats  # noqa: F401
from .ply_io import read_gaussian_ply, restore_optimizer, write_gaussian_ply  # noqa: F401
from .views import StreamedViewCache, ViewCache, image_to_float, load_image_resized, load_image_u8  # noqa: F401
from . import parallel, scene  # noqa: F401
