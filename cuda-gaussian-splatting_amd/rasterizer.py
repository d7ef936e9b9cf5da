"""Host-side mirror of the reference's rasterizer operator surface over the C ABI.

Same names, argument meaning and error behaviour as namespace cugs (file:line in the reference's src/):
  project_gaussians          rasterizer/projection.hpp:39      (projection.cu:195-289)
  sort_gaussians             rasterizer/sorting.hpp:41         (sorting.cu:115-227)
  rasterize_forward          rasterizer/forward.hpp:41         (forward.cu:180-240)
  rasterize_backward         rasterizer/backward.hpp:39        (backward.cu:239-306)
  project_backward           rasterizer/projection_backward.hpp:44 (projection_backward.cu:253-344)
  evaluate_sh_cuda           core/sh.hpp:29                    (sh.cu:81-123)
  evaluate_sh_backward_cuda  core/sh_backward.hpp:25           (sh_backward.cu:114-156)
  render / render_backward   rasterizer/rasterizer.hpp:57,88   (rasterizer.cpp:22-186)

PyTorch is plumbing only: it owns device memory and the stream.  Every computation is a call
into libcugs_hip.so; a TORCH_CHECK in the reference is a RuntimeError here.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence

import torch

from . import _lib
from ._lib import lib, check
from .types import (BackwardOutput, CameraInfo, ForwardOutput, GaussianModel, ProjectionBackwardOutput,
                    ProjectionOutput, RasterizeBackwardOutput, RenderOutput, RenderSettings,
                    SortingOutput, K_TILE_SIZE)


def _torch_check(cond: bool, msg: str) -> None:
    if not cond:
        raise RuntimeError(msg)          # c10::Error in the reference


def _ptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None or t.numel() == 0 else t.data_ptr())


def _stream(device: torch.device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _f32c(t: torch.Tensor) -> torch.Tensor:
    """`.contiguous().to(kFloat32)` as the launchers do (projection.cu:240-243)."""
    return t.contiguous().to(torch.float32)


# Host-side state of the sort - scratch, pair-count prediction, held capacity, depth-route bit - is kept per
# (device, STREAM): two renders queued on two streams of one device run concurrently and must not share a workspace
# (cugs_project_forward_keyed writes the sort keys into it) nor steer each other's predictions.  The library itself
# keeps no state; this is the caller-owned scratch of include/cugs_hip.h, one set per stream that renders.
_workspaces: Dict[tuple, torch.Tensor] = {}


def _skey(device: torch.device) -> tuple:
    """(device, current stream of that device): the key of all per-stream host state."""
    return (device, int(torch.cuda.current_stream(device).cuda_stream))


_pinned = {}


class _PinnedSlot:
    """One pinned int64 the sort's scan kernel stores the pair count into.  A slot belongs to ONE sort from its launch
    until the host has read the count: two renders in flight on a device (render(defer_count=True) twice before the
    first render_backward, or an evaluation render between a deferred render and its backward) must not share the
    word - the later sort would overwrite the count the earlier one is still to be judged by."""

    def __init__(self, tensor, pool):
        self.tensor, self._pool = tensor, pool

    def release(self):
        if self._pool is not None:
            pool, self._pool = self._pool, None
            pool.append(self.tensor)

    def __del__(self):                      # a PendingSort dropped without finish(): the word goes back to the pool
        try:
            self.release()
        except Exception:                   # interpreter shutdown
            pass


def _pinned_total(device: torch.device) -> "_PinnedSlot":
    """A free pinned pair-count word of `device` (a new one if all are held by sorts whose count is unread)."""
    pool = _pinned.setdefault(device, [])
    return _PinnedSlot(pool.pop() if pool else torch.zeros(1, dtype=torch.int64).pin_memory(), pool)


def _workspace(device: torch.device, nbytes: int, kind: str = "n") -> torch.Tensor:
    """Caller-owned scratch for the sort: grown on demand, reused across calls (the reference
    allocates CUB temp storage on every call, sorting.cu:198-200).  `kind`: "n" = the N-level buffer
    that carries state from cugs_sort_count_pairs to cugs_sort_pairs, "p" = the pair-level buffer."""
    key = (_skey(device), kind)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


# --------------------------------------------------------------------------------------
# stage functions
# --------------------------------------------------------------------------------------
# (device, main stream) -> the side stream the colour half of a split projection runs on
_side_streams = {}


def _side_stream(device: torch.device) -> torch.cuda.Stream:
    key = _skey(device)
    st = _side_streams.get(key)
    if st is None:
        # lowest priority the device offers: when the colour half (an HBM stream over 192 B/Gaussian) and the sort's
        # small latency-bound kernels compete for a CU, the sort - the frame's critical path - should win
        try:
            least, _greatest = torch.cuda.Stream.priority_range()
        except Exception:
            least = 0
        st = torch.cuda.Stream(device=device, priority=least)
        _side_streams[key] = st
    return st


def project_gaussians(positions: torch.Tensor, rotations: torch.Tensor, scales: torch.Tensor,
                      opacities: torch.Tensor, sh_coeffs: torch.Tensor, camera: CameraInfo,
                      active_sh_degree: int, scale_modifier: float = 1.0,
                      want_colour_gate: bool = True, key_sort: bool = False,
                      colour_on_side_stream: bool = False) -> ProjectionOutput:
    """`colour_on_side_stream` (not in the reference; what render() passes): the projection runs as two launches -
    cugs_project_forward_geometry on the current stream, cugs_project_forward_colour on a low-priority side stream - so
    that the sort, which needs the geometry half only, starts ~50 us earlier and the SH stream (81 % of the projection's
    reads) travels underneath the sort's latency-bound kernels.  ProjectionOutput.colour_ready is then the event the
    reader of rgb / colour_gate / packed must wait for (ProjectionOutput.wait_colour()).  Bit-identical outputs.
    `key_sort` (not in the reference; what render() passes): the kernel also leaves the sort's per-Gaussian depth
    keys and tile rectangles in this device's sort workspace (cugs_project_forward_keyed), and the returned
    ProjectionOutput carries that workspace as `.sort_workspace`; sort_gaussians_predicted(..., keyed_workspace=) then
    skips its first kernel.  The sort must be the next user of the workspace on this stream.
    `want_colour_gate` (not in the reference): also produce ProjectionOutput.colour_gate, the three ReLU gate bits
    per Gaussian the SH backward would otherwise recompute from the coefficients (~7 us per million Gaussians here,
    24 us saved there); a forward-only render does not ask for it."""
    _torch_check(positions.is_cuda, "positions must be on CUDA")
    _torch_check(positions.dim() == 2 and positions.shape[1] == 3, "positions must be [N, 3]")
    n = int(positions.shape[0])
    dev = positions.device
    f = dict(dtype=torch.float32, device=dev)
    i = dict(dtype=torch.int32, device=dev)
    means_2d = torch.empty((n, 2), **f)
    depths = torch.empty((n,), **f)
    cov_2d_inv = torch.empty((n, 3), **f)
    radii = torch.empty((n,), **i)
    tiles_touched = torch.empty((n,), **i)
    opacities_act = torch.empty((n,), **f)
    rgb = torch.empty((n, 3), **f)
    packed = torch.empty((n, _lib.PACKED_STRIDE), **f)
    colour_gate = torch.empty((n,), dtype=torch.uint8, device=dev) if want_colour_gate else None
    if n == 0:
        return ProjectionOutput(means_2d, depths, cov_2d_inv, radii, tiles_touched, rgb, opacities_act, packed,
                                colour_gate)
    _torch_check(sh_coeffs.dim() == 3 and sh_coeffs.shape[1] == 3 and sh_coeffs.shape[0] == n,
                 "sh_coeffs must be [N, 3, C]")
    num_coeffs = int(sh_coeffs.shape[2])
    _torch_check(0 <= active_sh_degree <= 3, f"SH degree must be 0..3, got {active_sh_degree}")
    _torch_check((active_sh_degree + 1) ** 2 <= num_coeffs,
                 f"Need at least {(active_sh_degree + 1) ** 2} coefficients for degree {active_sh_degree}")
    pos_c, rot_c, scl_c, opa_c, sh_c = map(_f32c, (positions, rotations, scales, opacities, sh_coeffs))
    cam = camera.to_abi()
    outs = (_ptr(means_2d), _ptr(depths), _ptr(cov_2d_inv), _ptr(radii), _ptr(tiles_touched), _ptr(opacities_act),
            _ptr(rgb), _ptr(packed), _ptr(colour_gate))
    if colour_on_side_stream:
        ws = _workspace(dev, lib.cugs_sort_workspace_bytes(n), "n") if key_sort else None
        main = torch.cuda.current_stream(dev)
        side = _side_stream(dev)
        start = torch.cuda.Event()
        start.record(main)                       # the inputs (and the recycled output blocks) are ready from here on
        check(lib.cugs_project_forward_geometry(n, _ptr(pos_c), _ptr(rot_c), _ptr(scl_c), _ptr(opa_c), C.byref(cam),
                                                float(scale_modifier), _ptr(means_2d), _ptr(depths), _ptr(cov_2d_inv),
                                                _ptr(radii), _ptr(tiles_touched), _ptr(opacities_act), _ptr(packed),
                                                _ptr(ws), ws.numel() if ws is not None else 0, _stream(dev)),
              "cugs_project_forward_geometry")
        side.wait_event(start)
        check(lib.cugs_project_forward_colour(n, num_coeffs, int(active_sh_degree), _ptr(pos_c), _ptr(sh_c), C.byref(cam),
                                              _ptr(rgb), _ptr(packed), _ptr(colour_gate),
                                              C.c_void_p(side.cuda_stream)), "cugs_project_forward_colour")
        done = torch.cuda.Event()
        done.record(side)
        return ProjectionOutput(means_2d, depths, cov_2d_inv, radii, tiles_touched, rgb, opacities_act, packed,
                                colour_gate, sort_workspace=ws, colour_ready=done)
    if key_sort:
        ws = _workspace(dev, lib.cugs_sort_workspace_bytes(n), "n")
        check(lib.cugs_project_forward_keyed(n, num_coeffs, int(active_sh_degree), _ptr(pos_c), _ptr(rot_c),
                                             _ptr(scl_c), _ptr(opa_c), _ptr(sh_c), C.byref(cam), float(scale_modifier),
                                             *outs, _ptr(ws), ws.numel(), _stream(dev)), "cugs_project_forward_keyed")
        return ProjectionOutput(means_2d, depths, cov_2d_inv, radii, tiles_touched, rgb, opacities_act, packed,
                                colour_gate, sort_workspace=ws)
    check(lib.cugs_project_forward(n, num_coeffs, int(active_sh_degree), _ptr(pos_c), _ptr(rot_c), _ptr(scl_c),
                                   _ptr(opa_c), _ptr(sh_c), C.byref(cam), float(scale_modifier),
                                   *outs, _stream(dev)), "cugs_project_forward")
    return ProjectionOutput(means_2d, depths, cov_2d_inv, radii, tiles_touched, rgb, opacities_act, packed,
                            colour_gate)


def evaluate_sh_cuda(degree: int, sh_coeffs: torch.Tensor, directions: torch.Tensor) -> torch.Tensor:
    _torch_check(0 <= degree <= 3, f"SH degree must be 0..3, got {degree}")
    _torch_check(sh_coeffs.is_cuda, "sh_coeffs must be on CUDA device")
    _torch_check(directions.is_cuda, "directions must be on CUDA device")
    _torch_check(sh_coeffs.dim() == 3 and sh_coeffs.shape[1] == 3, "sh_coeffs must be [N, 3, C]")
    _torch_check(directions.dim() == 2 and directions.shape[1] == 3, "directions must be [N, 3]")
    _torch_check(sh_coeffs.shape[0] == directions.shape[0], "Batch size mismatch")
    num_coeffs = int(sh_coeffs.shape[2])
    _torch_check(num_coeffs >= (degree + 1) ** 2,
                 f"Need at least {(degree + 1) ** 2} coefficients for degree {degree}")
    n = int(sh_coeffs.shape[0])
    coeffs, dirs = _f32c(sh_coeffs), _f32c(directions)
    out = torch.empty((n, 3), dtype=torch.float32, device=sh_coeffs.device)
    if n == 0:
        return out
    check(lib.cugs_evaluate_sh(int(degree), n, num_coeffs, _ptr(coeffs), _ptr(dirs), _ptr(out),
                               _stream(out.device)), "cugs_evaluate_sh")
    return out


def evaluate_sh_backward_cuda(degree: int, sh_coeffs: torch.Tensor, directions: torch.Tensor,
                              dL_dcolor: torch.Tensor) -> torch.Tensor:
    _torch_check(0 <= degree <= 3, f"SH degree must be 0..3, got {degree}")
    _torch_check(sh_coeffs.is_cuda, "sh_coeffs must be on CUDA device")
    _torch_check(directions.is_cuda, "directions must be on CUDA device")
    _torch_check(dL_dcolor.is_cuda, "dL_dcolor must be on CUDA device")
    _torch_check(sh_coeffs.dim() == 3 and sh_coeffs.shape[1] == 3, "sh_coeffs must be [N, 3, C]")
    _torch_check(directions.dim() == 2 and directions.shape[1] == 3, "directions must be [N, 3]")
    _torch_check(dL_dcolor.dim() == 2 and dL_dcolor.shape[1] == 3, "dL_dcolor must be [N, 3]")
    n, num_coeffs = int(sh_coeffs.shape[0]), int(sh_coeffs.shape[2])
    _torch_check(num_coeffs >= (degree + 1) ** 2, "not enough SH coefficients for the degree")
    coeffs, dirs, g = _f32c(sh_coeffs), _f32c(directions), _f32c(dL_dcolor)
    out = torch.empty_like(coeffs)
    if n == 0:
        return out
    check(lib.cugs_evaluate_sh_backward(int(degree), n, num_coeffs, _ptr(coeffs), _ptr(dirs), _ptr(g),
                                        _ptr(out), _stream(out.device)), "cugs_evaluate_sh_backward")
    return out


# The blend kernels hand their workgroups out in the order of an optional [tiles, 4] int32 tensor: the tiles sorted by the
# length of their lists, longest first, as records {tile, first pair, one past the last pair, 0} (cugs_tile_order; render()
# asks the sort for it).  On views whose splats cluster
# the kernels run a quarter shorter, on uniform ones the same; the results do not depend on the order.
TILE_ORDER = True
# ... from this many (predicted) pairs on: making the order is ~8 us of ONE workgroup inside the sort's last kernel, which a
# large frame hides and a small one does not (100 k Gaussians, forward only: 0.182 ms with the order, 0.165 without)
TILE_ORDER_MIN_PAIRS = 2_000_000


def wants_tile_order(dev) -> bool:
    """render()'s rule: the order is worth making for frames of TILE_ORDER_MIN_PAIRS pairs or more (last count seen on
    this stream)."""
    return bool(TILE_ORDER) and _last_pairs.get(_skey(dev), 0) >= TILE_ORDER_MIN_PAIRS


def tile_order_of(tile_ranges: torch.Tensor, img_w: int, img_h: int) -> Optional[torch.Tensor]:
    """The tiles ordered by list length, longest first, from any valid `tile_ranges` (one small launch)."""
    tiles = int(tile_ranges.shape[0])
    if tiles == 0:
        return None
    order = torch.empty((tiles, 4), dtype=torch.int32, device=tile_ranges.device)
    check(lib.cugs_tile_order(int(img_w), int(img_h), _ptr(tile_ranges.contiguous()), _ptr(order), _stream(tile_ranges.device)),
          "cugs_tile_order")
    return order


def sort_gaussians(means_2d: torch.Tensor, depths: torch.Tensor, radii: torch.Tensor,
                   tiles_touched: torch.Tensor, img_w: int, img_h: int, want_keys: bool = True,
                   wide_depth: bool = False, want_tile_order: bool = False) -> SortingOutput:
    """`wide_depth` (not in the reference): the caller knows that this view's depths leave the range of the three-pass
    depth ordering (an earlier sort reported -1): the general route at once (cugs_sort_count_pairs_wide).
    `want_tile_order` (not in the reference): also SortingOutput.tile_order (tile_order_of)."""
    _torch_check(means_2d.is_cuda, "means_2d must be on CUDA")
    n = int(means_2d.shape[0])
    dev = means_2d.device
    ntx = (img_w + K_TILE_SIZE - 1) // K_TILE_SIZE
    nty = (img_h + K_TILE_SIZE - 1) // K_TILE_SIZE
    num_tiles = ntx * nty
    i32 = dict(dtype=torch.int32, device=dev)
    tile_ranges = torch.empty((num_tiles, 2), **i32)
    st = _stream(dev)
    slot = _pinned_total(dev)                   # pinned: the 8-byte read-back is a single DMA
    total = slot.tensor
    total[0] = 0
    tiles_c = tiles_touched.contiguous().to(torch.int32)
    means_c, depths_c, radii_c = means_2d.contiguous(), depths.contiguous(), radii.contiguous()
    ws = _workspace(dev, lib.cugs_sort_workspace_bytes(n), "n")
    if n > 0:
        count = lib.cugs_sort_count_pairs_wide if wide_depth else lib.cugs_sort_count_pairs
        check(count(n, _ptr(means_c), _ptr(depths_c), _ptr(radii_c), _ptr(tiles_c), int(img_w), int(img_h), _ptr(ws),
                    ws.numel(), C.cast(total.data_ptr(), C.POINTER(C.c_int64)), st), "cugs_sort_count_pairs")
    p = int(total[0])
    slot.release()
    keys = torch.empty((p if want_keys else 0,), dtype=torch.int64, device=dev)
    vals = torch.empty((p,), **i32)
    if num_tiles > 0:
        wp = _workspace(dev, lib.cugs_sort_pair_workspace_bytes(p), "p")
        check(lib.cugs_sort_pairs(n, p, _ptr(means_c), _ptr(depths_c), _ptr(radii_c), _ptr(tiles_c), int(img_w),
                                  int(img_h), _ptr(ws), ws.numel(), _ptr(wp), wp.numel(),
                                  _ptr(keys) if want_keys else C.c_void_p(0), _ptr(vals), _ptr(tile_ranges), st),
              "cugs_sort_pairs")
    return SortingOutput(keys, vals, tile_ranges, p,
                         tile_order=tile_order_of(tile_ranges, img_w, img_h) if want_tile_order else None)


# --------------------------------------------------------------------------------------
# Sort without the host round trip (cugs_sort_pairs_predicted): the pair count of the previous frame on a
# device predicts this frame's; the host keeps launching (the forward blend) while the sort runs and reads
# the true count afterwards.  A wrong prediction is detected then and the exact path is taken instead.
# --------------------------------------------------------------------------------------
# device -> running estimate of the pair count: max(count of the last sort, 0.97 x previous estimate).  Training
# walks over views whose counts differ by tens of percent; an estimate that tracks the recent MAXIMUM keeps the
# capacity sufficient (spare capacity costs a few empty workgroups, a miss costs a wasted sort and blend).
_last_pairs = {}
PREDICT_MARGIN = (1.10, 65536)      # capacity = estimate * 1.10 + 64 Ki
# device -> capacity in use.  It is HELD while the estimate drifts below it (down to 80 %) and grown with 5 % to spare:
# a capacity that follows a slowly moving pair count changes the size of the output buffers on every step, and the
# caching allocator answers a stream of slightly different sizes with splits and, now and then, a hipMalloc (a device
# synchronisation: ~1 ms per step at 6 M Gaussians while Adam moves the scene).
_held_capacity = {}
# (device, stream) -> sorts left on the GENERAL depth route.  A sort that reports -1 (a splat that emits pairs lies
# outside [0.2, ~13 107): scenes in millimetres, far backdrops) makes the bit sticky: the following sorts on that
# stream go straight to cugs_sort_pairs_predicted_wide / cugs_sort_count_pairs_wide instead of paying a wasted
# three-pass attempt, a wasted blend and a blocking re-sort on every frame; after WIDE_DEPTH_HOLD sorts the narrow
# route is probed again (one wasted attempt if the view is still wide).
_wide_depth = {}
WIDE_DEPTH_HOLD = 256


def _capacity_for(dev, estimate: int) -> int:
    needed = min(int(estimate * PREDICT_MARGIN[0]) + PREDICT_MARGIN[1], 2147483647)
    key = _skey(dev)
    held = _held_capacity.get(key)
    if held is not None and needed <= held <= needed * 1.25:
        return held
    cap = needed if held is None or needed < held else min(int(needed * 1.05), 2147483647)
    _held_capacity[key] = cap
    return cap


class PendingSort:
    """Result of sort_gaussians_predicted(): tile_ranges / gaussian_values_sorted may be handed to
    rasterize_forward at once; finish() waits for the count and returns (SortingOutput, valid) - if not
    valid the prediction was too small and everything launched on these buffers must be redone with the
    returned (exact) SortingOutput."""

    def __init__(self, args, keys, vals, tile_ranges, capacity, slot, event, state_key, tile_order=None):
        self._args, self._keys, self._vals, self.tile_ranges = args, keys, vals, tile_ranges
        self.capacity, self._slot, self._event, self._key = capacity, slot, event, state_key
        self.gaussian_values_sorted = vals
        self.tile_order = tile_order         # [tiles] int32 or None: valid together with tile_ranges (misses included)
        self.wide_depth_found = False        # finish(): the miss was a depth outside the three-pass range

    def finish(self):
        self._event.synchronize()
        p = int(self._slot.tensor[0])
        self._slot.release()                 # the word is this sort's own until here (_PinnedSlot)
        means_2d, depths, radii, tiles, img_w, img_h, want_keys = self._args
        if p == -1:
            # a depth outside the range of the three-pass depth sort: the general route, now and - sticky - for the
            # following sorts on this stream; the exact count becomes the next prediction, so the NEXT render of such a
            # view neither misses nor re-sorts
            _wide_depth[self._key] = WIDE_DEPTH_HOLD
            self.wide_depth_found = True
            out = sort_gaussians(means_2d, depths, radii, tiles, img_w, img_h, want_keys, wide_depth=True,
                                 want_tile_order=self.tile_order is not None)
            _last_pairs[self._key] = out.total_pairs
            return out, False
        _torch_check(0 <= p <= 2147483647, "pair count exceeds the reference's int indexing")
        prev = _last_pairs.get(self._key, 0)
        _last_pairs[self._key] = p if p > self.capacity else max(p, int(prev * 0.97))
        if p <= self.capacity:
            keys = self._keys[:p] if want_keys else self._keys
            return SortingOutput(keys, self._vals[:p], self.tile_ranges, p, tile_order=self.tile_order), True
        return sort_gaussians(means_2d, depths, radii, tiles, img_w, img_h, want_keys,
                              wide_depth=_wide_depth.get(self._key, 0) > 0,
                              want_tile_order=self.tile_order is not None), False


def sort_gaussians_predicted(means_2d: torch.Tensor, depths: torch.Tensor, radii: torch.Tensor,
                             tiles_touched: torch.Tensor, img_w: int, img_h: int, want_keys: bool = False,
                             keyed_workspace: Optional[torch.Tensor] = None, want_tile_order: bool = False):
    """sort_gaussians with the pair count predicted from the previous call on this device.  Returns a
    PendingSort, or (no prediction yet / empty input) a finished SortingOutput.
    `keyed_workspace`: ProjectionOutput.sort_workspace of project_gaussians(..., key_sort=True) for these very arrays
    and this image size - the sort's key kernel is skipped (cugs_sort_pairs_predicted_keyed)."""
    dev = means_2d.device
    key = _skey(dev)
    n = int(means_2d.shape[0])
    last = _last_pairs.get(key)
    ntx = (img_w + K_TILE_SIZE - 1) // K_TILE_SIZE
    nty = (img_h + K_TILE_SIZE - 1) // K_TILE_SIZE
    wide = _wide_depth.get(key, 0) > 0
    if wide:
        _wide_depth[key] -= 1                # when it reaches 0 the next sort probes the three-pass route again
    if last is None or n == 0 or ntx * nty == 0:
        out = sort_gaussians(means_2d, depths, radii, tiles_touched, img_w, img_h, want_keys, wide_depth=wide,
                             want_tile_order=want_tile_order)
        _last_pairs[key] = out.total_pairs
        return out
    cap = _capacity_for(dev, last)
    i32 = dict(dtype=torch.int32, device=dev)
    tile_ranges = torch.empty((ntx * nty, 2), **i32)
    keys = torch.empty((cap if want_keys else 0,), dtype=torch.int64, device=dev)
    vals = torch.empty((cap,), **i32)
    tiles_c = tiles_touched.contiguous().to(torch.int32)
    means_c, depths_c, radii_c = means_2d.contiguous(), depths.contiguous(), radii.contiguous()
    ws = keyed_workspace if keyed_workspace is not None else _workspace(dev, lib.cugs_sort_workspace_bytes(n), "n")
    wp = _workspace(dev, lib.cugs_sort_pair_workspace_bytes(cap), "p")
    slot = _pinned_total(dev)
    total = slot.tensor
    entry = lib.cugs_sort_pairs_predicted_keyed if keyed_workspace is not None else lib.cugs_sort_pairs_predicted
    if wide:                                 # the general depth route rebuilds its keys from the arrays
        entry = lib.cugs_sort_pairs_predicted_wide
    common = (n, cap, _ptr(means_c), _ptr(depths_c), _ptr(radii_c), _ptr(tiles_c), int(img_w), int(img_h), _ptr(ws),
              ws.numel(), _ptr(wp), wp.numel(), _ptr(keys) if want_keys else C.c_void_p(0), _ptr(vals),
              _ptr(tile_ranges), C.cast(total.data_ptr(), C.POINTER(C.c_int64)))
    tile_order = None
    if want_tile_order and keyed_workspace is not None and not wide:
        tile_order = torch.empty((ntx * nty, 4), **i32)        # written by the sort itself (no extra launch)
        check(lib.cugs_sort_pairs_predicted_keyed_ordered(*common, _ptr(tile_order), _stream(dev)),
              "cugs_sort_pairs_predicted_keyed_ordered")
    else:
        check(entry(*common, _stream(dev)), "cugs_sort_pairs_predicted")
        if want_tile_order:
            tile_order = tile_order_of(tile_ranges, img_w, img_h)
    ev = torch.cuda.Event()
    ev.record()
    return PendingSort((means_c, depths_c, radii_c, tiles_c, img_w, img_h, want_keys), keys, vals, tile_ranges, cap, slot, ev,
                       key, tile_order=tile_order)


def rasterize_forward(means_2d: torch.Tensor, cov_2d_inv: torch.Tensor, rgb: torch.Tensor,
                      opacities: torch.Tensor, tile_ranges: torch.Tensor, gaussian_indices: torch.Tensor,
                      img_w: int, img_h: int, background: Sequence[float],
                      packed: Optional[torch.Tensor] = None, zero_buf: Optional[torch.Tensor] = None,
                      tile_order: Optional[torch.Tensor] = None) -> ForwardOutput:
    """`zero_buf` (optional, not in the reference): a contiguous float32 tensor the launch also fills with zeros -
    the accumulator of the backward blend, cleared for free by this issue-bound kernel (cugs_rasterize_forward_zero).
    `tile_order` (optional, not in the reference): [tiles, 4] int32, the order the workgroups take the tiles in
    (tile_order_of / SortingOutput.tile_order: longest list first); the outputs do not depend on it."""
    _torch_check(means_2d.is_cuda, "means_2d must be on CUDA")
    dev = means_2d.device
    color = torch.empty((img_h, img_w, 3), dtype=torch.float32, device=dev)
    final_T = torch.empty((img_h, img_w), dtype=torch.float32, device=dev)
    n_contrib = torch.empty((img_h, img_w), dtype=torch.int32, device=dev)
    if img_w == 0 or img_h == 0:
        if zero_buf is not None:
            zero_buf.zero_()                 # the promise holds without a blend launch too (render() relies on it)
        return ForwardOutput(color, final_T, n_contrib)
    bg = (C.c_float * 3)(*[float(b) for b in background])
    if zero_buf is not None:
        _torch_check(zero_buf.is_contiguous() and zero_buf.dtype == torch.float32 and zero_buf.numel() % 4 == 0,
                     "zero_buf must be a contiguous float32 tensor of a multiple of four elements")
    if tile_order is not None:
        _torch_check(tile_order.is_contiguous() and tile_order.dtype == torch.int32 and
                     tile_order.numel() == 4 * tile_ranges.shape[0], "tile_order must be a contiguous [tiles, 4] int32 tensor")
        check(lib.cugs_rasterize_forward_ordered(int(img_w), int(img_h), bg, _ptr(tile_ranges.contiguous()),
                                                 _ptr(gaussian_indices.contiguous()), _ptr(means_2d.contiguous()),
                                                 _ptr(cov_2d_inv.contiguous()), _ptr(rgb.contiguous()),
                                                 _ptr(opacities.contiguous()), _ptr(packed), _ptr(color), _ptr(final_T),
                                                 _ptr(n_contrib), _ptr(zero_buf),
                                                 zero_buf.numel() * 4 if zero_buf is not None else 0, _ptr(tile_order),
                                                 _stream(dev)), "cugs_rasterize_forward_ordered")
        return ForwardOutput(color, final_T, n_contrib)
    if zero_buf is not None:
        check(lib.cugs_rasterize_forward_zero(int(img_w), int(img_h), bg, _ptr(tile_ranges.contiguous()),
                                              _ptr(gaussian_indices.contiguous()), _ptr(means_2d.contiguous()),
                                              _ptr(cov_2d_inv.contiguous()), _ptr(rgb.contiguous()),
                                              _ptr(opacities.contiguous()), _ptr(packed), _ptr(color), _ptr(final_T),
                                              _ptr(n_contrib), _ptr(zero_buf), zero_buf.numel() * 4, _stream(dev)),
              "cugs_rasterize_forward_zero")
        return ForwardOutput(color, final_T, n_contrib)
    check(lib.cugs_rasterize_forward(int(img_w), int(img_h), bg, _ptr(tile_ranges.contiguous()),
                                     _ptr(gaussian_indices.contiguous()), _ptr(means_2d.contiguous()),
                                     _ptr(cov_2d_inv.contiguous()), _ptr(rgb.contiguous()),
                                     _ptr(opacities.contiguous()), _ptr(packed), _ptr(color), _ptr(final_T),
                                     _ptr(n_contrib), _stream(dev)), "cugs_rasterize_forward")
    return ForwardOutput(color, final_T, n_contrib)


def rasterize_backward(dL_dcolor: torch.Tensor, means_2d: torch.Tensor, cov_2d_inv: torch.Tensor,
                       rgb: torch.Tensor, opacities: torch.Tensor, tile_ranges: torch.Tensor,
                       gaussian_indices: torch.Tensor, final_T: torch.Tensor, n_contrib: torch.Tensor,
                       img_w: int, img_h: int, background: Sequence[float], n_gaussians: int,
                       packed: Optional[torch.Tensor] = None, unpack: bool = True,
                       zeroed_accum: Optional[torch.Tensor] = None,
                       tile_order: Optional[torch.Tensor] = None) -> RasterizeBackwardOutput:
    """`zeroed_accum` (optional, not in the reference): an [N, 16] accumulator that is already all zeros (cleared by
    rasterize_forward(..., zero_buf=...)): used as is, without the fill.
    `tile_order` (optional, not in the reference): as in rasterize_forward; the sums are the same up to the order of
    the atomic adds."""
    _torch_check(dL_dcolor.is_cuda, "dL_dcolor must be on CUDA")
    dev = dL_dcolor.device
    n = int(n_gaussians)
    f = dict(dtype=torch.float32, device=dev)
    if zeroed_accum is not None:
        _torch_check(tuple(zeroed_accum.shape) == (n, _lib.GRAD_STRIDE) and zeroed_accum.is_contiguous(),
                     "zeroed_accum must be a contiguous [N, 16] float32 tensor")
    accum = zeroed_accum if zeroed_accum is not None else torch.empty((n, _lib.GRAD_STRIDE), **f)
    entry = lib.cugs_rasterize_backward_prezeroed if zeroed_accum is not None else lib.cugs_rasterize_backward
    if unpack:
        d_rgb, d_opa = torch.empty((n, 3), **f), torch.empty((n,), **f)
        d_means, d_cov = torch.empty((n, 2), **f), torch.empty((n, 3), **f)
    else:
        d_rgb = d_opa = d_means = d_cov = None
    if n > 0 and tile_order is not None:
        _torch_check(tile_order.is_contiguous() and tile_order.dtype == torch.int32 and
                     tile_order.numel() == 4 * tile_ranges.shape[0], "tile_order must be a contiguous [tiles, 4] int32 tensor")
        bg = (C.c_float * 3)(*[float(b) for b in background])
        check(lib.cugs_rasterize_backward_ordered(int(img_w), int(img_h), bg, _ptr(tile_ranges.contiguous()),
                                                  _ptr(gaussian_indices.contiguous()), _ptr(means_2d.contiguous()),
                                                  _ptr(cov_2d_inv.contiguous()), _ptr(rgb.contiguous()),
                                                  _ptr(opacities.contiguous()), _ptr(packed),
                                                  _ptr(dL_dcolor.contiguous()), _ptr(final_T.contiguous()),
                                                  _ptr(n_contrib.contiguous()), n, _ptr(accum), _ptr(d_rgb), _ptr(d_opa),
                                                  _ptr(d_means), _ptr(d_cov), 1 if zeroed_accum is not None else 0,
                                                  _ptr(tile_order), _stream(dev)), "cugs_rasterize_backward_ordered")
    elif n > 0:
        bg = (C.c_float * 3)(*[float(b) for b in background])
        check(entry(int(img_w), int(img_h), bg, _ptr(tile_ranges.contiguous()),
                                          _ptr(gaussian_indices.contiguous()), _ptr(means_2d.contiguous()),
                                          _ptr(cov_2d_inv.contiguous()), _ptr(rgb.contiguous()),
                                          _ptr(opacities.contiguous()), _ptr(packed),
                                          _ptr(dL_dcolor.contiguous()), _ptr(final_T.contiguous()),
                                          _ptr(n_contrib.contiguous()), n, _ptr(accum), _ptr(d_rgb), _ptr(d_opa),
                                          _ptr(d_means), _ptr(d_cov), _stream(dev)), "cugs_rasterize_backward")
    return RasterizeBackwardOutput(d_rgb, d_opa, d_means, d_cov, accum)


def project_backward(dL_dmeans_2d: Optional[torch.Tensor], dL_dcov_2d_inv: Optional[torch.Tensor],
                     dL_drgb: Optional[torch.Tensor], dL_dopacity_act: Optional[torch.Tensor],
                     positions: torch.Tensor, rotations: torch.Tensor, scales: torch.Tensor,
                     opacities: torch.Tensor, sh_coeffs: torch.Tensor, radii: torch.Tensor,
                     camera: CameraInfo, active_sh_degree: int, scale_modifier: float = 1.0,
                     grad_accum: Optional[torch.Tensor] = None, colour_gate: Optional[torch.Tensor] = None,
                     dL_dmeans_2d_out: Optional[torch.Tensor] = None,
                     dL_drgb_gated_out: Optional[torch.Tensor] = None,
                     skip_sh_grad: bool = False,
                     geom_flat: Optional[torch.Tensor] = None) -> ProjectionBackwardOutput:
    _torch_check(positions.is_cuda, "positions must be on CUDA")
    n = int(positions.shape[0])
    dev = positions.device
    f = dict(dtype=torch.float32, device=dev)
    if geom_flat is not None:      # one allocation [rot 4N | pos 3N | scl 3N | opa N]: all-reduced in place
        d_rot, d_pos, d_scl, d_opa = geometry_views(geom_flat, n)
    else:
        d_pos, d_rot = torch.empty((n, 3), **f), torch.empty((n, 4), **f)
        d_scl, d_opa = torch.empty((n, 3), **f), torch.empty((n, 1), **f)
    sh_c = _f32c(sh_coeffs)
    d_sh = None if skip_sh_grad else torch.empty_like(sh_c)      # skipped in the data-parallel exchange
    if n == 0:
        return ProjectionBackwardOutput(d_pos, d_rot, d_scl, d_opa, d_sh)
    pos_c, rot_c, scl_c, opa_c = map(_f32c, (positions, rotations, scales, opacities))
    cam = camera.to_abi()
    cont = lambda t: None if t is None else t.contiguous()
    check(lib.cugs_project_backward(n, int(sh_c.shape[2]), int(active_sh_degree), _ptr(pos_c), _ptr(rot_c),
                                    _ptr(scl_c), _ptr(opa_c), _ptr(sh_c), _ptr(radii.contiguous()),
                                    _ptr(cont(colour_gate)), C.byref(cam), float(scale_modifier),
                                    _ptr(cont(grad_accum)), _ptr(cont(dL_dmeans_2d)), _ptr(cont(dL_dcov_2d_inv)),
                                    _ptr(cont(dL_drgb)), _ptr(cont(dL_dopacity_act)), _ptr(d_pos), _ptr(d_rot),
                                    _ptr(d_scl), _ptr(d_opa), _ptr(d_sh), _ptr(dL_dmeans_2d_out),
                                    _ptr(dL_drgb_gated_out), _stream(dev)),
          "cugs_project_backward")
    return ProjectionBackwardOutput(d_pos, d_rot, d_scl, d_opa, d_sh)


def geometry_views(flat: torch.Tensor, n: int):
    """Views (rot [N,4], pos [N,3], scl [N,3], opa [N,1]) into a flat [11N] float32 buffer.  Rotations come
    first so that their rows stay 16-byte aligned for the kernel's float4 stores."""
    _torch_check(flat.numel() == 11 * n and flat.is_contiguous(), "geom_flat must be a contiguous [11*N] tensor")
    return (flat[0:4 * n].view(n, 4), flat[4 * n:7 * n].view(n, 3), flat[7 * n:10 * n].view(n, 3),
            flat[10 * n:11 * n].view(n, 1))


def sh_backward_views(degree: int, positions: torch.Tensor, gated_rgb_views: torch.Tensor,
                      cam_centers, num_coeffs: int) -> torch.Tensor:
    """Sum over views of gated_rgb[v] (x) Y(dir_v) (cugs_sh_backward_views): the SH gradient of a
    multi-view batch from the all-gathered 12 B/Gaussian colour gradients.  gated_rgb_views [V,N,3]."""
    _torch_check(positions.is_cuda and gated_rgb_views.is_cuda, "inputs must be on CUDA")
    v, n = int(gated_rgb_views.shape[0]), int(positions.shape[0])
    _torch_check(gated_rgb_views.shape == (v, n, 3), "gated_rgb_views must be [V, N, 3]")
    pos_c, g_c = _f32c(positions), _f32c(gated_rgb_views)
    out = torch.empty((n, 3, num_coeffs), dtype=torch.float32, device=positions.device)
    if n == 0:
        return out
    cc = (C.c_float * (3 * v))(*[float(x) for c in cam_centers for x in c])
    check(lib.cugs_sh_backward_views(int(degree), n, int(num_coeffs), _ptr(pos_c), v, _ptr(g_c), cc, _ptr(out),
                                     _stream(positions.device)), "cugs_sh_backward_views")
    return out


def gated_colour_grad(grad_accum: torch.Tensor, colour_gate: torch.Tensor,
                      out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[N,3] gated colour gradient from the backward blend's accumulator rows and the projection's gate bits
    (cugs_gated_colour_grad): what project_backward(..., dL_drgb_gated_out=...) writes, available before the
    projection backward has run (data-parallel exchange: the all-gather starts here)."""
    _torch_check(grad_accum.is_cuda and grad_accum.dim() == 2 and grad_accum.shape[1] == _lib.GRAD_STRIDE
                 and grad_accum.is_contiguous() and grad_accum.dtype == torch.float32,
                 "grad_accum must be a contiguous [N, 16] float32 CUDA tensor")
    n = int(grad_accum.shape[0])
    _torch_check(colour_gate is not None and colour_gate.dtype == torch.uint8 and colour_gate.numel() == n,
                 "colour_gate must be the [N] uint8 gate bits of project_gaussians")
    if out is None:
        out = torch.empty((n, 3), dtype=torch.float32, device=grad_accum.device)
    _torch_check(tuple(out.shape) == (n, 3) and out.is_contiguous() and out.dtype == torch.float32,
                 "out must be a contiguous [N, 3] float32 tensor")
    if n > 0:
        check(lib.cugs_gated_colour_grad(n, _ptr(grad_accum), _ptr(colour_gate.contiguous()), _ptr(out),
                                         _stream(grad_accum.device)), "cugs_gated_colour_grad")
    return out


# --------------------------------------------------------------------------------------
# render / render_backward (rasterizer.cpp:22-186)
# --------------------------------------------------------------------------------------
OVERLAP_COLOUR = False     # render(): the projection's colour half on a side stream underneath the sort - measured
                           # in round 3 (profiles/r03_b_colour_overlap_ab.log, r03_c_colour_grid_caps.log): the sort's
                           # latency-bound kernels lose what the overlap saves, so the one-launch projection stays


def render(model: GaussianModel, camera: CameraInfo, settings: RenderSettings, for_backward: bool = True,
           defer_count: bool = False) -> RenderOutput:
    """`for_backward=False` (evaluation, viewer): skips preparing the backward's accumulator (64 B/Gaussian).
    `defer_count=True` (training loops; not in the reference): returns without waiting for the sort's pair count, so
    the caller can queue the loss behind the forward blend before the host blocks; `RenderOutput.wait()` - called by
    render_backward - completes it (`total_pairs`, the trimmed `gaussian_indices`) and raises PredictionMiss if the
    predicted capacity was too small (the view must then be rendered again: the image is invalid)."""
    _torch_check(model.is_valid(), "GaussianModel is not valid")
    _torch_check(model.positions.is_cuda, "GaussianModel must be on CUDA device")
    n = model.num_gaussians()
    dev = model.positions.device
    f = dict(dtype=torch.float32, device=dev)
    i = dict(dtype=torch.int32, device=dev)
    if n == 0:                                           # rasterizer.cpp:36-55
        color = torch.empty((camera.height, camera.width, 3), **f)
        for ch in range(3):
            color[..., ch] = float(settings.background[ch])
        return RenderOutput(color, torch.ones((camera.height, camera.width), **f),
                            torch.zeros((camera.height, camera.width), **i), torch.empty((0, 2), **f),
                            torch.empty((0,), **f), torch.empty((0, 3), **f), torch.empty((0,), **i),
                            torch.empty((0, 3), **f), torch.empty((0,), **f), torch.empty((0,), **i),
                            torch.empty((0, 2), **i))
    active_degree = min(int(settings.active_sh_degree), model.max_sh_degree())
    proj = project_gaussians(model.positions, model.rotations, model.scales, model.opacities, model.sh_coeffs,
                             camera, active_degree, settings.scale_modifier, want_colour_gate=for_backward,
                             key_sort=True,       # the sort below is the next user of this stream's sort workspace
                             colour_on_side_stream=OVERLAP_COLOUR)   # the SH stream travels underneath the sort
    # the backward blend's accumulator, cleared in passing by the forward blend (which leaves HBM idle)
    accum = torch.empty((n, _lib.GRAD_STRIDE), **f) if for_backward else None
    blend = lambda s: rasterize_forward(proj.means_2d, proj.cov_2d_inv, proj.rgb, proj.opacities_act, s.tile_ranges,
                                        s.gaussian_values_sorted, camera.width, camera.height, settings.background,
                                        packed=proj.packed, zero_buf=accum, tile_order=s.tile_order)
    srt = sort_gaussians_predicted(proj.means_2d, proj.depths, proj.radii, proj.tiles_touched, camera.width,
                                   camera.height, want_keys=False, keyed_workspace=proj.sort_workspace,
                                   want_tile_order=wants_tile_order(dev))
    proj.wait_colour()                                   # the blend reads the colour half's outputs
    fwd = blend(srt)                                     # queued behind the sort; the host has not waited yet
    if defer_count and isinstance(srt, PendingSort):
        return RenderOutput(fwd.color, fwd.final_T, fwd.n_contrib, proj.means_2d, proj.depths, proj.cov_2d_inv,
                            proj.radii, proj.rgb, proj.opacities_act, srt.gaussian_values_sorted, srt.tile_ranges,
                            packed=proj.packed, colour_gate=proj.colour_gate, total_pairs=-1, zeroed_accum=accum,
                            pending=srt, tile_order=srt.tile_order)
    if isinstance(srt, PendingSort):
        srt, valid = srt.finish()
        if not valid:                                    # prediction too small (e.g. right after densification)
            fwd = blend(srt)
    return RenderOutput(fwd.color, fwd.final_T, fwd.n_contrib, proj.means_2d, proj.depths, proj.cov_2d_inv,
                        proj.radii, proj.rgb, proj.opacities_act, srt.gaussian_values_sorted, srt.tile_ranges,
                        packed=proj.packed, colour_gate=proj.colour_gate, total_pairs=srt.total_pairs,
                        zeroed_accum=accum, tile_order=srt.tile_order)


def render_backward(dL_dcolor: torch.Tensor, render_out: RenderOutput, model: GaussianModel,
                    camera: CameraInfo, settings: RenderSettings,
                    dL_drgb_gated_out: Optional[torch.Tensor] = None,
                    geom_flat: Optional[torch.Tensor] = None, fused_adam=None, on_gated_ready=None) -> BackwardOutput:
    """`dL_drgb_gated_out` ([N,3], optional, not in the reference): when given, the per-view SH gradient
    is NOT materialised (dL_dsh_coeffs is None) and the gated colour gradient is written there instead,
    for parallel.exchange_gradients() to rebuild the summed SH gradient after the all-gather.
    `on_gated_ready` (callable, optional; needs dL_drgb_gated_out): the gated colour gradient is made right after
    the backward blend (cugs_gated_colour_grad) and the callable is invoked with it BEFORE the projection backward
    is queued - parallel.begin_colour_gather starts the all-gather there, which then travels under that kernel.
    `fused_adam` (a FusedAdam built on `model`, optional, not in the reference): single-GPU training - the
    projection backward applies the optimizer step to the model IN PLACE (cugs_project_backward_adam) and the
    five parameter gradients are never materialised (they are None in the result; dL_dmeans_2d is returned).
    Equivalent, bit for bit, to render_backward + apply_gradients + step."""
    _torch_check(dL_dcolor.is_cuda, "dL_dcolor must be on CUDA device")
    _torch_check(dL_dcolor.dim() == 3 and dL_dcolor.shape[2] == 3, "dL_dcolor must be [H, W, 3]")
    n = model.num_gaussians()
    dev = dL_dcolor.device
    f = dict(dtype=torch.float32, device=dev)
    if n == 0:                                           # rasterizer.cpp:130-139
        return BackwardOutput(torch.zeros((0, 3), **f), torch.zeros((0, 4), **f), torch.zeros((0, 3), **f),
                              torch.zeros((0, 1), **f), torch.zeros_like(model.sh_coeffs),
                              torch.zeros((0, 2), **f))
    active_degree = min(int(settings.active_sh_degree), model.max_sh_degree())
    render_out.wait()                                    # a deferred render: read the pair count now (may raise)
    # the accumulator render() had the forward blend clear is good for ONE backward
    zeroed, render_out.zeroed_accum = getattr(render_out, "zeroed_accum", None), None
    # ... whichever copy of the RenderOutput asks first: the mark travels with the tensor, so a second backward through
    # a copied RenderOutput (which still holds the now dirty rows) gets a fresh, filled accumulator instead
    if zeroed is not None and getattr(zeroed, "_cugs_consumed", False):
        zeroed = None
    if zeroed is not None:
        zeroed._cugs_consumed = True
    rb = rasterize_backward(dL_dcolor, render_out.means_2d, render_out.cov_2d_inv, render_out.rgb,
                            render_out.opacities_act, render_out.tile_ranges, render_out.gaussian_indices,
                            render_out.final_T, render_out.n_contrib, camera.width, camera.height,
                            settings.background, n, packed=render_out.packed, unpack=False, zeroed_accum=zeroed,
                            tile_order=getattr(render_out, "tile_order", None))
    d_means_2d = torch.empty((n, 2), **f)
    if fused_adam is not None:
        _torch_check(fused_adam.model_ is model, "fused_adam must have been built on this model")
        _torch_check(dL_drgb_gated_out is None and geom_flat is None,
                     "the fused optimizer step is for single-GPU training (no gradient exchange)")
        _torch_check(render_out.colour_gate is not None, "the fused optimizer step needs render()'s colour_gate")
        adam = fused_adam.begin_fused_step()
        cam = camera.to_abi()
        check(lib.cugs_project_backward_adam(n, int(model.sh_coeffs.shape[2]), active_degree, _ptr(model.positions),
                                             _ptr(model.rotations), _ptr(model.scales), _ptr(model.opacities),
                                             _ptr(model.sh_coeffs), _ptr(render_out.radii.contiguous()),
                                             _ptr(render_out.colour_gate.contiguous()), C.byref(cam),
                                             float(settings.scale_modifier), _ptr(rb.grad_accum), C.byref(adam),
                                             _ptr(d_means_2d), _stream(dev)), "cugs_project_backward_adam")
        return BackwardOutput(None, None, None, None, None, d_means_2d)
    gated_by_projection = dL_drgb_gated_out
    if on_gated_ready is not None:
        _torch_check(dL_drgb_gated_out is not None, "on_gated_ready needs dL_drgb_gated_out")
        _torch_check(render_out.colour_gate is not None, "on_gated_ready needs render()'s colour_gate")
        on_gated_ready(gated_colour_grad(rb.grad_accum, render_out.colour_gate, out=dL_drgb_gated_out))
        gated_by_projection = None                       # already made: the projection backward is geometry only
    pb = project_backward(None, None, None, None, model.positions, model.rotations, model.scales,
                          model.opacities, model.sh_coeffs, render_out.radii, camera, active_degree,
                          settings.scale_modifier, grad_accum=rb.grad_accum, colour_gate=render_out.colour_gate,
                          dL_dmeans_2d_out=d_means_2d, dL_drgb_gated_out=gated_by_projection,
                          skip_sh_grad=dL_drgb_gated_out is not None, geom_flat=geom_flat)
    return BackwardOutput(pb.dL_dpositions, pb.dL_drotations, pb.dL_dscales, pb.dL_dopacities,
                          pb.dL_dsh_coeffs, d_means_2d, geom_flat=geom_flat)
