"""Data-parallel training views: one process per GPU, one camera view per rank, parameters
replicated, and ONE exchange step per iteration - the sum-all-reduce of the five parameter
gradients.  The reference has no distributed code; this is the MI355X-native
extension the north star asks for: torch.distributed with backend "nccl" (= RCCL over xGMI) on
GPUs, "gloo" on CPU for tests.

Gradients are SUMMED (not averaged): N ranks x 1 view equals N sequential render_backward calls
accumulated before one FusedAdam step.  dL_dmeans_2d is a per-view densification statistic
(densification.cpp:71-87) and is not reduced.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

from .types import BackwardOutput

# ParamGroup order (lr_schedule.hpp:23-29); floats per Gaussian for C coefficients: 3 + 3C + 1 + 3 + 4
GRAD_FIELDS = ("dL_dpositions", "dL_dsh_coeffs", "dL_dopacities", "dL_dscales", "dL_drotations")


def grad_tensors(grads: BackwardOutput) -> List[torch.Tensor]:
    return [getattr(grads, f) for f in GRAD_FIELDS]


def allreduce_gradients(grads: BackwardOutput, group: Optional[dist.ProcessGroup] = None,
                        async_op: bool = False):
    """In-place SUM all-reduce of the five gradient tensors.  Each tensor is one collective (the
    SH gradient is 81% of the bytes: 192 of 236 B/Gaussian at degree 3), issued back to back so
    RCCL can pipeline them; returns the work handles when async_op."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return []
    works = [dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True) for t in grad_tensors(grads)]
    if async_op:
        return works
    for w in works:
        w.wait()
    return []


# --------------------------------------------------------------------------------------
# Compact exchange: 161 MB moved per rank instead of 413 MB at 1 M Gaussians / SH 3 / 8 ranks.
#   * geometry gradients (positions 3, opacities 1, scales 3, rotations 4 = 44 B/Gaussian): ONE flat
#     SUM all-reduce;
#   * SH gradients (12C B/Gaussian, 81% of the bytes): NOT exchanged.  dL_dsh = gated_rgb_grad (x) Y(dir)
#     is an outer product, so each rank all-gathers the 12 B/Gaussian gated colour gradients (and the 12-byte
#     camera centres) and rebuilds the summed SH gradient locally with cugs_sh_backward_views, in view
#     (= rank) order: the result is bit-identical on every rank, which an all-reduce does not promise.
# --------------------------------------------------------------------------------------
def collect_views(grads: BackwardOutput, gated_rgb: torch.Tensor, cam_center: torch.Tensor,
                  group: Optional[dist.ProcessGroup] = None, need_centres: bool = True, defer_geometry: bool = False):
    """The collective half of the compact exchange (device-agnostic: RCCL on GPUs, gloo on CPU).
    The geometry gradients are all-reduced IN PLACE: as one collective when they are views of one flat
    buffer (render_backward(..., geom_flat=...)), else one collective each.
    Returns (gated_views [V,N,3], centres [V,3] or None, pending): with defer_geometry the geometry
    all-reduce is still in flight on return and `pending` holds its work handles (wait_all them before
    reading the geometry gradients), so the caller can rebuild the SH gradient underneath it."""
    n = gated_rgb.shape[0]
    if not dist.is_initialized():
        return gated_rgb.reshape(1, n, 3), (cam_center.reshape(1, 3) if cam_center is not None else None), []
    world = dist.get_world_size(group)          # a 1-rank group still goes through the collectives
    # outputs are the rank-order concatenation along dim 0 (the shape both RCCL and gloo accept)
    views = torch.empty((world * n, 3), dtype=gated_rgb.dtype, device=gated_rgb.device)
    geom = [grads.geom_flat] if grads.geom_flat is not None else \
        [grads.dL_dpositions, grads.dL_dopacities, grads.dL_dscales, grads.dL_drotations]
    # issue order = execution order on the communicator's stream: what the SH rebuild needs goes first
    first = [dist.all_gather_into_tensor(views, gated_rgb.reshape(n, 3).contiguous(), group=group, async_op=True)]
    centres = None
    if need_centres:
        centres = torch.empty((world * 3,), dtype=cam_center.dtype, device=cam_center.device)
        first.append(dist.all_gather_into_tensor(centres, cam_center.reshape(3).contiguous(), group=group,
                                                 async_op=True))
    pending = [dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True) for t in geom]
    wait_all(first)
    if not defer_geometry:
        wait_all(pending)
        pending = []
    return views.view(world, n, 3), (centres.view(world, 3) if centres is not None else None), pending


def exchange_gradients(grads: BackwardOutput, gated_rgb: torch.Tensor, positions: torch.Tensor,
                       cam_center, active_sh_degree: int, num_coeffs: int,
                       group: Optional[dist.ProcessGroup] = None, all_cam_centers=None) -> BackwardOutput:
    """Compact data-parallel exchange.  `grads` comes from
    render_backward(..., dL_drgb_gated_out=gated_rgb[, geom_flat=...]) (its dL_dsh_coeffs is None).
    On return the four geometry gradients hold the sums over all ranks' views (in place) and
    dL_dsh_coeffs is the summed SH gradient; dL_dmeans_2d stays per-view.  `all_cam_centers` ([V,3] host
    values, if every rank knows every view's camera) saves the gather of the centres and its device-to-host
    read.  The SH rebuild runs while the geometry all-reduce is still on the wire."""
    from .rasterizer import sh_backward_views
    # the device copy of this rank's centre is only needed when the centres have to be gathered (a pageable
    # host-to-device copy would stall the launching thread behind the backward kernels)
    cc = torch.as_tensor(cam_center, dtype=torch.float32, device=gated_rgb.device).reshape(3) \
        if all_cam_centers is None else None
    views, centres, pending = collect_views(grads, gated_rgb, cc, group, need_centres=all_cam_centers is None,
                                            defer_geometry=True)
    host_centres = all_cam_centers if all_cam_centers is not None else centres.cpu().tolist()
    grads.dL_dsh_coeffs = sh_backward_views(active_sh_degree, positions, views, host_centres, num_coeffs)
    wait_all(pending)
    return grads


# --------------------------------------------------------------------------------------
# Compact exchange with the colour gather started EARLY: the gated colour gradient is complete when the backward
# blend is (it is the blend's dL_drgb times the projection's gate bits, cugs_gated_colour_grad), one kernel
# before the geometry gradients exist.  begin_colour_gather() puts the all-gather - half of the exchange's bytes -
# on the wire at that point (render_backward(..., on_gated_ready=...)), so it travels under k_project_backward;
# finish_exchange() then all-reduces the geometry gradients and rebuilds the SH gradient as exchange_gradients does.
# Same collectives in the same order on the communicator, same bits.
# --------------------------------------------------------------------------------------
class ColourGather:
    """Handle of a colour gather in flight: `views` [V*N,3] (rank order), `centres` [V*3] or None, work handles."""

    def __init__(self, views, centres, works, world, n):
        self.views, self.centres, self.works, self.world, self.n = views, centres, works, world, n


def begin_colour_gather(gated_rgb: torch.Tensor, cam_center: Optional[torch.Tensor] = None,
                        group: Optional[dist.ProcessGroup] = None) -> ColourGather:
    """Starts the all-gather of this rank's [N,3] gated colour gradient (and of its camera centre when given) and
    returns at once.  Without a process group the "gather" is the tensor itself."""
    n = int(gated_rgb.shape[0])
    if not dist.is_initialized():
        return ColourGather(gated_rgb.reshape(n, 3), cam_center.reshape(3) if cam_center is not None else None, [], 1, n)
    world = dist.get_world_size(group)
    views = torch.empty((world * n, 3), dtype=gated_rgb.dtype, device=gated_rgb.device)
    works = [dist.all_gather_into_tensor(views, gated_rgb.reshape(n, 3).contiguous(), group=group, async_op=True)]
    centres = None
    if cam_center is not None:
        centres = torch.empty((world * 3,), dtype=cam_center.dtype, device=cam_center.device)
        works.append(dist.all_gather_into_tensor(centres, cam_center.reshape(3).contiguous(), group=group, async_op=True))
    return ColourGather(views, centres, works, world, n)


def finish_exchange(grads: BackwardOutput, gather: ColourGather, positions: torch.Tensor, active_sh_degree: int,
                    num_coeffs: int, group: Optional[dist.ProcessGroup] = None, all_cam_centers=None,
                    rebuild=None) -> BackwardOutput:
    """Second half of the early-gather exchange: SUM all-reduce of the geometry gradients (in place; one collective
    when they are views of `grads.geom_flat`), the SH gradient rebuilt from the gathered views underneath it.
    `all_cam_centers` ([V,3] host values) or the centres gathered by begin_colour_gather(cam_center=...) supply the
    view directions.  `rebuild` (tests on CPU): replaces rasterizer.sh_backward_views."""
    geom = [grads.geom_flat] if grads.geom_flat is not None else \
        [grads.dL_dpositions, grads.dL_dopacities, grads.dL_dscales, grads.dL_drotations]
    pending = []
    if dist.is_initialized():
        pending = [dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True) for t in geom]
    wait_all(gather.works)
    if all_cam_centers is not None:
        host_centres = all_cam_centers
    else:
        if gather.centres is None:
            raise ValueError("finish_exchange: no camera centres (pass all_cam_centers, or cam_center to begin_colour_gather)")
        host_centres = gather.centres.view(gather.world, 3).cpu().tolist()
    if rebuild is None:
        from .rasterizer import sh_backward_views as rebuild
    grads.dL_dsh_coeffs = rebuild(active_sh_degree, positions, gather.views.view(gather.world, gather.n, 3),
                                  host_centres, num_coeffs)
    wait_all(pending)
    return grads


def allreduce_densify_stats(controller, group: Optional[dist.ProcessGroup] = None) -> None:
    """SURVEY §8f N2 under data parallelism: every rank accumulates the densification statistics of its own
    views (DensificationController.accumulate_gradients); before densify() the replicas must agree, so the
    gradient-norm sums and observation counts are SUM-reduced (one collective over a [2, N] buffer) and the
    screen radii MAX-reduced.  Equivalent to one controller having seen all views (densification.cpp:59-88)."""
    if not dist.is_initialized() or controller.grad_accum_ is None:
        return
    both = torch.stack([controller.grad_accum_, controller.grad_count_])
    w0 = dist.all_reduce(both, op=dist.ReduceOp.SUM, group=group, async_op=True)
    w1 = dist.all_reduce(controller.max_radii_2d_, op=dist.ReduceOp.MAX, group=group, async_op=True)
    w0.wait(); w1.wait()
    controller.grad_accum_, controller.grad_count_ = both[0].contiguous(), both[1].contiguous()


def shared_split_noise(n: int, device, step: int, seed: int = 0, group: Optional[dist.ProcessGroup] = None,
                       src: int = 0) -> torch.Tensor:
    """The [2, N, 3] standard-normal split noise of DensificationController.densify, identical on every rank:
    drawn ONCE, on rank `src`, from a CPU generator keyed by (seed, step) and broadcast (no reliance on the ranks'
    device generators being in the same state).  Without a process group: the same draw, locally."""
    gen = torch.Generator(device="cpu").manual_seed((int(seed) * 1_000_003 + int(step)) & 0x7FFFFFFFFFFFFFFF)
    multi = dist.is_initialized() and dist.get_world_size(group) > 1
    if not multi or dist.get_rank(group) == src:
        noise = torch.randn((2, int(n), 3), dtype=torch.float32, generator=gen).to(device)
    else:
        noise = torch.empty((2, int(n), 3), dtype=torch.float32, device=device)
    if multi:
        dist.broadcast(noise, src=dist.get_global_rank(group, src) if group is not None else src, group=group)
    return noise


def densify_replicated(controller, model, step: int, optimizer=None, seed: int = 0,
                       group: Optional[dist.ProcessGroup] = None):
    """densify() for replicated models: the statistics are made to agree (allreduce_densify_stats), the split
    noise is one shared draw, so every rank performs the identical surgery and the replicas stay bit-equal."""
    allreduce_densify_stats(controller, group)
    noise = shared_split_noise(model.num_gaussians(), model.positions.device, step, seed, group)
    return controller.densify(model, step, noise=noise, optimizer=optimizer)


def wait_all(works: Sequence) -> None:
    for w in works:
        w.wait()


def view_for_rank(step: int, rank: int, world_size: int, num_views: int) -> int:
    """Which training view a rank renders at `step`: consecutive views, disjoint across ranks."""
    return (step * world_size + rank) % max(num_views, 1)
