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


def wait_all(works: Sequence) -> None:
    for w in works:
        w.wait()


def view_for_rank(step: int, rank: int, world_size: int, num_views: int) -> int:
    """Which training view a rank renders at `step`: consecutive views, disjoint across ranks."""
    return (step * world_size + rank) % max(num_views, 1)
