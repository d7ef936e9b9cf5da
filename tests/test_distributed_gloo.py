"""World-size-2 coverage of the data-parallel exchange step on CPU (gloo): the same
allreduce_gradients() that bench.py runs over RCCL/xGMI, SUM semantics, dL_dmeans_2d untouched."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import __graft_entry__ as ge
    pkg = ge.load_package()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, c = 50, 16
        g = torch.Generator().manual_seed(1000 + rank)
        mk = lambda *s: torch.randn(*s, generator=g)
        grads = pkg.BackwardOutput(mk(n, 3), mk(n, 4), mk(n, 3), mk(n, 1), mk(n, 3, c), mk(n, 2))
        local = {f: getattr(grads, f).clone() for f in pkg.parallel.GRAD_FIELDS + ("dL_dmeans_2d",)}
        works = pkg.parallel.allreduce_gradients(grads, async_op=True)
        pkg.parallel.wait_all(works)
        out = {f: getattr(grads, f).clone() for f in local}
        views = [pkg.parallel.view_for_rank(s, rank, world, 8) for s in range(4)]
        # the collective half of the compact exchange on fresh per-rank data
        g2 = pkg.BackwardOutput(mk(n, 3), mk(n, 4), mk(n, 3), mk(n, 1), None, mk(n, 2))
        gated, centre = mk(n, 3), mk(3)
        mine = dict(pos=g2.dL_dpositions.clone(), rot=g2.dL_drotations.clone(), gated=gated.clone(), centre=centre.clone())
        gviews, centres, pending = pkg.parallel.collect_views(g2, gated, centre)   # geometry reduced in place
        assert pending == []
        compact = dict(sum_pos=g2.dL_dpositions.numpy(), sum_rot=g2.dL_drotations.numpy(), views=gviews.numpy(),
                       centres=centres.numpy(), **{k: v.numpy() for k, v in mine.items()})
        # the early-gather form of the same exchange: the colour gather is started first (under the projection
        # backward on a GPU), the geometry all-reduce and the rebuild follow; views in rank order, centres gathered
        g3 = pkg.BackwardOutput(None, None, None, None, None, mk(n, 2), geom_flat=mk(11 * n))
        flat_mine, gated3 = g3.geom_flat.clone(), mk(n, 3)
        handle = pkg.parallel.begin_colour_gather(gated3, centre)
        seen = {}
        def rebuild(deg, positions, views, centres_, num_coeffs):
            seen.update(deg=deg, views=views.clone(), centres=centres_, num_coeffs=num_coeffs)
            return views.sum(0)
        pkg.parallel.finish_exchange(g3, handle, mk(n, 3), 2, 9, rebuild=rebuild)
        compact.update(e_flat=flat_mine.numpy(), e_sum=g3.geom_flat.numpy(), e_gated=gated3.numpy(),
                       e_views=seen["views"].numpy(), e_centres=np.array(seen["centres"], dtype=np.float32),
                       e_sh=g3.dL_dsh_coeffs.numpy(), e_args=(seen["deg"], seen["num_coeffs"]))
        # N2 statistics: per-rank accumulators (CPU tensors stand in for the device arrays) -> SUM / SUM / MAX
        ctrl = pkg.DensificationController(pkg.DensificationConfig(), 5.0)
        ctrl.grad_accum_, ctrl.grad_count_, ctrl.max_radii_2d_ = mk(n).abs(), torch.full((n,), float(rank + 1)), mk(n).abs() * 10
        compact.update(acc=ctrl.grad_accum_.numpy().copy(), cnt=ctrl.grad_count_.numpy().copy(),
                       rad=ctrl.max_radii_2d_.numpy().copy())
        pkg.parallel.allreduce_densify_stats(ctrl)
        compact.update(acc_out=ctrl.grad_accum_.numpy(), cnt_out=ctrl.grad_count_.numpy(), rad_out=ctrl.max_radii_2d_.numpy())
        # split noise: one draw, identical on every rank; densify() refuses a per-rank draw under DP
        noise = pkg.parallel.shared_split_noise(n, torch.device("cpu"), step=700, seed=3)
        compact.update(noise=noise.numpy().copy())
        model = pkg.GaussianModel(mk(n, 3), mk(n, 3, c), mk(n, 1), mk(n, 4), mk(n, 3))
        try:
            ctrl.densify(model, 700)
            compact.update(refused=False)
        except RuntimeError as e:
            compact.update(refused="replicas" in str(e))
        # numpy: pickled by value (torch tensors would travel as shared-memory handles)
        q.put((rank, {k: v.numpy() for k, v in local.items()}, {k: v.numpy() for k, v in out.items()}, views, compact))
    finally:
        dist.destroy_process_group()


def test_allreduce_gradients_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    (_, l0, o0, v0, c0), (_, l1, o1, v1, c1) = res
    for f in l0:
        if f == "dL_dmeans_2d":                       # per-view statistic: not reduced
            assert np.array_equal(o0[f], l0[f]) and np.array_equal(o1[f], l1[f])
        else:
            assert np.allclose(o0[f], l0[f] + l1[f]) and np.array_equal(o0[f], o1[f])
    assert set(v0).isdisjoint(v1) and v0 == [0, 2, 4, 6] and v1 == [1, 3, 5, 7]
    # compact exchange: views stacked in RANK order on both ranks, geometry summed in place
    for c in (c0, c1):
        assert np.array_equal(c["views"][0], c0["gated"]) and np.array_equal(c["views"][1], c1["gated"])
        assert np.array_equal(c["centres"][0], c0["centre"]) and np.array_equal(c["centres"][1], c1["centre"])
        assert np.allclose(c["sum_pos"], c0["pos"] + c1["pos"]) and np.allclose(c["sum_rot"], c0["rot"] + c1["rot"])
    assert np.array_equal(c0["sum_pos"], c1["sum_pos"])
    for c in (c0, c1):                                 # early gather: same result, whoever started when
        assert np.array_equal(c["e_views"][0], c0["e_gated"]) and np.array_equal(c["e_views"][1], c1["e_gated"])
        assert np.array_equal(c["e_centres"][0], c0["centre"]) and np.array_equal(c["e_centres"][1], c1["centre"])
        assert np.allclose(c["e_sum"], c0["e_flat"] + c1["e_flat"]) and c["e_args"] == (2, 9)
        assert np.array_equal(c["e_sh"], c0["e_gated"] + c1["e_gated"])
    assert np.array_equal(c0["e_sum"], c1["e_sum"])
    assert np.array_equal(c0["noise"], c1["noise"]) and c0["noise"].shape == (2, 50, 3) and c0["noise"].std() > 0.5
    assert c0["refused"] is True and c1["refused"] is True
    for c in (c0, c1):                                 # densification statistics agree on every replica
        assert np.allclose(c["acc_out"], c0["acc"] + c1["acc"]) and np.array_equal(c["cnt_out"], c0["cnt"] + c1["cnt"])
        assert np.array_equal(c["rad_out"], np.maximum(c0["rad"], c1["rad"]))


def test_early_gather_without_process_group(pkg):
    """One process, no group: the "gather" is the rank's own tensor, finish_exchange rebuilds from it."""
    gated, flat = torch.arange(12.0).reshape(4, 3), torch.ones(44)
    grads = pkg.BackwardOutput(None, None, None, None, None, None, geom_flat=flat)
    h = pkg.parallel.begin_colour_gather(gated)
    out = pkg.parallel.finish_exchange(grads, h, torch.zeros(4, 3), 1, 4, all_cam_centers=[[0.0, 0.0, 0.0]],
                                       rebuild=lambda d, p, v, c, k: v[0] * 2)
    assert torch.equal(out.dL_dsh_coeffs, gated * 2) and bool((flat == 1).all())
    try:
        pkg.parallel.finish_exchange(grads, pkg.parallel.begin_colour_gather(gated), torch.zeros(4, 3), 1, 4)
        assert False, "no centres: must raise"
    except ValueError:
        pass


def test_allreduce_is_noop_without_process_group(pkg):
    grads = pkg.BackwardOutput(*(torch.ones(2, k) for k in (3, 4, 3, 1)), torch.ones(2, 3, 1), torch.ones(2, 2))
    assert pkg.parallel.allreduce_gradients(grads) == []
    assert bool((grads.dL_dpositions == 1).all())
