"""SURVEY §8(e) with two REAL ranks on the GPU box: two processes, one training view each, both rendering on the
one GPU of the box, the compact exchange between them (all-gather of the gated colour gradients and camera
centres + in-place all-reduce of the flat geometry gradients; gloo carries the bytes here - RCCL needs one GPU
per rank - through the same parallel.collect_views code path), the SH rebuild on each rank.  Every rank must end
up with the sum of the two views' gradients, bit-identical SH gradients on both ranks; then both replicas densify
(statistics reduced, one shared noise draw) and must stay bit-equal."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, W, H, DEG = 12000, 480, 270, 3


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import __graft_entry__ as ge
    pkg = ge.load_package()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        arrays = pkg.scene.make_gaussians(N, W, H, sh_degree=DEG, seed=21, mu_s=-3.9)
        model = pkg.scene.to_model(arrays, dev)
        settings = pkg.RenderSettings(active_sh_degree=DEG)
        cam = pkg.scene.make_camera(W, H, view=pkg.parallel.view_for_rank(0, rank, world, 8))
        g = torch.from_numpy(pkg.scene.make_dl_dcolor(W, H, seed=100 + rank)).to(dev)
        out = pkg.render(model, cam, settings)
        gated = torch.empty((N, 3), device=dev)
        flat = torch.empty((11 * N,), device=dev)
        grads = pkg.render_backward(g, out, model, cam, settings, dL_drgb_gated_out=gated, geom_flat=flat)
        # the collective half on host copies (gloo); the geometry views of `flat` are summed in place
        flat_cpu = flat.cpu()
        host = pkg.BackwardOutput(None, None, None, None, None, None, geom_flat=flat_cpu)
        centre = torch.tensor(cam.camera_center().tolist(), dtype=torch.float32)
        views, centres, pending = pkg.parallel.collect_views(host, gated.cpu(), centre)
        assert pending == []
        flat.copy_(flat_cpu)                                   # the summed geometry gradients back on the device
        d_sh = pkg.sh_backward_views(DEG, model.positions, views.to(dev), centres.tolist(), int(model.sh_coeffs.shape[2]))
        # N2 under data parallelism: per-view statistics -> SUM/SUM/MAX over the ranks (host copies: gloo), ONE
        # shared draw of the split noise, then the identical surgery on every replica
        ctrl = pkg.DensificationController(pkg.DensificationConfig(grad_threshold=2e-8, percent_dense=0.004), 5.0)
        ctrl.accumulate_gradients(grads.dL_dmeans_2d, out.radii)
        host_ctrl = pkg.DensificationController(ctrl.config_, 5.0)
        host_ctrl.grad_accum_, host_ctrl.grad_count_, host_ctrl.max_radii_2d_ = (
            ctrl.grad_accum_.cpu(), ctrl.grad_count_.cpu(), ctrl.max_radii_2d_.cpu())
        pkg.parallel.allreduce_densify_stats(host_ctrl)
        ctrl.grad_accum_, ctrl.grad_count_, ctrl.max_radii_2d_ = (
            host_ctrl.grad_accum_.to(dev), host_ctrl.grad_count_.to(dev), host_ctrl.max_radii_2d_.to(dev))
        refused = False
        try:
            ctrl.densify(model, 600)                            # no noise under DP: must refuse
        except RuntimeError:
            refused = True
        noise = pkg.parallel.shared_split_noise(N, torch.device("cpu"), step=600, seed=5).to(dev)
        stats = ctrl.densify(model, 600, noise=noise)
        q.put((rank, {"sh": d_sh.cpu().numpy(), "pos": grads.dL_dpositions.cpu().numpy(),
                      "rot": grads.dL_drotations.cpu().numpy(), "scl": grads.dL_dscales.cpu().numpy(),
                      "opa": grads.dL_dopacities.cpu().numpy(), "pairs": out.total_pairs, "refused": refused,
                      "densify": (stats.num_cloned, stats.num_split, stats.num_pruned, stats.num_after),
                      "model": {k: getattr(model, k).cpu().numpy() for k in
                                ("positions", "sh_coeffs", "opacities", "rotations", "scales")}}))
    finally:
        dist.destroy_process_group()


def test_two_ranks_two_views_compact_exchange(pkg, dev):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = dict(q.get(timeout=300) for _ in procs)
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    # what one process gets by accumulating the two views' render_backward results
    arrays = pkg.scene.make_gaussians(N, W, H, sh_degree=DEG, seed=21, mu_s=-3.9)
    model = pkg.scene.to_model(arrays, dev)
    settings = pkg.RenderSettings(active_sh_degree=DEG)
    want = None
    for rank in range(2):
        cam = pkg.scene.make_camera(W, H, view=pkg.parallel.view_for_rank(0, rank, 2, 8))
        g = torch.from_numpy(pkg.scene.make_dl_dcolor(W, H, seed=100 + rank)).to(dev)
        gr = pkg.render_backward(g, pkg.render(model, cam, settings), model, cam, settings)
        cur = {"sh": gr.dL_dsh_coeffs, "pos": gr.dL_dpositions, "rot": gr.dL_drotations, "scl": gr.dL_dscales,
               "opa": gr.dL_dopacities}
        want = cur if want is None else {k: want[k] + cur[k] for k in cur}
    assert res[0]["pairs"] != res[1]["pairs"]                  # the two ranks really rendered different views
    for k in ("sh", "pos", "rot", "scl", "opa"):
        ref = want[k].cpu().numpy()
        scale = max(float(np.abs(ref).max()), 1e-30)
        for rank in (0, 1):
            assert np.max(np.abs(res[rank][k].reshape(ref.shape) - ref)) / scale <= 1e-5, (k, rank)
    assert np.array_equal(res[0]["sh"], res[1]["sh"])          # rebuilt in view order: identical on every rank
    # densification under DP: refused without shared noise; with it the replicas are bit-equal after the surgery
    assert res[0]["refused"] and res[1]["refused"]
    assert res[0]["densify"] == res[1]["densify"] and res[0]["densify"][1] > 0 and res[0]["densify"][0] > 0
    for k, v in res[0]["model"].items():
        assert v.shape[0] == res[0]["densify"][3] and np.array_equal(v, res[1]["model"][k]), k
