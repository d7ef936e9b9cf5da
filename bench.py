#!/usr/bin/env python3
"""bench.py — forward+backward throughput of the MI355X-native Gaussian-splat rasterizer.

    python bench.py --gpus N --steps K --warmup W [--config config3] [--adam]

With N > 1 and no external launcher (RANK unset) the script starts its N ranks itself (self_launch: N fresh child
processes, one GPU each; the parent never touches a GPU); under `python -m torch.distributed.run ... bench.py --gpus N`
it is one of the ranks.  `n_gpus` in the JSON line is the world size RCCL reports; --gpus != world size is an error.

One "step" = one pass of the hot path over one training view per GPU:
render() + render_backward() (+ the gradient exchange over RCCL when N > 1: --exchange compact | compact-early | allreduce |
auto, the default, which times each after the clock spin-up and keeps the fastest; + FusedAdam.step with --adam /
config4).  Inputs (parameters, dL_dcolor) are resident in HBM before the timed
region.  Default workload = BASELINE.json configs[2]: 1 M synthetic Gaussians, 1920x1080, SH
degree 3 (scene generator: cuda-gaussian-splatting_amd/scene.py).

Rank 0 prints ONE JSON line.  Besides the driver's fields it carries
  roofline      — the dominant kernel: ALGORITHMIC bytes per launch (SURVEY §8d per-unit figures x this
                  run's N, P, pixels) / its mean duration from HIP events recorded on the launch stream
                  inside the timed region (on every 4th timed step: an event between two kernels costs ~5 us
                  of idle device); peak = 8.0 TB/s HBM3E (MI355X_MICROARCH.md);
  frame_roofline— the same for the whole fwd+bwd frame (the north-star target is stated on it);
  stages_ms     — mean per-stage durations from the same events;
  cpu_baseline  — the CPU oracle (a port of the reference's CUDA kernels; the reference has no CPU path)
                  timed on this box's host cores on a bounded sample of the same workload;
  parity        — max-rel-err of RGB and gradients vs the oracle on a reduced scene (checker, not timed).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_MEASURED_GBS = 6290.0      # same table: 6.29 TB/s measured float4 copy

STAGES = ("project_forward", "sort", "raster_forward", "raster_backward", "project_backward")
# the kernel that makes up (all but a memset of) each single-kernel stage, as named in profiles/*_pmc_traffic.json
STAGE_KERNEL = {"project_forward": "k_project_forward", "raster_forward": "k_raster_forward",
                "raster_backward": "k_raster_backward", "project_backward": "k_project_backward"}
CHIP_SIMDS = 256 * 4           # MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32
CHIP_CLOCK_HZ = 2.4e9          # peak shader clock; a wave64 VALU instruction occupies a SIMD for >= 2 cycles


def _latest_table(pattern, kernel):
    """Entry of `kernel` (matched by its bare name, template arguments ignored) in the newest committed PMC table."""
    import glob
    tables = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    if not tables or kernel is None:
        return None, None
    t = json.load(open(tables[-1]))
    hits = [v for k, v in t["kernels"].items() if k.split("<")[0] == kernel]
    if not hits:
        return None, os.path.basename(tables[-1])
    return max(hits, key=lambda v: v.get("launches_sampled", 0)), os.path.basename(tables[-1])


def pmc_traffic(kernel):
    """HBM bytes per launch from the committed PMC passes of this same command (rocprofv3 cannot run
    inside the timed process); None if no table is present or the kernel is not in it."""
    k, table = _latest_table("*_pmc_traffic.json", kernel)
    return (k["hbm_bytes_corrected"] if k else None), table


def pmc_frame_bytes():
    """HBM bytes of ONE frame (all kernels) from the newest committed PMC traffic table, or None."""
    import glob
    tables = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if not tables:
        return None, None
    t = json.load(open(tables[-1]))
    return t.get("frame_hbm_bytes_corrected"), os.path.basename(tables[-1])


def pmc_valu(kernel):
    """SQ_INSTS_VALU per launch (wave-instructions) of `kernel` from the committed SQ counter pass."""
    k, table = _latest_table("*_pmc_sq.json", kernel)
    return (k["SQ_INSTS_VALU"] if k and "SQ_INSTS_VALU" in k else None), table


def algorithmic_bytes(n, c, p, w, h):
    """SURVEY.md §8d, split by stage.  S = 8 + 24*ceil((32+ceil(log2 tiles))/8) is the survey's
    sort term for the REFERENCE-shaped sort; kept as the algorithmic yardstick even though this
    build's sort moves fewer bytes."""
    tiles = ((w + 15) // 16) * ((h + 15) // 16)
    s_sort = 8 + 24 * math.ceil((32 + math.ceil(math.log2(max(tiles, 2)))) / 8)
    b = {
        "project_forward": n * (44 + 12 * c + 48),
        "sort": n * (8 + 20) + p * (12 + s_sort + 8) + tiles * 8,
        "raster_forward": p * 40 + w * h * 20,
        "raster_backward": n * 36 + p * (40 + 36) + w * h * 20,
        "project_backward": n * (72 + 44 + 12 + 12 * c),
    }
    b["frame"] = sum(b.values())
    b["adam"] = 28 * (11 + 3 * c) * n
    return b


ADAM_LR_SCALE = 0.01  # config 4 / --adam: fraction of the reference's default learning rates (see main)
COLOUR_OVERLAP = False # --colour-overlap: geometry + colour halves, the colour half on a side stream under the sort (A/B)
SPATIAL_ORDER = False # --spatial-tile-order
KEY_SORT = True      # --unkeyed-sort: the stage-by-stage route (the sort derives its keys from the projection's arrays)


def timed_step(pkg, model, cam, settings, g, events, exchange, do_allreduce, opt, all_centres=None):
    """One step with a HIP event before/after every stage (events live on torch's current stream,
    which is the stream every kernel of the C ABI is launched on)."""
    R = pkg.rasterizer
    n = model.num_gaussians()
    deg = min(settings.active_sh_degree, model.max_sh_degree())
    class _NoEvent:                       # steps that are not sampled carry no event packets at all
        def record(self):
            pass
    sampled = events is not None
    ev = [torch.cuda.Event(enable_timing=True) if sampled else _NoEvent() for _ in range(len(STAGES) + 1)]
    ev[0].record()
    proj = R.project_gaussians(model.positions, model.rotations, model.scales, model.opacities, model.sh_coeffs,
                               cam, deg, settings.scale_modifier, key_sort=KEY_SORT,
                               colour_on_side_stream=COLOUR_OVERLAP)   # as render(): the SH stream under the sort
    ev[1].record()
    # as in render(): the projection has keyed the sort's workspace, the sort runs on the predicted pair count, the
    # host reads the true one after queueing the blend
    # ... and - as in render() - leaves the order the blend kernels hand their workgroups out in (longest tile list first)
    srt = R.sort_gaussians_predicted(proj.means_2d, proj.depths, proj.radii, proj.tiles_touched, cam.width, cam.height,
                                     want_keys=False, keyed_workspace=proj.sort_workspace,
                                     want_tile_order=R.wants_tile_order(g.device))
    ev[2].record()
    accum = torch.empty((n, pkg._lib.GRAD_STRIDE), dtype=torch.float32, device=g.device)   # cleared by the forward blend
    blend = lambda s: R.rasterize_forward(proj.means_2d, proj.cov_2d_inv, proj.rgb, proj.opacities_act, s.tile_ranges,
                                          s.gaussian_values_sorted, cam.width, cam.height, settings.background,
                                          packed=proj.packed, zero_buf=accum, tile_order=s.tile_order)
    proj.wait_colour()                     # the blend reads rgb / the colour words of the packed records
    fwd = blend(srt)
    if isinstance(srt, R.PendingSort):
        srt, valid = srt.finish()
        if not valid:
            fwd = blend(srt)
    ev[3].record()
    rb = R.rasterize_backward(g, proj.means_2d, proj.cov_2d_inv, proj.rgb, proj.opacities_act, srt.tile_ranges,
                              srt.gaussian_values_sorted, fwd.final_T, fwd.n_contrib, cam.width, cam.height,
                              settings.background, n, packed=proj.packed, unpack=False, zeroed_accum=accum,
                              tile_order=srt.tile_order)
    ev[4].record()
    d_means = torch.empty((n, 2), dtype=torch.float32, device=g.device)
    if opt is not None and not do_allreduce:
        # single GPU: the optimizer step rides in the projection backward (cugs_project_backward_adam) - same bits
        # as project_backward + FusedAdam.step, 472 B/Gaussian less traffic (tests/test_gpu_parity.py)
        import ctypes as C
        adam = opt.begin_fused_step()
        cam_abi = cam.to_abi()
        P = lambda t: C.c_void_p(t.data_ptr())
        pkg._lib.check(pkg._lib.lib.cugs_project_backward_adam(
            n, int(model.sh_coeffs.shape[2]), deg, P(model.positions), P(model.rotations), P(model.scales),
            P(model.opacities), P(model.sh_coeffs), P(proj.radii), P(proj.colour_gate), C.byref(cam_abi),
            float(settings.scale_modifier), P(rb.grad_accum), C.byref(adam), P(d_means),
            C.c_void_p(torch.cuda.current_stream().cuda_stream)), "cugs_project_backward_adam")
        ev[5].record()
        if sampled:
            events.append(ev)
        return srt.total_pairs, fwd, pkg.BackwardOutput(None, None, None, None, None, d_means)
    early = do_allreduce and exchange == "compact-early"
    compact = do_allreduce and (exchange == "compact" or early)
    gated = torch.empty((n, 3), dtype=torch.float32, device=g.device) if compact else None
    flat = torch.empty((11 * n,), dtype=torch.float32, device=g.device) if compact else None
    gather = None
    if early:          # as render_backward(..., on_gated_ready=...): the colour gather starts before the projection backward
        gather = pkg.parallel.begin_colour_gather(R.gated_colour_grad(rb.grad_accum, proj.colour_gate, out=gated))
    pb = R.project_backward(None, None, None, None, model.positions, model.rotations, model.scales,
                            model.opacities, model.sh_coeffs, proj.radii, cam, deg, settings.scale_modifier,
                            grad_accum=rb.grad_accum, colour_gate=proj.colour_gate, dL_dmeans_2d_out=d_means,
                            dL_drgb_gated_out=None if early else gated, skip_sh_grad=compact, geom_flat=flat)
    ev[5].record()
    grads = pkg.BackwardOutput(pb.dL_dpositions, pb.dL_drotations, pb.dL_dscales, pb.dL_dopacities,
                               pb.dL_dsh_coeffs, d_means, geom_flat=flat)
    if early:          # geometry all-reduce + SH rebuild; the gather has been travelling under k_project_backward
        grads = pkg.parallel.finish_exchange(grads, gather, model.positions, deg, int(model.sh_coeffs.shape[2]),
                                             all_cam_centers=all_centres)
    elif compact:      # all-gather 12 B/G colour grads + all-reduce 44 B/G geometry grads, SH grads rebuilt locally
        grads = pkg.parallel.exchange_gradients(grads, gated, model.positions, cam.camera_center(), deg,
                                                int(model.sh_coeffs.shape[2]), all_cam_centers=all_centres)
    elif do_allreduce:   # plain SUM all-reduce of all five tensors (236 B/G)
        pkg.parallel.allreduce_gradients(grads)
    if opt is not None:
        opt.apply_gradients(grads)
        opt.step()
    if sampled:
        events.append(ev)
    return srt.total_pairs, fwd, grads


_PROJ_SNIPPET = r"""
import importlib.util, json, statistics, sys, time
import numpy as np
root, n, w, h, deg, ncores, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m); return m
orc = load("cugs_oracle", root + "/oracle/oracle.py")
rng = np.random.default_rng(1234)
c = (deg + 1) ** 2
z = rng.uniform(2, 10, n)
pos = np.stack([rng.uniform(-1, 1, n) * z * 0.67, rng.uniform(-1, 1, n) * z * 0.67 * h / w, z], 1).astype(np.float32)
arr = dict(positions=pos, rotations=rng.standard_normal((n, 4)).astype(np.float32),
           scales=(rng.standard_normal((n, 3)) * 0.5 - 4.6).astype(np.float32),
           opacities=rng.standard_normal((n, 1)).astype(np.float32),
           sh_coeffs=(0.5 * rng.standard_normal((n, 3, c))).astype(np.float32))
R, t = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
out = {}
for label, th in (("mgaussians_per_s_1_thread", 1), ("mgaussians_per_s_all_cores", ncores)):
    ts = []
    for k in range(reps + 2):                          # BASELINE.md 3: median after two warm-ups
        t0 = time.perf_counter()
        used = orc.project_sh_forward_mt(th, arr, R, t, 0.78 * w, 0.78 * w, w / 2, h / 2, w, h, deg)
        if k >= 2:
            ts.append(time.perf_counter() - t0)
    out[label] = round(n / statistics.median(ts) / 1e6, 2)
    out[label.replace("mgaussians_per_s", "threads")] = used     # small inputs are not spread thinner than 2048 per thread
out["runs"] = reps
print(json.dumps(out))
"""


def projection_cpu_baseline(n, width, height, sh_degree, ncores, reps):
    """SURVEY 8d / BASELINE.md 3: the per-Gaussian forward (k_project_gaussians restated + the reference's
    evaluate_sh_cpu loop shape), single-threaded and OpenMP-parallel over Gaussians on all host cores the process may
    use, median of `reps` runs after two warm-ups - in a fresh interpreter without torch, whose bundled OpenMP runtime
    otherwise shares the process with the oracle's libgomp and serialises it."""
    import subprocess
    try:
        env = dict(os.environ, OMP_WAIT_POLICY="active", OMP_PROC_BIND="false")    # idle-thread wake-ups otherwise dominate
        res = subprocess.run([sys.executable, "-c", _PROJ_SNIPPET, ROOT, str(n), str(width), str(height),
                              str(sh_degree), str(ncores), str(reps)], capture_output=True, text=True, timeout=300, env=env)
        proj = json.loads(res.stdout.strip().splitlines()[-1])
    except Exception as e:                          # a reported baseline must not take the benchmark down
        proj = {"error": str(e)[:200]}
    proj["cores"] = ncores
    return proj


def config1_cpu_baseline(pkg, orc, ncores):
    """BASELINE.json configs[0] / BASELINE.md 3: 10 k Gaussians, 256x256, SH 0 - the projection + 2-D covariance on
    1 thread and on all cores (median of 20), and the whole oracle forward+backward of that scene on one thread
    (median of 5; the oracle's blend is not threaded)."""
    import statistics
    wl1 = pkg.scene.CONFIGS["config1"]
    out = {"workload": wl1.name,
           "projection": projection_cpu_baseline(wl1.n, wl1.width, wl1.height, wl1.sh_degree, ncores, 20)}
    arrays = pkg.scene.make_gaussians(wl1.n, wl1.width, wl1.height, sh_degree=wl1.sh_degree)
    cam = pkg.scene.make_camera(wl1.width, wl1.height)
    K = cam.intrinsics
    g = pkg.scene.make_dl_dcolor(wl1.width, wl1.height)
    ts = []
    for k in range(6):
        t0 = time.perf_counter()
        ref = orc.render(arrays, cam.rotation, cam.translation, K.fx, K.fy, K.cx, K.cy, wl1.width, wl1.height,
                         active_degree=wl1.sh_degree)
        orc.render_backward(g, ref, arrays, K.fx, K.fy, K.cx, K.cy, wl1.width, wl1.height)
        if k >= 1:
            ts.append(time.perf_counter() - t0)
    out["fwd_bwd_mpixels_per_s_1_thread"] = round(wl1.width * wl1.height / statistics.median(ts) / 1e6, 3)
    out["fwd_bwd_runs"] = len(ts)
    return out


def cpu_baseline(pkg, orc, wl, arrays, cam, g, budget_rows):
    """The oracle on host cores, single thread, on a bounded sample of the SAME workload: the full
    per-Gaussian work (projection, SH, pair sort) for all N Gaussians plus the blend forward+backward
    of `budget_rows` image rows, extrapolated linearly in rows to the full frame; each part the median of three
    runs.  Beside it: config 1 as BASELINE.md 3 states it, and the per-Gaussian forward at this workload's N."""
    import statistics
    K = cam.intrinsics

    def med3(fn):
        ts, res = [], None
        for _ in range(3):
            t0 = time.perf_counter()
            res = fn()
            ts.append(time.perf_counter() - t0)
        return statistics.median(ts), res

    t_n, ref = med3(lambda: orc.render(arrays, cam.rotation, cam.translation, K.fx, K.fy, K.cx, K.cy, wl.width,
                                       wl.height, active_degree=wl.sh_degree, rows=(0, 0)))
    budget_rows = min(budget_rows, wl.height)
    r0 = (wl.height - budget_rows) // 2 // 16 * 16
    rows = (r0, min(wl.height, r0 + budget_rows))

    def blend():
        fwd = orc.rasterize_forward(wl.width, wl.height, (0, 0, 0), ref["tile_ranges"], ref["values"], ref["means_2d"],
                                    ref["cov_2d_inv"], ref["rgb"], ref["opacities_act"], rows=rows)
        orc.rasterize_backward(wl.width, wl.height, (0, 0, 0), ref["tile_ranges"], ref["values"], ref["means_2d"],
                               ref["cov_2d_inv"], ref["rgb"], ref["opacities_act"], g, fwd["final_T"], fwd["n_contrib"],
                               wl.n, rows=rows)
    t_rows, _ = med3(blend)
    n = wl.n

    def per_gaussian_backward():
        gm, gc, go = np.zeros((n, 2), np.float32), np.zeros((n, 3), np.float32), np.zeros(n, np.float32)
        orc.project_backward(arrays["positions"], arrays["rotations"], arrays["scales"], arrays["opacities"],
                             ref["view"], K.fx, K.fy, K.cx, K.cy, 1.0, ref["radii"], gm, gc, go)
        orc.sh_backward(ref["degree"], arrays["sh_coeffs"], ref["dirs"], np.zeros((n, 3), np.float32))
    t_nb, _ = med3(per_gaussian_backward)
    nrows = rows[1] - rows[0]
    t_frame = t_n + t_nb + t_rows * wl.height / nrows
    # the GPU boxes give one GPU's job a share of 16 host CPUs; use what the process may run on, at most that
    ncores = max(1, min(len(os.sched_getaffinity(0)), 16))
    return {"value": wl.width * wl.height / t_frame / 1e6, "unit": "Mpixels/s", "cores": 1, "kind": "port",
            "projection_sh_forward": projection_cpu_baseline(wl.n, wl.width, wl.height, wl.sh_degree, ncores, 5),
            "config1": config1_cpu_baseline(pkg, orc, ncores),
            "sample": (f"oracle/cugs_oracle.c single-thread, medians of 3: projection+SH+sort+projection-bwd+SH-bwd of all "
                       f"{wl.n} Gaussians ({t_n + t_nb:.2f} s) + blend fwd+bwd of image rows {rows[0]}..{rows[1]} "
                       f"({t_rows:.2f} s), extrapolated x{wl.height / nrows:.2f} in rows to {wl.width}x{wl.height}"),
            "seconds_per_run": round(t_n + t_nb + t_rows, 3), "host_cores_available": os.cpu_count()}


def parity_probe(pkg, orc, dev):
    """Checker (not timed): a reduced scene through the same code path vs the oracle.  Gradients are reported on both
    yardsticks (oracle/parity.py): `grad_max_rel_err_8d` is SURVEY 8d's element-wise max |g - r| / max(|r|, 1e-6 max|r|),
    `grad_max_err_over_scale` the error against each tensor's own scale (the one the 1e-4 bar is asserted on in tests/).
    Where the element-wise figure exceeds 1e-4 the element is a cancelling sum; `blend_accumulators` shows it: for each
    of the four 2-D accumulators, how many elements are over 1e-4 element-wise and how many of THOSE differ by more
    than the fp32 bound of the magnitudes of their own terms (must be 0)."""
    par = ge.load_oracle_module("parity")
    w, h, n = 640, 360, 20000
    arrays = pkg.scene.make_gaussians(n, w, h, sh_degree=3, seed=77)
    cam = pkg.scene.make_camera(w, h)
    model = pkg.scene.to_model(arrays, dev)
    st = pkg.RenderSettings()
    out = pkg.render(model, cam, st)
    g = pkg.scene.make_dl_dcolor(w, h)
    gd = torch.from_numpy(g).to(dev)
    grads = pkg.render_backward(gd, out, model, cam, st)
    K = cam.intrinsics
    ref = orc.render(arrays, cam.rotation, cam.translation, K.fx, K.fy, K.cx, K.cy, w, h)
    refb = orc.render_backward(g, ref, arrays, K.fx, K.fy, K.cx, K.cy, w, h)
    names = ("dL_dpositions", "dL_drotations", "dL_dscales", "dL_dopacities", "dL_dsh_coeffs")
    rep = par.gradient_report({k: getattr(grads, k).cpu().numpy() for k in names}, refb, names)
    res = {"scene": f"{n}/{w}x{h}/SH3 seed 77", "rgb": par.rel_8d(out.color.cpu().numpy(), ref["color"]),
           "rgb_bit_identical": bool(np.array_equal(out.color.cpu().numpy().view(np.uint32), ref["color"].view(np.uint32))),
           "sort_order_equal": bool(np.array_equal(out.gaussian_indices.cpu().numpy(), ref["values"])),
           "n_contrib_equal": bool(np.array_equal(out.n_contrib.cpu().numpy(), ref["n_contrib"])),
           "tiles_touched_equal": bool(np.array_equal(out.tile_ranges.cpu().numpy(), ref["tile_ranges"]))}
    for name, v in rep["per_tensor"].items():
        res[name] = {"rel_8d": v["rel_8d"], "over_scale": v["over_scale"], "over_1e-4": v["over_bar"],
                     "elements": v["elements"]}
    res["grad_max_rel_err_8d"] = rep["grad_max_rel_err_8d"]
    res["grad_max_err_over_scale"] = rep["grad_max_err_over_scale"]
    res["grad_max_rel_err"] = rep["grad_max_err_over_scale"]      # the name earlier rounds' lines used (of-scale figure)
    # the cancellation evidence, at the stage where the GPU's summation order enters
    out2 = pkg.render(model, cam, st)
    rb = pkg.rasterize_backward(gd, out2.means_2d, out2.cov_2d_inv, out2.rgb, out2.opacities_act, out2.tile_ranges,
                                out2.gaussian_indices, out2.final_T, out2.n_contrib, w, h, (0.0, 0.0, 0.0), n,
                                packed=out2.packed)
    want = orc.rasterize_backward_magnitudes(w, h, (0.0, 0.0, 0.0), ref["tile_ranges"], ref["values"], ref["means_2d"],
                                             ref["cov_2d_inv"], ref["rgb"], ref["opacities_act"], g, ref["final_T"],
                                             ref["n_contrib"], n)
    stage = par.blend_accumulator_report({k: getattr(rb, k).cpu().numpy() for k in par.ACCUMULATORS}, want, want["mag"],
                                         ref["cov_2d_inv"], np.bincount(ref["values"], minlength=n))
    res["blend_accumulators"] = {
        k: {"rel_8d": v["rel_8d"], "over_scale": v["over_scale"], "over_1e-4": v["over_bar"],
            "over_1e-4_beyond_term_bound": v["over_bar_beyond_term_bound"],
            "worst_diff_over_bound": round(v["worst_diff_over_bound"], 4),
            **({"worst_over_1e-4": v["worst_over_bar"]} if v.get("worst_over_bar") else {})}
        for k, v in stage.items()}
    return res


RCCL_ALGOS = {0: "Tree", 1: "Ring", 2: "CollNetDirect", 3: "CollNetChain", 4: "NVLS", 5: "NVLSTree"}
RCCL_PROTOS = {0: "LL", 1: "LL128", 2: "Simple"}


def exchange_payload(mode: str, n: int, c: int, world: int) -> dict:
    """What ONE rank hands to the collectives of one step, and what a bandwidth-optimal (ring or direct) schedule moves
    per rank for it: all-reduce of B bytes -> 2 (W-1)/W B sent and as much received; all-gather of B bytes per rank
    -> (W-1) B received, (W-1)/W of the gathered buffer sent on a ring.  SURVEY 8e's figures at W = 8."""
    geom, sh, colour = 44 * n, 12 * c * n, 12 * n
    f = (world - 1) / world if world > 0 else 0.0
    if mode == "allreduce":
        coll = [{"op": "all_reduce", "bytes": geom + sh, "tensors": 5}]
        sent = 2 * f * (geom + sh)
    elif mode in ("compact", "compact-early"):
        coll = [{"op": "all_gather", "bytes_per_rank": colour}, {"op": "all_reduce", "bytes": geom, "tensors": 1}]
        sent = (world - 1) * colour + 2 * f * geom            # all-gather: W-1 blocks of `colour` bytes leave the rank
    else:
        coll, sent = [], 0.0
    return {"mode": mode, "collectives": coll, "bytes_sent_per_rank": int(sent), "bytes_received_per_rank": int(sent)}


def parse_rccl_log(paths) -> dict:
    """Which algorithm / protocol RCCL picked, from NCCL_DEBUG=INFO lines (NCCL_DEBUG_SUBSYS=INIT,TUNING written to
    NCCL_DEBUG_FILE).  The wording differs between RCCL versions; every line that names a collective together with an
    algorithm and a protocol is taken, numeric ids mapped to names.  Returns {collective: ["Ring/Simple", ...]} plus the
    channel count if announced; {} when nothing could be read (then only the raw tail is kept)."""
    import re
    choices, channels, coll_channels, tail = {}, None, None, []
    pat = re.compile(r"(AllReduce|AllGather|ReduceScatter|Broadcast)\b.*?[Aa]lgo(?:rithm)?\s*[:=]?\s*(\w+).*?[Pp]roto(?:col)?\s*[:=]?\s*(\w+)")
    chan = re.compile(r"(\d+)\s+coll channels|nChannels\s*[:=]?\s*(\d+)|Channel\s+(\d+)/(\d+)")
    for path in paths:
        try:
            lines = open(path, errors="replace").read().splitlines()
        except OSError:
            continue
        tail = [ln for ln in lines if " WARN " not in ln][-5:]
        for ln in lines:
            m = pat.search(ln)
            if m:
                algo = RCCL_ALGOS.get(int(m.group(2)), m.group(2)) if m.group(2).isdigit() else m.group(2)
                proto = RCCL_PROTOS.get(int(m.group(3)), m.group(3)) if m.group(3).isdigit() else m.group(3)
                choices.setdefault(m.group(1), set()).add(f"{algo}/{proto}")
            mc = chan.search(ln)
            if mc and mc.group(1):                                      # "N coll channels, ..." is the authoritative line
                coll_channels = max(coll_channels or 0, int(mc.group(1)))
            elif mc:
                vals = [int(v) for v in mc.groups() if v]
                channels = max(channels or 0, max(vals))
    out = {k: sorted(v) for k, v in choices.items()}
    if coll_channels or channels:
        out["channels"] = coll_channels or channels
    if not out and tail:
        out["unparsed_tail"] = [t[-160:] for t in tail]
    return out


def self_launch(n: int, argv) -> int:
    """`python bench.py --gpus N` without an external launcher: start N fresh child processes of this script, one
    rank (and one GPU, via LOCAL_RANK) each, with the torch.distributed rendezvous variables set.  The parent never
    touches a GPU (no HIP call before or after the spawn) and never re-executes itself; rank 0 inherits stdout and
    prints the one JSON line, the other ranks' stdout is dropped, stderr is shared.  Returns the exit code: 0 only
    if every rank exited 0; when one rank fails the others are terminated (by PID)."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC only on these hosts (RCCL needs it)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    code = 0
    live = list(procs)
    while live:
        for pr in list(live):
            rc = pr.poll()
            if rc is None:
                continue
            live.remove(pr)
            if rc != 0 and code == 0:
                code = rc if rc > 0 else 1
                print(f"bench.py: rank {procs.index(pr)} exited with {rc}; stopping the other ranks", file=sys.stderr)
                for other in live:
                    other.terminate()
        time.sleep(0.05)
    return code


def launcher_dry_run(args, world, rank):
    """The N-rank plumbing without a GPU (tests/test_bench_launcher.py): rendezvous over gloo, barrier, MAX over
    ranks of a per-rank number, one JSON line from rank 0."""
    import torch.distributed as dist
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)                                 # the backends' own chatter goes to stderr (see main)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    sys.stdout.flush()
    os.dup2(stdout_fd, 1)
    if rank == 0:
        w = dist.get_world_size()
        # the N > 1 diagnostics of the real line, with the fields a run over RCCL fills in left null
        exch = {m: dict(exchange_payload(m, 1_000_000, 16, w), exposed_ms=None) for m in ("compact", "compact-early", "allreduce")}
        print(json.dumps({"dry_run": True, "n_gpus": w, "max_over_ranks": float(t.item()),
                          "steps": args.steps, "warmup": args.warmup,
                          "exchange": {"mode": args.exchange, "compute_only_ms": None, "per_mode": exch, "rccl": None}}),
              flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="config3", choices=["config2", "config3", "config4"])
    ap.add_argument("--adam", action="store_true", help="include FusedAdam.step in the step (implied by config4)")
    ap.add_argument("--mu-s", type=float, default=None, help="override the log-scale mean (dense variant: -3.5)")
    ap.add_argument("--cluster", default=None, metavar="FRAC:AREA",
                    help="skewed variant of the scene: FRAC of the Gaussians inside a centred window covering AREA of the "
                         "screen (0.8:0.1 = 80 %% of the splats on 10 %% of the image); scene.make_gaussians(cluster=)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "compact", "compact-early", "allreduce"],
                    help="N>1 gradient exchange: compact = all-gather colour grads + all-reduce geometry grads "
                         "(SH grads rebuilt locally); compact-early = the same with the colour gather started before "
                         "the projection backward (it travels under that kernel); allreduce = SUM all-reduce of all "
                         "five gradient tensors; auto = time each after the spin-up and keep the fastest")
    ap.add_argument("--rehearse-calibration", action="store_true",
                    help="run the auto exchange calibration even with a single rank (one-GPU rehearsal of the N>1 path)")
    ap.add_argument("--spinup-ms", type=float, default=250.0,
                    help="run the step untimed for this long before the warmup so the clocks have left their idle "
                         "state (reported as `spinup` in the JSON line; 0 disables)")
    ap.add_argument("--unkeyed-sort", action="store_true",
                    help="A/B: cugs_project_forward + cugs_sort_pairs_predicted instead of their _keyed variants")
    ap.add_argument("--spatial-tile-order", action="store_true",
                    help="A/B: the blend kernels' workgroups in the spatial (XCD-interleaved) order instead of longest "
                         "tile list first")
    ap.add_argument("--colour-overlap", action="store_true",
                    help="A/B: the projection as geometry + colour halves with the colour half on a side stream underneath "
                         "the sort, instead of ONE launch on the main stream (measured slower in round 3: default off)")
    ap.add_argument("--no-colour-overlap", action="store_true", help="(default; kept for the round-3 A/B scripts)")
    ap.add_argument("--colour-grid-cap", type=int, default=0,
                    help="development library only (CUGS_HIP_LIBRARY=.../libcugs_hip_dev.so): workgroups of the colour half")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=368,
                    help="image rows blended by the CPU baseline sample (default: a third of a 1080p frame; the "
                         "whole leg, repeated three times, takes ~15 s)")
    ap.add_argument("--launcher-dry-run", action="store_true",
                    help="exercise the N-rank launch + rendezvous + one-JSON-line plumbing over gloo, no GPU work")
    args = ap.parse_args()
    global KEY_SORT, COLOUR_OVERLAP, SPATIAL_ORDER
    KEY_SORT = not args.unkeyed_sort
    COLOUR_OVERLAP = args.colour_overlap and not args.no_colour_overlap
    SPATIAL_ORDER = args.spatial_tile_order

    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ      # under torch.distributed.run / self_launch
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and not launched:
        # no external launcher: start the N ranks ourselves, BEFORE anything in this process touches a GPU
        if not args.launcher_dry_run:
            ge._ensure_built()                                           # make (CPU only), once, not N times
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1")) if launched else 1
    rank = int(os.environ.get("RANK", "0")) if launched else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if launched else 0
    if args.gpus != world:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if args.launcher_dry_run:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if launched:
            launcher_dry_run(args, world, rank)
        else:
            print(json.dumps({"dry_run": True, "n_gpus": 1}), flush=True)
        return
    stdout_fd = None
    if launched:
        # RCCL (and gloo) write banners to stdout when a communicator comes up; the contract is ONE JSON line there,
        # so file descriptor 1 points at stderr until rank 0 prints its line
        sys.stdout.flush()
        stdout_fd = os.dup(1)
        os.dup2(2, 1)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # which algorithm / protocol / channel count RCCL picks on this node's xGMI links is the first thing to know when
        # the scaling curve disappoints: its INFO lines go to a per-rank file that rank 0 parses into the JSON line
        rccl_log = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"cugs_bench_rccl_{os.getpid()}_%h_%p.log")
        if os.environ.get("NCCL_DEBUG", "").upper() not in ("INFO", "TRACE"):      # a box default of WARN/VERSION is
            os.environ["NCCL_DEBUG"] = "INFO"                                       # replaced; a louder user setting kept
        os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,TUNING")     # not COLL: a line per call and rank inside the timed region
        os.environ["NCCL_DEBUG_FILE"] = rccl_log
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        world = dist.get_world_size()                                    # what RCCL actually sees
    n_gpus = world

    ge._ensure_built()
    pkg = ge.load_package()
    if SPATIAL_ORDER:
        pkg.rasterizer.TILE_ORDER = False
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    if args.colour_grid_cap:
        import ctypes as _C
        _C.CDLL(pkg.LIB_PATH).cugsdbg_colour_grid_cap(int(args.colour_grid_cap))     # AttributeError on the release build
    wl = pkg.scene.CONFIGS[args.config]
    if args.mu_s is not None:
        wl.mu_s = args.mu_s
    forward_only = args.config == "config2"
    use_adam = args.adam or args.config == "config4"
    cluster = tuple(float(x) for x in args.cluster.split(":")) if args.cluster else None
    arrays = pkg.scene.make_gaussians(wl.n, wl.width, wl.height, sh_degree=wl.sh_degree, mu_s=wl.mu_s, cluster=cluster)
    cam = pkg.scene.make_camera(wl.width, wl.height, view=rank)         # one distinct view per rank
    model = pkg.scene.to_model(arrays, dev)
    settings = pkg.RenderSettings(active_sh_degree=wl.sh_degree)
    g_host = pkg.scene.make_dl_dcolor(wl.width, wl.height, seed=pkg.scene.GRAD_SEED + rank)
    g = torch.from_numpy(g_host).to(dev)
    opt = pkg.FusedAdam(model) if use_adam else None
    if opt is not None:
        # A real optimizer step (every launch, every byte, parameters and moments updated), at 1 % of the default
        # learning rates: Adam with eps = 1e-15 takes lr-sized steps whatever the gradient, and on this synthetic
        # gradient field the defaults blow the splats up (config 4: 40 M -> 110 M pairs within 3000 steps) - the timed
        # steps would not be the named workload.  At 1 % the few hundred steps of a run move the scene by what three
        # default steps would (the pair count stays within a fraction of a percent; reported in config.pairs).
        opt.learning_rates_ = [lr * ADAM_LR_SCALE for lr in opt.learning_rates_]
    c = pkg.sh_coeff_count(wl.sh_degree)
    # every rank knows every view's camera in this benchmark: no device-to-host read of the gathered centres
    all_centres = [pkg.scene.make_camera(wl.width, wl.height, view=r).camera_center().tolist() for r in range(world)]

    exchange = {"mode": args.exchange if args.exchange != "auto" else "compact", "calibration_ms": None}

    def step(events):
        if forward_only:
            out = pkg.render(model, cam, settings, for_backward=False)      # no accumulator to clear, no gate bits
            return out.total_pairs, out, None
        # under torch.distributed.run the exchange step always runs (also for a 1-rank rehearsal)
        return timed_step(pkg, model, cam, settings, g, events, exchange["mode"], launched and exchange["mode"] != "none",
                          opt, all_centres)

    def fence():
        torch.cuda.synchronize(dev)
        if launched:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    # Clock spin-up, untimed and before the warmup: the first ~40 ms of work after the device was idle run on
    # ramping clocks (tools/step_first.py: 1.05 -> 0.90 ms/step over the first 35 steps), and a 5-step warmup ends
    # inside that ramp.  A training run sees the steady state, so that is what the K timed steps should see too.
    # It also runs at least 128 steps (when enabled): some 50-100 steps into a process the HIP runtime stalls once for
    # ~40 ms (tools/step_trend.py: one block of 50 steps at 1.74 instead of 0.96 ms/step; it grows a pool), which a
    # short timed region must not straddle either.
    # Under a launcher every rank must run the SAME number of steps (each one holds collectives): the ranks agree on
    # the elapsed time (MAX over ranks) after every spin-up step.
    spin_t0, spin_steps = time.perf_counter(), 0
    while args.spinup_ms > 0:
        spin_ms = (time.perf_counter() - spin_t0) * 1e3
        if launched:
            t = torch.tensor([spin_ms], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            spin_ms = float(t.item())
        if spin_ms >= args.spinup_ms and spin_steps >= 128:
            break
        step(None)
        spin_steps += 1
    if args.exchange == "auto" and launched and (world > 1 or args.rehearse_calibration) and not forward_only:
        # which exchange is faster depends on what RCCL makes of this node's xGMI links: measure, don't guess.
        # Untimed, AFTER the spin-up (on ramping clocks the mode measured second would win) and before the warmup;
        # the modes alternate (two rounds of one untimed + eight timed steps each), each keeps its best round, and
        # every rank takes the same decision (MAX over ranks of each time).
        modes = ("compact", "compact-early", "allreduce", "none")      # "none": the same step without any collective -
        best = {m: float("inf") for m in modes}                        # the compute time the exchange is exposed on top of
        for _round in range(2):
            for mode in modes:
                exchange["mode"] = mode
                step(None)
                fence()
                t0 = time.perf_counter()
                for _ in range(8):
                    step(None)
                fence()
                best[mode] = min(best[mode], (time.perf_counter() - t0) / 8 * 1e3)
        t = torch.tensor([best[m] for m in modes], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        cal = [float(x) for x in t.tolist()]
        real = [i for i, m in enumerate(modes) if m != "none"]
        exchange["mode"] = modes[min(real, key=lambda i: cal[i])]                   # ties: the earlier mode
        exchange["calibration_ms"] = {m: round(c_, 4) for m, c_ in zip(modes, cal)}

    for _ in range(args.warmup):
        step(None)
    events = []
    fence()
    t0 = time.perf_counter()
    # per-stage HIP events cost ~5 us of device idle each (a marker packet between two kernels); they are
    # recorded on every 4th timed step, which is plenty for the stage averages and the roofline figure
    stride = 4 if args.steps >= 8 else 1
    for k in range(args.steps):
        pairs, _, _ = step(events if k % stride == 0 else None)
    fence()
    elapsed = time.perf_counter() - t0
    if launched:                                          # MAX over ranks
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = n_gpus * wl.width * wl.height * args.steps / elapsed / 1e6
        alg = algorithmic_bytes(wl.n, c, pairs, wl.width, wl.height)
        fused_adam = use_adam and not launched
        if fused_adam:
            # the optimizer step rides in k_project_backward: its stage moves the backward's bytes plus Adam's, minus
            # the five gradient tensors, which are neither written nor read back (1548 B/Gaussian at C = 16)
            extra = alg["adam"] - 2 * 4 * (11 + 3 * c) * wl.n
            alg["project_backward"] += extra
            alg["frame"] += extra
        stages_ms = {}
        if events:
            for i, name in enumerate(STAGES):
                stages_ms[name] = float(np.mean([ev[i].elapsed_time(ev[i + 1]) for ev in events]))
            gpu_ms = float(np.mean([ev[0].elapsed_time(ev[-1]) for ev in events]))
            dom = max(stages_ms, key=stages_ms.get)
            ach = alg[dom] / (stages_ms[dom] * 1e-3) / 1e9
            plain = args.config == "config3" and args.mu_s is None and not use_adam and not args.cluster
            traffic, table = pmc_traffic(STAGE_KERNEL.get(dom)) if plain else (None, None)
            roofline = {"bound": "hbm", "binding": "valu" if dom in ("raster_backward", "raster_forward") else "hbm",
                        "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "frac_hbm": round(ach / HBM_PEAK_GBS, 4),
                        "traffic": traffic,
                        "traffic_source": table, "algorithmic_bytes": int(alg[dom]), "ms": round(stages_ms[dom], 4)}
            if dom in ("raster_backward", "raster_forward"):
                # the blend kernels are bound by VALU issue, not by HBM: their second yardstick is wave-instructions
                # per launch (committed SQ_INSTS_VALU pass of this command) against what the chip can issue in the
                # measured time: SIMDs x clock / 2 (a wave64 instruction holds a SIMD-32 for two cycles at least;
                # DPP / v_cmp / v_cndmask / transcendental instructions hold it 1.6-3.2x longer, profiles/README.md)
                insts, sq_table = pmc_valu(STAGE_KERNEL.get(dom)) if plain else (None, None)
                if insts:
                    ceiling = CHIP_SIMDS * CHIP_CLOCK_HZ / 2.0 * stages_ms[dom] * 1e-3
                    roofline.update({"valu_wave_instructions": int(insts), "valu_frac": round(insts / ceiling, 4),
                                     "valu_source": sq_table})
                roofline["note"] = ("the blend kernels are VALU-issue-bound, not HBM-bound (profiles/README.md): `bound`/`frac` "
                                    "(= `frac_hbm`) is the HBM yardstick the contract asks for, `binding` names what really "
                                    "limits the kernel and `valu_frac` measures it (plain-instruction issue slots used; "
                                    "~85 % of issue TIME with the measured per-class costs)")
            if dom == "sort":
                # SURVEY 8d prices the sort as the reference does it (8 passes over 12-byte pairs: 172 B/pair); this sort
                # moves ~45 B/pair, so `frac` against that yardstick can exceed 1 - it says how much less this sort moves,
                # not how fast it moves it; the rate on its OWN bytes is `achieved_own_bytes`
                actual = wl.n * (4 * 16 + 20 + 32 + 8) + pairs * 45.0
                roofline["achieved_own_bytes"] = round(actual / (stages_ms[dom] * 1e-3) / 1e9, 1)
                roofline["frac_own_bytes"] = round(roofline["achieved_own_bytes"] / HBM_PEAK_GBS, 4)
                roofline["note"] = ("algorithmic bytes are the reference's 8-pass 12-byte-pair sort (SURVEY 8d); this sort moves "
                                    "~45 B/pair: judge it by frac_own_bytes")
            fach = alg["frame"] / (gpu_ms * 1e-3) / 1e9
            frame = {"algorithmic_bytes": int(alg["frame"]), "gpu_ms": round(gpu_ms, 4), "achieved": round(fach, 1),
                     "unit": "GB/s", "frac_of_8TBs": round(fach / HBM_PEAK_GBS, 4),
                     "frac_of_6.29TBs": round(fach / HBM_MEASURED_GBS, 4)}
            # the same with the sort priced at what THIS build's sort moves (~45 B/pair + 124 B/Gaussian) instead of the
            # survey's reference-shaped 172 B/pair: the stricter reading of the frame's memory efficiency
            own = alg["frame"] - alg["sort"] + wl.n * (4 * 16 + 20 + 32 + 8) + pairs * 45.0
            frame["frac_of_8TBs_sort_at_own_bytes"] = round(own / (gpu_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            # third yardstick: the bytes the frame REALLY moves (FETCH_SIZE/WRITE_SIZE of every kernel of a frame, from
            # the committed PMC passes of this command) against the measured 6.29 TB/s copy rate
            measured, mtable = pmc_frame_bytes() if plain else (None, None)
            if measured:
                frame.update({"measured_bytes": int(measured), "measured_source": mtable,
                              "frac_of_6.29TBs_measured_bytes": round(measured / (gpu_ms * 1e-3) / 1e9 / HBM_MEASURED_GBS, 4),
                              "frac_of_8TBs_measured_bytes": round(measured / (gpu_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
            if fused_adam:
                frame["note"] = ("frame = fwd+bwd stages + the optimizer step fused into project_backward "
                                 "(28*(11+3C)*N - 8*(11+3C)*N bytes on top of SURVEY 8d's A_bwd term, counted)")
                if dom == "project_backward":
                    roofline["kernel"] = "project_backward+adam"
            if stages_ms.get("sort", 0.0) > 0.4 * gpu_ms:
                frame["note"] = ("the sort dominates this workload and SURVEY 8d prices it at the reference's 172 B/pair: the "
                                 "frame fraction is inflated by bytes this build never moves (see roofline.frac_own_bytes)")
        else:
            # forward only (configs[1]): no per-stage events; the whole render() against SURVEY 8d's A_fwd
            fwd_bytes = alg["project_forward"] + alg["sort"] + alg["raster_forward"]
            fach = fwd_bytes / (ms_per_step * 1e-3) / 1e9
            roofline = {"bound": "hbm", "kernel": "render(forward only)", "achieved": round(fach, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(fach / HBM_PEAK_GBS, 4), "traffic": None,
                        "algorithmic_bytes": int(fwd_bytes), "ms": round(ms_per_step, 4),
                        "note": "whole forward frame (projection + sort + blend) over the wall time of a step; the sort term "
                                "is SURVEY 8d's reference-shaped 172 B/pair"}
            frame = None
        out = {
            "metric": "fwd+bwd Mpixels/s @1080p, 1M Gaussians, SH3" if args.config == "config3"
                      else f"{'fwd' if forward_only else 'fwd+bwd' + ('+adam' if use_adam else '')} Mpixels/s, {wl.name}",
            "value": round(value, 2), "unit": "Mpixels/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "spinup": {"ms": args.spinup_ms, "steps": spin_steps, "timed": False},
            "untimed_steps": args.warmup + spin_steps,     # everything that ran before the K timed steps
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl.name + (" fwd only" if forward_only else " fwd+bwd") + (" +adam" if use_adam else ""),
                       "n_gaussians": wl.n, "width": wl.width, "height": wl.height, "sh_degree": wl.sh_degree,
                       "pairs": int(pairs), "mu_s": wl.mu_s, "views_per_step": n_gpus,
                       **({"cluster": args.cluster} if args.cluster else {}),
                       **({"adam_learning_rates": f"{ADAM_LR_SCALE} x the reference's defaults (non-zero: a real update)"}
                          if use_adam else {}),
                       "parallelism": f"dp{n_gpus}-views" + (f"+rccl-{exchange['mode']}" if launched else ""),
                       "exchange_calibration_ms": exchange["calibration_ms"]},
            "roofline": roofline, "frame_roofline": frame,
            "stages_ms": {k: round(v, 4) for k, v in stages_ms.items()},
        }
        if launched:
            # N > 1 diagnostics: per exchange mode the bytes a rank puts on the wire and the EXPOSED exchange time
            # (step time with that mode minus the compute-only step of the same box, both from the calibration rounds;
            # for the mode that ran the timed region also from ms_per_step), and what RCCL chose
            cal = exchange["calibration_ms"] or {}
            base = cal.get("none")
            per_mode = {}
            for m in ("compact", "compact-early", "allreduce"):
                e = exchange_payload(m, wl.n, c, n_gpus)
                e["step_ms"] = cal.get(m)
                e["exposed_ms"] = round(cal[m] - base, 4) if (m in cal and base is not None) else None
                per_mode[m] = e
            import glob as _glob
            logs = _glob.glob(os.path.join(os.environ.get("TMPDIR", "/tmp"), f"cugs_bench_rccl_{os.getpid()}_*.log"))
            out["exchange"] = {"mode": exchange["mode"], "compute_only_ms": base,
                               "timed_region_exposed_ms": round(ms_per_step - base, 4) if base is not None else None,
                               "per_mode": per_mode,
                               "rccl": parse_rccl_log(logs) if logs else {"unparsed_tail": ["no RCCL log file was written"]}}
        need_oracle = n_gpus == 1 and not (args.no_cpu_baseline and args.no_parity)
        orc = ge.load_oracle() if need_oracle else None
        if not args.no_cpu_baseline and n_gpus == 1:      # rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(pkg, orc, wl, arrays, cam, g_host, args.cpu_rows)
        if not args.no_parity and not forward_only and n_gpus == 1:
            out["parity"] = parity_probe(pkg, orc, dev)
        if stdout_fd is not None:
            sys.stdout.flush()
            os.dup2(stdout_fd, 1)
        print(json.dumps(out), flush=True)
    if launched:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
