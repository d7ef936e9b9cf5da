"""Where does the one ~40 ms stall 50-100 steps into a process come from?  (VERDICT r2 #10a)
Runs the config-3 step a few hundred times, timing every host-side call of the step separately, and reports each step in
which one call took more than 3 ms: which call it was, and what the caching allocator / the HIP runtime did across it
(device allocations, segments, reserved bytes).  A blocked KERNEL LAUNCH with no allocator activity points at the HIP
runtime's own pools (kernarg / signal / command buffers); a hipMalloc shows as num_device_alloc moving.
    python tools/stall_trace.py [steps=300] [prealloc_mb=0] [events=0]
events = 1: six timing events recorded around the stages of every 4th step, as bench.py does.
prealloc_mb > 0: reserve-and-release that much device memory through the caching allocator first (does pre-growing the
allocator's pool remove the stall?)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge   # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    prealloc = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    with_events = len(sys.argv) > 3 and sys.argv[3] == "1"
    kept_events = []
    pkg = ge.load_package()
    R = pkg.rasterizer
    dev = torch.device("cuda:0")
    wl = pkg.scene.CONFIGS["config3"]
    arrays = pkg.scene.make_gaussians(wl.n, wl.width, wl.height, sh_degree=wl.sh_degree, mu_s=wl.mu_s)
    model = pkg.scene.to_model(arrays, dev)
    cam = pkg.scene.make_camera(wl.width, wl.height)
    st = pkg.RenderSettings(active_sh_degree=wl.sh_degree)
    g = torch.from_numpy(pkg.scene.make_dl_dcolor(wl.width, wl.height)).to(dev)
    if prealloc:
        x = torch.empty(prealloc << 20, dtype=torch.uint8, device=dev)
        del x
    torch.cuda.synchronize()
    n = wl.n

    def stats():
        s = torch.cuda.memory_stats(dev)
        return (s.get("num_device_alloc", 0), s.get("num_device_free", 0), s.get("segment.all.current", 0),
                s.get("reserved_bytes.all.current", 0) >> 20, s.get("num_alloc_retries", 0))

    t_begin = time.perf_counter()
    slow = []
    per_step = []
    for k in range(steps):
        marks = []
        s0 = stats()
        t = time.perf_counter()

        def lap(name):
            nonlocal t
            now = time.perf_counter()
            marks.append((name, (now - t) * 1e3))
            t = now
        t_step = t
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)] if (with_events and k % 4 == 0) else None
        if ev:
            ev[0].record()
            lap("create + record events")
        proj = R.project_gaussians(model.positions, model.rotations, model.scales, model.opacities, model.sh_coeffs, cam, 3,
                                   1.0, key_sort=True)
        lap("project_gaussians")
        srt = R.sort_gaussians_predicted(proj.means_2d, proj.depths, proj.radii, proj.tiles_touched, wl.width, wl.height,
                                         want_keys=False, keyed_workspace=proj.sort_workspace)
        lap("sort_gaussians_predicted")
        accum = torch.empty((n, 16), dtype=torch.float32, device=dev)
        lap("alloc accumulator")
        fwd = R.rasterize_forward(proj.means_2d, proj.cov_2d_inv, proj.rgb, proj.opacities_act, srt.tile_ranges,
                                  srt.gaussian_values_sorted, wl.width, wl.height, st.background, packed=proj.packed,
                                  zero_buf=accum)
        lap("rasterize_forward")
        if isinstance(srt, R.PendingSort):
            srt, valid = srt.finish()
        lap("finish (wait for the pair count)")
        rb = R.rasterize_backward(g, proj.means_2d, proj.cov_2d_inv, proj.rgb, proj.opacities_act, srt.tile_ranges,
                                  srt.gaussian_values_sorted, fwd.final_T, fwd.n_contrib, wl.width, wl.height,
                                  st.background, n, packed=proj.packed, unpack=False, zeroed_accum=accum)
        lap("rasterize_backward")
        d_means = torch.empty((n, 2), dtype=torch.float32, device=dev)
        pb = R.project_backward(None, None, None, None, model.positions, model.rotations, model.scales, model.opacities,
                                model.sh_coeffs, proj.radii, cam, 3, 1.0, grad_accum=rb.grad_accum,
                                colour_gate=proj.colour_gate, dL_dmeans_2d_out=d_means)
        lap("project_backward")
        if ev:
            for e in ev[1:]:
                e.record()
            kept_events.append(ev)
            if len(kept_events) > 8:                     # read some back, as bench.py does at the end
                old = kept_events.pop(0)
                old[-1].synchronize()
                _ = old[0].elapsed_time(old[-1])
            lap("record + read events")
        del proj, srt, fwd, rb, pb, accum, d_means
        lap("free")
        total = (time.perf_counter() - t_step) * 1e3
        per_step.append(total)
        worst = max(marks, key=lambda m: m[1])
        # the wait for the pair count absorbs the GPU time of a step (~0.7 ms): only launches / allocations that block count
        if worst[1] > 3.0:
            slow.append((k, time.perf_counter() - t_begin, total, worst, s0, stats(), marks))
    torch.cuda.synchronize()
    import statistics
    print("steps %d, median %.3f ms/step, sum of steps over 3 ms: %s" %
          (steps, statistics.median(per_step), ["%d: %.1f ms" % (i, v) for i, v in enumerate(per_step) if v > 3.0]))
    for k, when, total, worst, s0, s1, marks in slow:
        print("step %d (t = %.3f s): %.2f ms, slowest call %-34s %.2f ms" % (k, when, total, worst[0], worst[1]))
        print("    allocator (device allocs, frees, segments, reserved MiB, retries): before %s  after %s" % (s0, s1))
        print("    calls: " + ", ".join("%s %.2f" % m for m in marks))
    if not slow:
        print("no host call above 3 ms")


if __name__ == "__main__":
    main()
