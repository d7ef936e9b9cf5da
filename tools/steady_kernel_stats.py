#!/usr/bin/env python3
"""Steady-state kernel table from a rocprofv3 --kernel-trace of bench.py (profiles/<tag>_kernel_stats.csv).

    python3 tools/steady_kernel_stats.py <trace dir> <out.csv> [steps=20] [anchor=k_project_forward]

`rocprofv3 --stats` averages over every launch of the process, including bench.py's clock spin-up and warm-up: its table
sits on ramping clocks and cannot reproduce the bench line (r02: 0.424 ms for k_raster_backward against 0.375 ms from the
in-process HIP events).  This tool keeps the spin-up under the profiler and summarises only the launches of the LAST
`steps` frames of the trace - the timed region of the same command - in the --stats column layout.  A frame starts at
each launch of the anchor kernel (one per step).  Also prints the per-frame kernel sum and the frame span (first kernel
start to last kernel end of a frame, i.e. including the gaps between kernels), both to compare with ms_per_step."""
import csv
import glob
import statistics
import sys


def main():
    trace_dir, out = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    anchor = sys.argv[4] if len(sys.argv) > 4 else "k_project_forward"
    rows = []
    for f in glob.glob(trace_dir + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if anchor in r[2]]
    if len(starts) < steps + 1:
        sys.exit(f"only {len(starts)} frames in the trace, need more than {steps}")
    # the last frame may be cut short by the end of the timed region (parity probe etc. follow): use the `steps` frames
    # that END at the last anchor launch that is followed by a complete frame
    first, last = starts[-steps - 1], starts[-1]
    sel = rows[first:last]
    per = {}
    for s, e, name in sel:
        per.setdefault(name, []).append(e - s)
    total = sum(sum(v) for v in per.values())
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for name, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([name, len(v), sum(v), round(statistics.mean(v), 3), round(100.0 * sum(v) / total, 4), min(v), max(v),
                        round(statistics.pstdev(v), 3)])
    frame_bounds = starts[-steps - 1:]
    spans = [rows[b - 1][1] - rows[a][0] for a, b in zip(frame_bounds[:-1], frame_bounds[1:])]
    sums = [sum(e - s for s, e, _ in rows[a:b]) for a, b in zip(frame_bounds[:-1], frame_bounds[1:])]
    period = [(rows[b][0] - rows[a][0]) for a, b in zip(frame_bounds[:-1], frame_bounds[1:])]
    print(f"steady-state table over the last {steps} frames of {len(starts)} in the trace -> {out}")
    print(f"per-frame kernel sum   mean {statistics.mean(sums) / 1e6:.4f} ms  (min {min(sums) / 1e6:.4f}, max {max(sums) / 1e6:.4f})")
    print(f"per-frame span         mean {statistics.mean(spans) / 1e6:.4f} ms  (first kernel start to last kernel end)")
    print(f"frame period           mean {statistics.mean(period) / 1e6:.4f} ms  (anchor to anchor, under the profiler)")
    for name, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:6]:
        short = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        print(f"  {short[:70]:70s} {len(v) // steps:3d}/frame  mean {statistics.mean(v) / 1e3:8.2f} us")


if __name__ == "__main__":
    main()
