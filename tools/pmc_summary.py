#!/usr/bin/env python3
"""Turn rocprofv3 PMC passes into a per-kernel HBM-traffic table (profiles/<tag>_pmc_traffic.json).

Collect, on the GPU box, in SEPARATE passes (FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2;
MI355X_MICROARCH.md, rocprofv3 PMC slots), with --kernel-trace only:
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ...
Units and gfx950 corrections (MI355X_MICROARCH.md, HBM): both counters are in KiB; FETCH_SIZE reports
exactly half the bytes of a wide (16 B/lane) read stream on gfx950 -> doubled; WRITE_SIZE is exact for
16-B streaming stores and for float atomics (one 64-B request per row).  Other access widths are
uncalibrated; the x2 is checked here against kernels whose read volume is known (project_forward /
project_backward read 236 / 124 B per Gaussian).
"""
import collections, csv, glob, json, statistics, sys

def load(pattern, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(pattern):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg

def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0]

fetch = load(sys.argv[1] + "/*/*counter_collection.csv", "FETCH_SIZE")
write = load(sys.argv[2] + "/*/*counter_collection.csv", "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    f = statistics.mean(fetch.get(k, [0.0])) * 1024
    w = statistics.mean(write.get(k, [0.0])) * 1024
    out[short(k)] = {"launches_sampled": len(fetch.get(k, [])), "FETCH_SIZE_bytes_raw": round(f), "WRITE_SIZE_bytes": round(w),
                     "hbm_bytes_corrected": round(2 * f + w)}
# one frame = everything between two launches of the projection: total corrected bytes of all sampled launches / frames
frames = max((len(v) for k, v in fetch.items() if "k_project_forward" in k), default=0)
# ... of a STEADY frame: the first frame of a process sorts on the blocking route (no pair-count prediction yet) and
# launches kernels no later frame does; only kernels seen in all frames but one count, each at its mean per launch
# times its launches per frame
frame_bytes = None
first_frame_only = []
if frames:
    total = 0.0
    for k in set(fetch) | set(write):
        launches = max(len(fetch.get(k, [])), len(write.get(k, [])))
        if launches < max(frames - 1, 1):
            first_frame_only.append(short(k))
            continue
        per_frame = max(1, round(launches / frames))
        total += per_frame * (2 * statistics.mean(fetch.get(k, [0.0])) + statistics.mean(write.get(k, [0.0]))) * 1024
    frame_bytes = round(total)
json.dump({"frames_sampled": frames, "frame_hbm_bytes_corrected": frame_bytes, "first_frame_only_kernels": sorted(first_frame_only), "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py config3, per launch means",
           "correction": "hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950: FETCH_SIZE counts 128-B requests as 64 B)",
           "kernels": out}, open(sys.argv[3], "w"), indent=1)
if frame_bytes:
    print(f"one steady frame, all kernels: {frame_bytes/1e6:.1f} MB corrected ({frames} sampled frames; first-frame-only kernels left out: {len(first_frame_only)})")
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_corrected"])[:12]:
    print(f"{k:40s} fetch_raw {v['FETCH_SIZE_bytes_raw']/1e6:8.1f} MB  write {v['WRITE_SIZE_bytes']/1e6:8.1f} MB  corrected {v['hbm_bytes_corrected']/1e6:8.1f} MB")
