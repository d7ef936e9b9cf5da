#!/usr/bin/env python3
"""Turn rocprofv3 SQ counter passes into a per-kernel table (profiles/<tag>_pmc_sq.json): per-launch means of
every counter collected.  Collect on the GPU box, --kernel-trace only, at most 8 SQ counters per pass:
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES \
              SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmc_sq1 -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS \
              SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_sq2 -- python3 bench.py ...
Usage: pmc_sq_summary.py <out.json> <dir> [<dir> ...]
Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles; GRBM_GUI_ACTIVE is summed
over the 8 XCDs (divide by 8 for shader cycles of the dispatch)."""
import collections, csv, glob, json, statistics, sys

def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0]

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[2:]:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, counters in sorted(agg.items()):
    e = {c: round(statistics.mean(v)) for c, v in counters.items()}
    e["launches_sampled"] = max(len(v) for v in counters.values())
    if "GRBM_GUI_ACTIVE" in e and "SQ_INSTS_VALU" in e and e["GRBM_GUI_ACTIVE"] > 0:
        cyc = e["GRBM_GUI_ACTIVE"] / 8.0
        e["valu_wave_instructions_per_simd_cycle"] = round(e["SQ_INSTS_VALU"] / (1024 * cyc), 4)   # ceiling 0.5
    out[k] = e
json.dump({"source": "rocprofv3 --pmc SQ_* (separate passes), bench.py config3, per launch means", "kernels": out},
          open(sys.argv[1], "w"), indent=1)
for k, e in sorted(out.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0))[:8]:
    print(f"{k:48s} VALU {e.get('SQ_INSTS_VALU', 0)/1e6:8.1f} M  SALU {e.get('SQ_INSTS_SALU', 0)/1e6:7.1f} M  LDS {e.get('SQ_INSTS_LDS', 0)/1e6:6.1f} M  "
          f"valu/simd-cycle {e.get('valu_wave_instructions_per_simd_cycle', float('nan')):.3f}")
