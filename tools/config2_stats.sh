#!/bin/bash
# Config 2 (100 k Gaussians, 1080p, SH 0, forward only): bench line + kernel table -> gpurun_out/<tag>_c2_*
set -e
TAG=$1
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-parity --config config2"
$B --steps 400 --warmup 20 > $O/${TAG}_c2_bench.json 2> $O/${TAG}_c2_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_c2_stats -- $B --steps 50 --warmup 5 --spinup-ms 0 > $O/${TAG}_c2_stats.log 2>&1
cp $O/${TAG}_c2_stats/*/*kernel_stats.csv $O/${TAG}_c2_kernel_stats.csv
python3 - <<PY
import json, csv
d = json.loads(open("$O/${TAG}_c2_bench.json").read().strip().splitlines()[-1])
print("config2", d["ms_per_step"], d["value"])
rows = list(csv.DictReader(open("$O/${TAG}_c2_kernel_stats.csv")))
calls = max(int(r["Calls"]) for r in rows if "raster_forward" in r["Name"])
tot = 0
for r in rows:
    per = float(r["TotalDurationNs"]) / calls / 1000
    tot += per
    print("%-70s %5s %8.1f us  %8.1f us/step" % (r["Name"].replace("(anonymous namespace)::", "")[:70], r["Calls"], float(r["AverageNs"]) / 1000, per))
print("kernel time per step %.1f us, launches per step %.1f" % (tot, sum(int(r["Calls"]) for r in rows) / calls))
PY
