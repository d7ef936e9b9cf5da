#!/bin/bash
# Kernel table of config 3 (and the dense variant) only: tools/quick_sort_stats.sh <tag> -> gpurun_out/<tag>_*
set -e
TAG=$1
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-parity"
rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_stats -- $B --steps 20 --warmup 5 > $O/${TAG}_stats_bench.json 2> $O/${TAG}_stats.log
python3 $R/tools/steady_kernel_stats.py $O/${TAG}_stats $O/${TAG}_kernel_stats.csv 20 > $O/${TAG}_kernel_stats.txt
rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_stats_dense -- $B --mu-s -3.5 --steps 20 --warmup 5 > $O/${TAG}_stats_dense_bench.json 2> $O/${TAG}_stats_dense.log
python3 $R/tools/steady_kernel_stats.py $O/${TAG}_stats_dense $O/${TAG}_kernel_stats_dense.csv 20 > $O/${TAG}_kernel_stats_dense.txt
rm -rf $O/${TAG}_stats $O/${TAG}_stats_dense
python3 - <<PY
import csv
for f in ("$O/${TAG}_kernel_stats.csv", "$O/${TAG}_kernel_stats_dense.csv"):
    print(f)
    tot = 0.0
    for r in csv.DictReader(open(f)):
        per = float(r["TotalDurationNs"]) / 20000.0
        tot += per
        print(f"  {r['Name'][28:84]:58s} {int(r['Calls'])//20:2d}/frame {float(r['AverageNs'])/1000:9.1f} us")
    print(f"  sum per frame {tot:.1f} us")
PY
