#!/bin/bash
# Bench lines of the four workloads + the kernel table of config 3 and of the dense variant (run through gpurun):
#   tools/quick_stats.sh <tag>   -> gpurun_out/<tag>_*
set -e
TAG=$1
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-parity"
$B --steps 200 --warmup 20 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
$B --config config4 --steps 20 --warmup 3 > $O/${TAG}_bench_config4.json 2>> $O/${TAG}_bench.err
$B --mu-s -3.5 --steps 50 --warmup 5 > $O/${TAG}_bench_dense.json 2>> $O/${TAG}_bench.err
# steady-state tables: spin-up kept under the profiler, only the last 20 frames summarised (tools/steady_kernel_stats.py)
rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_stats -- $B --steps 20 --warmup 5 > $O/${TAG}_stats_bench.json 2> $O/${TAG}_stats.log
python3 $R/tools/steady_kernel_stats.py $O/${TAG}_stats $O/${TAG}_kernel_stats.csv 20 > $O/${TAG}_kernel_stats.txt
rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_stats_dense -- $B --mu-s -3.5 --steps 20 --warmup 5 > $O/${TAG}_stats_dense_bench.json 2> $O/${TAG}_stats_dense.log
python3 $R/tools/steady_kernel_stats.py $O/${TAG}_stats_dense $O/${TAG}_kernel_stats_dense.csv 20 > $O/${TAG}_kernel_stats_dense.txt
rm -rf $O/${TAG}_stats $O/${TAG}_stats_dense
cat $O/${TAG}_kernel_stats.txt
python3 - <<PY
import json
for k in ("bench", "bench_config4", "bench_dense"):
    d = json.loads(open("$O/${TAG}_%s.json" % k).read().strip().splitlines()[-1])
    print(k, d["ms_per_step"], d["value"], d.get("stages_ms"))
PY
