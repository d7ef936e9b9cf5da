#!/bin/bash
# Same-box A/B of library builds on config 3 and the two clustered views:
#   tools/ab_libs_views.sh <tag> <build_variants/libcugs_X.so stem> ...   (three rounds, interleaved)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for rep in 1 2 3; do
  for view in "" "--cluster 0.8:0.1" "--cluster 0.5:0.02"; do
    for v in "$@"; do
      CUGS_HIP_LIBRARY=$R/build_variants/libcugs_$v.so python3 $R/bench.py --no-cpu-baseline --no-parity --steps 200 --warmup 20 $view 2>> $O/${TAG}.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v [$view] round $rep:', d['ms_per_step'], d['stages_ms'])"
    done
  done
done
