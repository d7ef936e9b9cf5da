#!/bin/bash
# Same-box A/B of library variants (build_variants/*.so; see DESIGN): tools/ab_variants.sh <tag> <variant> [<variant> ...]
# each variant runs the uniform scene and the two clustered ones, twice, interleaved
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-parity --steps 150 --warmup 10"
for rep in 1 2; do
  for v in "$@"; do
    export CUGS_HIP_LIBRARY=$R/build_variants/libcugs_$v.so
    $B > $O/${TAG}_${v}_uniform_$rep.json 2>> $O/${TAG}.err
    $B --cluster 0.8:0.1 > $O/${TAG}_${v}_cluster_$rep.json 2>> $O/${TAG}.err
    $B --cluster 0.5:0.02 > $O/${TAG}_${v}_cluster2_$rep.json 2>> $O/${TAG}.err
  done
done
python3 - "$O" "$TAG" "$@" <<'PY'
import json, sys
O, TAG, vs = sys.argv[1], sys.argv[2], sys.argv[3:]
for v in vs:
    for s in ("uniform", "cluster", "cluster2"):
        row = []
        for rep in (1, 2):
            d = json.loads(open(f"{O}/{TAG}_{v}_{s}_{rep}.json").read().strip().splitlines()[-1])
            row.append("%.4f (fwd %.3f bwd %.3f)" % (d["ms_per_step"], d["stages_ms"]["raster_forward"], d["stages_ms"]["raster_backward"]))
        print("%-12s %-9s %s" % (v, s, "  |  ".join(row)))
PY
