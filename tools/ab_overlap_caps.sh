#!/bin/bash
# Sweep of the colour half's grid cap (development library), against the one-launch projection:  tools/ab_overlap_caps.sh <tag>
set -e
TAG=$1
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export CUGS_HIP_LIBRARY=$R/cuda-gaussian-splatting_amd/libcugs_hip_dev.so
B="python3 $R/bench.py --no-cpu-baseline --no-parity --steps 100 --warmup 10"
$B --no-colour-overlap > $O/${TAG}_cap_fused.json 2>> $O/${TAG}_caps.err
for cap in 64 128 256 512 1024 4096; do
  $B --colour-overlap --colour-grid-cap $cap > $O/${TAG}_cap_$cap.json 2>> $O/${TAG}_caps.err
done
$B --no-colour-overlap > $O/${TAG}_cap_fused2.json 2>> $O/${TAG}_caps.err
python3 - <<PY
import json
for k in ("fused","64","128","256","512","1024","4096","fused2"):
    d = json.loads(open("$O/${TAG}_cap_%s.json" % k).read().strip().splitlines()[-1])
    print("%-7s %.4f ms  %s" % (k, d["ms_per_step"], d.get("stages_ms")))
PY
