#!/bin/bash
# A/B of render()'s colour-half overlap (run through gpurun):  tools/ab_overlap.sh <tag>
set -e
TAG=$1
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-parity"
for i in 1 2; do
  $B --steps 200 --warmup 20 --colour-overlap > $O/${TAG}_overlap_$i.json 2>> $O/${TAG}_ab.err
  $B --steps 200 --warmup 20 --no-colour-overlap > $O/${TAG}_fused_$i.json 2>> $O/${TAG}_ab.err
done
$B --config config4 --steps 40 --warmup 5 --colour-overlap > $O/${TAG}_overlap_c4.json 2>> $O/${TAG}_ab.err
$B --config config4 --steps 40 --warmup 5 --no-colour-overlap > $O/${TAG}_fused_c4.json 2>> $O/${TAG}_ab.err
$B --mu-s -3.5 --steps 100 --warmup 10 --colour-overlap > $O/${TAG}_overlap_dense.json 2>> $O/${TAG}_ab.err
$B --mu-s -3.5 --steps 100 --warmup 10 --no-colour-overlap > $O/${TAG}_fused_dense.json 2>> $O/${TAG}_ab.err
rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_trace -- $B --steps 20 --warmup 5 --colour-overlap > /dev/null 2> $O/${TAG}_trace.log
python3 $R/tools/steady_kernel_stats.py $O/${TAG}_trace $O/${TAG}_overlap_kernel_stats.csv 20 k_project_forward > $O/${TAG}_overlap_kernel_stats.txt
python3 - <<PY
import json, glob, csv
for k in ("overlap_1","fused_1","overlap_2","fused_2","overlap_c4","fused_c4","overlap_dense","fused_dense"):
    d = json.loads(open("$O/${TAG}_%s.json" % k).read().strip().splitlines()[-1])
    print(k, d["ms_per_step"], d.get("stages_ms"))
# timeline of one steady frame: start offsets of each kernel relative to the geometry kernel
rows=[]
for f in glob.glob("$O/${TAG}_trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ","").replace("(anonymous namespace)::","").split("(")[0][:60], r.get("Queue_Id","")))
rows.sort()
anchors=[i for i,r in enumerate(rows) if "k_project_forward" in r[2] and ", 1>" in r[2]]
if len(anchors) > 3:
    a,b=anchors[-3],anchors[-2]
    t0=rows[a][0]
    for s,e,nm,q in rows[a:b]:
        print("%9.1f us  +%7.1f us  q%s  %s" % ((s-t0)/1e3,(e-s)/1e3,q,nm))
PY
rm -rf $O/${TAG}_trace
