#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
for spec in "0.8:0.1" "0.5:0.02"; do
rocprofv3 --kernel-trace --output-format csv -d $O/cl_stats -- python3 $R/bench.py --no-cpu-baseline --no-parity --steps 20 --warmup 5 --cluster $spec > /dev/null 2> $O/cl_stats.log
python3 $R/tools/steady_kernel_stats.py $O/cl_stats $O/cl_kernel_stats.csv 20 > /dev/null
echo "cluster $spec"
python3 - <<PY
import csv
for r in csv.DictReader(open("$O/cl_kernel_stats.csv")):
    print(f"  {r['Name'][28:84]:58s} {int(r['Calls'])//20:2d}/frame {float(r['AverageNs'])/1000:9.1f} us")
PY
rm -rf $O/cl_stats
done
