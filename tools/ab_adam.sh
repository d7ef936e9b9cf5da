R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
  for v in base adamfac; do
    CUGS_HIP_LIBRARY=$R/build_variants/libcugs_$v.so python3 $R/bench.py --no-cpu-baseline --no-parity --config config4 --steps 40 --warmup 5 2>> $O/ab_adam.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v config4 round $rep:', d['ms_per_step'], d['stages_ms'])"
    CUGS_HIP_LIBRARY=$R/build_variants/libcugs_$v.so python3 $R/tools/bench_iteration.py 2>> $O/ab_adam.err | grep "fused into" | sed "s/^/$v /"
  done
done
