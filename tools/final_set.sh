set -e
T=${1:-r03_zq}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
bash $R/tools/profile_round.sh $T > $O/${T}_round.log 2>&1
cd /tmp
B="python3 $R/bench.py --no-cpu-baseline --no-parity"
$B --cluster 0.8:0.1 --steps 100 --warmup 10 > $O/${T}_bench_cluster.json 2>> $O/${T}_bench.err
$B --cluster 0.5:0.02 --steps 100 --warmup 10 > $O/${T}_bench_cluster2.json 2>> $O/${T}_bench.err
python3 $R/tools/bench_cpp_host.py > $O/${T}_cpp_host.log 2>&1
python3 $R/tools/bench_iteration.py > $O/${T}_iteration.log 2>&1
cd $R && timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/${T}_gpu_tests.log 2>&1
tail -3 $O/${T}_gpu_tests.log
