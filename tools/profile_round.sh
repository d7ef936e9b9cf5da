#!/bin/bash
# Collect the evidence profiles/ holds for one code state (run on the GPU box through gpurun):
#   tools/profile_round.sh <tag>      e.g. r02_a   -> gpurun_out/<tag>_*
# kernel table (rocprofv3 --kernel-trace --stats), HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes),
# SQ counters (two passes), and the bench lines of the four workloads.  rocprofv3 always gets the program itself.
set -e
TAG=$1
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-parity"
python3 $R/bench.py --steps 200 --warmup 20 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
$B --config config2 --steps 20 --warmup 5 > $O/${TAG}_bench_config2.json 2>> $O/${TAG}_bench.err
$B --config config4 --steps 40 --warmup 5 > $O/${TAG}_bench_config4.json 2>> $O/${TAG}_bench.err
$B --mu-s -3.5 --steps 100 --warmup 10 > $O/${TAG}_bench_dense.json 2>> $O/${TAG}_bench.err
# kernel table: the SAME command as the bench line (spin-up included, so the clocks are where the timed region sees
# them); tools/steady_kernel_stats.py keeps only the launches of the last 20 frames = the timed steps
rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_stats -- $B --steps 20 --warmup 5 > $O/${TAG}_stats_bench.json 2> $O/${TAG}_stats.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmc_fetch -- $B --steps 3 --warmup 1 --spinup-ms 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pmc_write -- $B --steps 3 --warmup 1 --spinup-ms 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $O/${TAG}_pmc_sq1 -- $B --steps 3 --warmup 1 --spinup-ms 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/${TAG}_pmc_sq2 -- $B --steps 3 --warmup 1 --spinup-ms 0 > /dev/null 2>&1
python3 $R/tools/pmc_summary.py $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write $O/${TAG}_pmc_traffic.json > $O/${TAG}_pmc_traffic.txt
python3 $R/tools/pmc_sq_summary.py $O/${TAG}_pmc_sq.json $O/${TAG}_pmc_sq1 $O/${TAG}_pmc_sq2 > $O/${TAG}_pmc_sq.txt
python3 $R/tools/steady_kernel_stats.py $O/${TAG}_stats $O/${TAG}_kernel_stats.csv 20 > $O/${TAG}_kernel_stats.txt
rm -rf $O/${TAG}_stats $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write $O/${TAG}_pmc_sq1 $O/${TAG}_pmc_sq2      # raw traces: tens of MB
cat $O/${TAG}_kernel_stats.txt
tail -c 400 $O/${TAG}_bench.json; echo; cat $O/${TAG}_pmc_sq.txt
