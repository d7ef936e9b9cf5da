#!/bin/bash
# SQ counter passes + HBM traffic of one bench command, summarised per kernel (run through gpurun):
#   tools/pmc_quick.sh <tag> [bench args...]   -> gpurun_out/<tag>_pmc_sq.{json,txt}, <tag>_pmc_traffic.{json,txt}
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-parity --spinup-ms 0 $@"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmc_fetch -- $B --steps 3 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pmc_write -- $B --steps 3 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $O/${TAG}_pmc_sq1 -- $B --steps 3 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/${TAG}_pmc_sq2 -- $B --steps 3 --warmup 1 > /dev/null 2>&1
python3 $R/tools/pmc_summary.py $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write $O/${TAG}_pmc_traffic.json > $O/${TAG}_pmc_traffic.txt
python3 $R/tools/pmc_sq_summary.py $O/${TAG}_pmc_sq.json $O/${TAG}_pmc_sq1 $O/${TAG}_pmc_sq2 > $O/${TAG}_pmc_sq.txt
cat $O/${TAG}_pmc_traffic.txt $O/${TAG}_pmc_sq.txt
