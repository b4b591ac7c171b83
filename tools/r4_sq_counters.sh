#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): SQ counter passes of the bench command (round-3 verdict, item 3b: is the resident plan kernel
# issue-bound, LDS-bound or waiting?).  One rocprofv3 --pmc pass per counter group (8 SQ slots per pass on gfx950), the program directly
# after `--`.  Writes per-kernel sums under gpurun_out/<tag>/.
# usage: bash tools/r4_sq_counters.sh [tag]
set -e
TAG=${1:-r4_sq}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BENCH="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --patch-inputs host"
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$i -- $BENCH > $OUT/bench_pmc_$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/bench_pmc_$i.log; continue; }
  python3 tools/pmc_summary.py $(find $OUT/pmc_$i -name "*counter_collection.csv" | head -1) > $OUT/sq_pass_$i.txt
  rm -rf $OUT/pmc_$i
done
grep -A12 "k_relax<0, 0, false, 2>\|k_replan_region" $OUT/sq_pass_1.txt | head -60
