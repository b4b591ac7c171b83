#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes
# (FETCH_SIZE, WRITE_SIZE: MI355X_MICROARCH.md, rocprofv3 PMC slots) of the bench command and of
# the counter calibration program.  Writes summaries under gpurun_out/profiles_<tag>/.
set -e
TAG=${1:-r2}
OUT=gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p build
[ -x build/traffic_calib ] || /opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -o build/traffic_calib tools/traffic_calib.hip
BENCH="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --patch-inputs host"      # (the headline leg: patches from host memory)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $BENCH > $OUT/bench_kernel_trace.log 2>&1
cp $(find $OUT/kt -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- $BENCH > $OUT/bench_pmc_$c.log 2>&1
  python3 tools/pmc_summary.py $(find $OUT/pmc_$c -name "*counter_collection.csv" | head -1) > $OUT/${TAG}_pmc_$c.txt
  rocprofv3 --pmc $c --output-format csv -d $OUT/calib_$c -- ./build/traffic_calib > $OUT/calib_$c.log 2>&1
  python3 tools/pmc_summary.py $(find $OUT/calib_$c -name "*counter_collection.csv" | head -1) > $OUT/${TAG}_calib_$c.txt
done
rm -rf $OUT/kt $OUT/pmc_* $OUT/calib_FETCH_SIZE $OUT/calib_WRITE_SIZE
tail -1 $OUT/bench_kernel_trace.log
head -5 $OUT/${TAG}_kernel_stats.csv
cat $OUT/${TAG}_calib_FETCH_SIZE.txt $OUT/${TAG}_calib_WRITE_SIZE.txt
grep -A3 "k_relax" $OUT/${TAG}_pmc_FETCH_SIZE.txt | head -20
