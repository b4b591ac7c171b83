#!/bin/bash
# round 3, GPU session 1: sanity of the suite, baselines (replan timeline, replan probe, plan probe) and the 32 x 32 tile probe
set -o pipefail
O=gpurun_out/r3_s1; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt; tail -3 $O/tests.log | tee -a $O/summary.txt
timeout -k 10 120 python tools/replan_timeline.py > $O/timeline.txt 2>&1; echo "timeline rc=$?" | tee -a $O/summary.txt
timeout -k 10 120 python tools/replan_probe.py > $O/replan_probe.txt 2>&1; echo "replan_probe rc=$?" | tee -a $O/summary.txt; cat $O/replan_probe.txt | tee -a $O/summary.txt
for args in "4096 FD 7" "4096 FD 7 owned_waves=16" "4096 FD 7 lib=build/exp/libufm_t32.so" "4096 FD 7 lib=build/exp/libufm_t32.so owned_flags=2" "4096 FD 7 lib=build/exp/libufm_t32.so owned_band=2" "4096 FD 7 lib=build/exp/libufm_t32.so owned_band=8" "4096 FD 7 lib=build/exp/libufm_t32.so max_iters=64" "2048 SG 1234" "2048 SG 1234 lib=build/exp/libufm_t32.so" "8192 FD 42 heur=1" "8192 FD 42 heur=1 lib=build/exp/libufm_t32.so"; do
  timeout -k 10 120 python tools/plan_probe.py $args >> $O/plan_probe.txt 2>&1; echo "plan_probe $args rc=$?" >> $O/summary.txt
done
cat $O/plan_probe.txt | tee -a $O/summary.txt
