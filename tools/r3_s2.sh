#!/bin/bash
# round 3, GPU session 2: patch-level causal filter -- parity, plan and replan timings against the build without it
O=gpurun_out/r3_s2; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt; tail -3 $O/tests.log | tee -a $O/summary.txt
for args in "4096 FD 7" "4096 FD 7 lib=build/exp/libufm_nopf.so" "4096 FD 7 owned_waves=16" "4096 FD 7 owned_waves=16 lib=build/exp/libufm_nopf.so" "2048 SG 1234" "2048 SG 1234 lib=build/exp/libufm_nopf.so" "8192 FD 42 heur=1" "8192 FD 42 heur=1 lib=build/exp/libufm_nopf.so"; do
  timeout -k 10 120 python tools/plan_probe.py $args 2>&1 | grep -v amdgpu.ids >> $O/plan_probe.txt; echo "plan_probe $args rc=$?" >> $O/summary.txt
done
cut -c1-200 $O/plan_probe.txt | tee -a $O/summary.txt
timeout -k 10 120 python tools/replan_probe.py 4096 FD 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
timeout -k 10 120 python tools/replan_probe.py 4096 FD lib=build/exp/libufm_nopf.so 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
timeout -k 10 120 python tools/replan_timeline.py > $O/timeline.txt 2>&1
