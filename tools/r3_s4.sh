#!/bin/bash
O=gpurun_out/r3_s4; mkdir -p $O; rm -f $O/*
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt; tail -15 $O/tests.log | cut -c1-300 | tee -a $O/summary.txt
timeout -k 10 120 python tools/plan_probe.py 4096 FD 7 2>&1 | grep -v amdgpu.ids | cut -c1-175 | tee -a $O/summary.txt
timeout -k 10 120 python tools/replan_probe.py 4096 FD 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
timeout -k 10 120 python tools/replan_timeline.py 2>&1 | grep -v amdgpu.ids > $O/timeline.txt; cut -c1-60,150-400 $O/timeline.txt | tee -a $O/summary.txt
