#!/bin/bash
O=gpurun_out/r3_s5; mkdir -p $O; rm -f $O/*
for args in "4096 FD 7" "4096 FD 7 owned_waves=16" "4096 FD 7 owned_waves=16 owned_flags=2" "4096 FD 7 owned_waves=16 owned_flags=16" "4096 FD 7 owned_flags=32" "4096 FD 7 owned=0" "2048 SG 1234" "2048 SG 1234 owned_flags=2"; do
  timeout -k 10 120 python tools/plan_probe.py $args 2>&1 | grep -v amdgpu.ids | cut -c1-40,150-330 >> $O/plan_probe.txt
done
cat $O/plan_probe.txt
timeout -k 10 120 python tools/replan_probe.py 4096 FD 2>&1 | grep -v amdgpu.ids | cut -c1-60,140-300
