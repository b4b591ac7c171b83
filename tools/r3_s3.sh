#!/bin/bash
O=gpurun_out/r3_s3; mkdir -p $O; rm -f $O/*
for args in "max_iters=32" "max_iters=48" "max_iters=64" "max_iters=128" "max_iters=64 owned_band=2" "max_iters=64 owned_band=3" "max_iters=128 owned_band=3" "max_iters=64 owned_waves=16" "max_iters=128 owned_waves=16"; do
  timeout -k 10 120 python tools/plan_probe.py 4096 FD 7 $args 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-175 >> $O/plan_probe.txt
done
for args in "max_iters=64" "max_iters=128"; do
  timeout -k 10 120 python tools/plan_probe.py 2048 SG 1234 $args 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-175 >> $O/plan_probe.txt
  timeout -k 10 120 python tools/plan_probe.py 8192 FD 42 heur=1 $args 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-175 >> $O/plan_probe.txt
done
cat $O/plan_probe.txt
