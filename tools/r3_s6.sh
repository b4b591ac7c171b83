#!/bin/bash
for args in "owned_flags=0" "owned_flags=1" "owned_flags=1 owned_band=3" "owned_flags=1 owned_band=2" "owned_flags=1 owned_waves=16" "owned_flags=33" ; do
  timeout -k 10 120 python tools/plan_probe.py 4096 FD 7 $args 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-175
done
