#!/bin/bash
for lib in "lib=build/exp/libufm_urg1.0f.so" "lib=build/exp/libufm_urg0.5f.so" "" "lib=build/exp/libufm_urg0.1f.so" "lib=build/exp/libufm_urg0.0f.so"; do
  echo "== $lib"
  timeout -k 10 120 python tools/plan_probe.py 4096 FD 7 owned_switch_at=0 $lib 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-150
  timeout -k 10 120 python tools/plan_probe.py 4096 FD 7 owned_switch_at=0 owned_waves=16 $lib 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-150
done
