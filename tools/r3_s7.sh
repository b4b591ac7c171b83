#!/bin/bash
for lib in "" "lib=build/exp/libufm_fatlooks.so" "lib=build/exp/libufm_lean32.so" "lib=build/exp/libufm_lean_allhints.so"; do
  echo "== $lib"
  timeout -k 10 120 python tools/plan_probe.py 4096 FD 7 $lib 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-130
  timeout -k 10 120 python tools/plan_probe.py 8192 FD 42 heur=1 $lib 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-130
  timeout -k 10 120 python tools/plan_probe.py 2048 SG 1234 $lib 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-130
done
timeout -k 10 600 python -m pytest tests -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-200
