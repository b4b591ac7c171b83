#!/bin/bash
# round 3: diagnostics that go into profiles/ (run on the GPU box)
O=gpurun_out/r3_prof; mkdir -p $O
( echo "== tools/front_timing.py --ys 5 (the shipped form at 4096^2: 8 waves per visit, 512 workgroups, idle ones help; -DUFM_TIMING build)"; timeout -k 10 300 python tools/front_timing.py --ys 5 --param owned_waves=8 2>&1 | grep -v amdgpu.ids
  echo; echo "== tools/front_timing.py --ys 4 --param owned_waves=16 (16 waves per visit, 256 workgroups, early hand-off + in-visit refresh)"; timeout -k 10 300 python tools/front_timing.py --ys 4 --param owned_waves=16 2>&1 | grep -v amdgpu.ids ) > $O/r3_front_timing.txt
( for a in "owned_waves=8" "owned_waves=16" "owned_waves=8 owned_band=2" "owned=0"; do timeout -k 10 200 python tools/sweep_stats.py lib=build/exp/libufm_sstat.so $a 2>&1 | grep -v amdgpu; done ) > $O/r3_sweep_stats.txt
timeout -k 10 120 python tools/replan_timeline.py 2>&1 | grep -v amdgpu.ids > $O/r3_replan_timeline.txt
tail -8 $O/r3_front_timing.txt; head -4 $O/r3_sweep_stats.txt; head -3 $O/r3_replan_timeline.txt | cut -c1-250
