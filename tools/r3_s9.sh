#!/bin/bash
timeout -k 10 600 python -m pytest tests -m gpu -q -x 2>&1 | grep -v amdgpu | tail -3 | cut -c1-200
run() { timeout -k 10 300 python bench.py "$@" --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); rr=d.get('roofline_replans',{}); ph=d['phases']
print('%-70s ms/step %.2f plan %.2f replans %.2f region %.1f us done %s/%s' % (' '.join(sys.argv[1:]), d['ms_per_step'], ph['plan_ms'], ph['replans_ms'], rr.get('avg_launch_us',0), rr.get('region_replans_done'), rr.get('region_replans')))" "$@"; }
run --steps 5 --warmup 2
run --steps 5 --warmup 2 --param cont_lower=0
for prm in "" "--param cont_lower=0" "--param cont_lower=4" "--param cont_lower=16" "--param region_tiles=8 --param region_ahead=3" "--param region_tiles=8 --param region_ahead=3 --param cont_lower=16"; do
  run --algo DFM --size 2048 --batch 8 --steps 2 --warmup 1 $prm
done
run --size 8192 --seed 42 --heuristic --steps 3 --warmup 1
run --size 8192 --seed 42 --heuristic --steps 3 --warmup 1 --param cont_lower=0
