#!/bin/bash
# final-code numbers of the BASELINE configurations on one GPU (GPU box): the suite, then bench.py for configs 3 (headline), 2, 4, 5
# usage: bash tools/final_numbers.sh [directory under gpurun_out/]
O=gpurun_out/${1:-r4_final}; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; echo "tests rc=$?"; tail -2 $O/gputests.log
timeout -k 10 300 python bench.py --steps 10 --warmup 3 > $O/headline.json 2> $O/headline.err; echo "headline rc=$?"
timeout -k 10 300 python bench.py --algo SG --size 2048 --seed 1234 --patches 0 --steps 10 --warmup 3 --no-cpu-baseline > $O/config2.json 2> $O/config2.err; echo "config2 rc=$?"
timeout -k 10 600 python bench.py --algo DFM --size 2048 --batch 8 --steps 3 --warmup 1 > $O/config4.json 2> $O/config4.err; echo "config4 rc=$?"
timeout -k 10 600 python bench.py --size 8192 --seed 42 --heuristic --steps 3 --warmup 1 --no-cpu-baseline > $O/config5.json 2> $O/config5.err; echo "config5 rc=$?"
python - "$O" <<'PY'
import json, sys
for n in ("headline", "config2", "config4", "config5"):
    try:
        d = json.loads([l for l in open("%s/%s.json" % (sys.argv[1], n)) if l.startswith("{")][-1])
        r = d.get("roofline", {}); rr = d.get("roofline_replans", {}); ph = d.get("phases", {})
        print(n, "ms/step %.2f value %.1f M cells/s | plan %.2f replans %.2f set_map %.2f | resident %.0f us frac %.4f visits %.0f | region %.1f us x %s done %s/%s | cpu %s" % (
            d["ms_per_step"], d["value"] / 1e6, ph.get("plan_ms", 0), ph.get("replans_ms", 0), ph.get("set_map_ms", 0), r.get("avg_launch_us", 0), r.get("frac", 0), r.get("tile_visits_per_launch", 0),
            rr.get("avg_launch_us", 0), rr.get("launches"), rr.get("region_replans_done"), rr.get("region_replans"), json.dumps(d.get("cpu_baseline", {}).get("phases"))))
    except Exception as e:
        print(n, "failed", e)
PY
