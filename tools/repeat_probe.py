"""Repeatability of a plan (GPU box): the same plan N times.  What has to be the same bit for bit in every run is what the reference
guarantees after step(): the elements whose key lies below the start's key (the planners stop at end_condition(); what lies beyond -- here
the last elements behind the start corner -- is left as the schedule found it, like the entries the reference leaves in its queue) and
ufm_check_info's counts 1..3 (no value below the key without its support).  MS-DFM's float fixed point is not unique:
there the runs are held to one another within the planner's tolerance (ufm_amd.tolerances.DFM_RTOL) instead.
usage: repeat_probe.py [size] [algo] [seed] [reps] [lib=path] [name=value ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, ufm_amd
pos = [a for a in sys.argv[1:] if "=" not in a]
kv = dict(a.split("=", 1) for a in sys.argv[1:] if "=" in a)
size = int(pos[0]) if len(pos) > 0 else 4096
algo = pos[1] if len(pos) > 1 else "FD"
seed = int(pos[2]) if len(pos) > 2 else 7
reps = int(pos[3]) if len(pos) > 3 else 40
if "lib" in kv:
    ufm_amd.use_library(os.path.join(ROOT, kv.pop("lib")))
A = {"FD": ufm_amd.ALGO_FD, "SG": ufm_amd.ALGO_SG, "DFM": ufm_amd.ALGO_DFM}[algo]
cost = ufm_amd.synth.cost_map(seed, size, size)
start, goal = ufm_amd.synth.start_goal(size, size)
p = ufm_amd.Planner(A, 2 if algo == "SG" else 1, False)
p.set_occupancy_threshold(1)
for k, v in kv.items():
    p.set_param(k, float(v))
p.set_map(cost)
sx, sy = int(round(start[0])), int(round(start[1]))
ref = None
below_diff = beyond_diff = beyond_runs = support = 0
worst = 0.0
for rep in range(reps):
    p.reset(); p.set_start(*start); p.set_goal(*goal)
    assert p.step() == 0
    g = p.g().copy()
    # the start's key without a heuristic: the largest value among the start elements (a cell planner has one, a node planner the cell's four corners)
    el = g[sx:sx + 1, sy:sy + 1] if algo == "DFM" else g[sx:sx + 2, sy:sy + 2]
    key = float(el[np.isfinite(el)].max())
    if algo != "DFM":
        ci = p.check_info()
        support += int(sum(ci[1:4]))
    if ref is None:
        ref, ref_key = g, key
        continue
    d = ~((ref == g) | (np.isinf(ref) & np.isinf(g)))
    below = d & (np.minimum(ref, g) < min(key, ref_key))
    nb = int((d & ~below).sum())
    if algo == "DFM":
        if below.any():
            worst = max(worst, float((np.abs(ref[below] - g[below]) / np.maximum(np.abs(ref[below]), 1.0)).max()))
        below = below & (np.abs(ref - g) > ufm_amd.tolerances.DFM_RTOL * np.maximum(np.abs(ref), 1.0))
    below_diff += int(below.sum())
    beyond_diff += nb
    beyond_runs += nb > 0
print("%s %d^2 seed %d %s: %d runs; elements below the start's key (%.3f) that differ from the first run%s: %d; beyond the key: %d elements in %d runs; "
      "values without support below the key (check_info 1..3, summed): %d" % (algo, size, seed, kv, reps, ref_key, (" by more than %g (worst %.2e)" % (ufm_amd.tolerances.DFM_RTOL, worst)) if algo == "DFM" else "", below_diff, beyond_diff, beyond_runs, support), flush=True)
p.close()
sys.exit(1 if below_diff or support else 0)
