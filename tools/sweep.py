"""Tuning helper (GPU box): one FD full plan + N replans at a given size for a set of
scheduler parameters; prints time and work counters.  Not part of the product."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ufm_amd

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=4096)
ap.add_argument("--patches", type=int, default=100)
ap.add_argument("--algo", default="FD")
ap.add_argument("--scales", default="0.25,0.5,1,2,4,1e9")
ap.add_argument("--max-iters", default="128")
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--profile", type=int, default=0)
ap.add_argument("--param", action="append", default=[], help="name=value for ufm_set_param")
ap.add_argument("--heuristic", type=int, default=0, help="heuristic keys with hm = min cost (BASELINE config 5)")
ap.add_argument("--seed", type=int, default=7)
a = ap.parse_args()
algo = {"FD": 0, "SG": 1, "DFM": 2}[a.algo]
size, seed = a.size, a.seed
cost = ufm_amd.synth.cost_map(seed, size, size)
start, goal = ufm_amd.synth.start_goal(size, size)
script = list(ufm_amd.synth.replan_script(seed, size, size, n_patches=a.patches))
p = ufm_amd.Planner(algo, 1 if algo != 1 else 2, bool(a.heuristic))
if a.heuristic:
    p.set_heuristic_multiplier(float(cost.min()))
p.set_occupancy_threshold(1)
p.set_profiling(a.profile)
for kv in a.param:
    name, val = kv.split("=")
    p.set_param(name, float(val))
ref = None
for mi in [int(v) for v in a.max_iters.split(",")]:
  for sc in [float(v) for v in a.scales.split(",")]:
    p.set_param("delta_scale", sc)
    p.set_param("max_iters", mi)
    for kv in a.param:          # after the swept ones: a --param may refine them (e.g. max_iters_short)
        name, val = kv.split("=")
        p.set_param(name, float(val))
    for rep in range(a.reps):
        p.set_map(cost); p.reset(); p.set_start(*start); p.set_goal(*goal)
        t0 = time.perf_counter()
        assert p.step() == 0
        t1 = time.perf_counter()
        s0 = p.stats.as_dict()
        acc = dict(expanded=0, tile_visits=0, launches=0, raise_launches=0, elem_evals=0, tile_iters=0, kernel_ms=0.0, crit_sweeps=0)
        for k, s, top, left, patch in script:
            p.patch_map(patch, top, left); p.set_start(*s)
            assert p.step() == 0
            for kk in acc: acc[kk] += getattr(p.stats, kk)
            qmax = max(locals().get("qmax", 0), p.stats.queued_lower + p.stats.queued_raise)
        t2 = time.perf_counter()
    g = p.g()
    if ref is None: ref = g
    same = bool(np.array_equal(ref, g))
    print("scale %-6g maxit %3d | plan %7.2f ms visits %7d launches %5d evals/elem %6.1f | %d replans %7.2f ms visits %7d launches %5d (raise %5d) cells %8d | same=%s" % (
        sc, mi, (t1 - t0) * 1e3, s0["tile_visits"], s0["launches"], s0["elem_evals"] / max(1, s0["expanded"]),
        len(script), (t2 - t1) * 1e3, acc["tile_visits"], acc["launches"], acc["raise_launches"], acc["expanded"], same), flush=True)
    print("      queued after plan: %d lower / %d raise; max queued during replans: %d" % (s0["queued_lower"], s0["queued_raise"], locals().get("qmax", 0)))
    print("      plan: sweeps/visit(max wave) %.1f  evals/visit %.0f kernel_ms %.2f crit_sweeps/launch %.1f | replans: sweeps/visit %.1f evals/visit %.0f kernel_ms %.2f crit_sweeps/launch %.1f" % (
        s0["tile_iters"] / max(1, s0["tile_visits"]), s0["elem_evals"] / max(1, s0["tile_visits"]), s0["kernel_ms"], s0["crit_sweeps"] / max(1, s0["launches"]),
        acc["tile_iters"] / max(1, acc["tile_visits"]), acc["elem_evals"] / max(1, acc["tile_visits"]), acc["kernel_ms"], acc["crit_sweeps"] / max(1, acc["launches"])), flush=True)
