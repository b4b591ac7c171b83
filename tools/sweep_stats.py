"""Diagnostic (GPU box, -DUFM_SWEEPSTAT build): what the patch sweeps of a plan find -- how many change nothing, how many node
values change per sweep, how long the bursts are.  usage: sweep_stats.py lib=build/exp/libufm_sstat.so [size] [algo] [name=value ...]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, ufm_amd
pos = [a for a in sys.argv[1:] if "=" not in a]
kv = dict(a.split("=", 1) for a in sys.argv[1:] if "=" in a)
ufm_amd.use_library(os.path.join(ROOT, kv.pop("lib")))
size = int(pos[0]) if pos else 4096
algo = pos[1] if len(pos) > 1 else "FD"
A = {"FD": ufm_amd.ALGO_FD, "SG": ufm_amd.ALGO_SG, "DFM": ufm_amd.ALGO_DFM}[algo]
L = ufm_amd.load_library()
L.ufm_debug_sstat.argtypes = [C.c_void_p, C.c_int]
cost = ufm_amd.synth.cost_map(7, size, size)
start, goal = ufm_amd.synth.start_goal(size, size)
p = ufm_amd.Planner(A, 2 if algo == "SG" else 1)
for k, v in kv.items():
    p.set_param(k, float(v))
p.set_occupancy_threshold(1); p.set_map(cost); p.set_start(*start); p.set_goal(*goal)
L.ufm_debug_sstat(None, 1)
assert p.step() == 0
d = (C.c_ulonglong * 32)(); L.ufm_debug_sstat(d, 1)
n = p.stats.expanded
print("%s %d^2 %s: visits %d, elements %d" % (algo, size, kv, p.stats.tile_visits, n))
print("  patch sweeps %d (%.1f per element x 16), of which changed nothing %d (%.0f %%); node values changed %d (%.2f per element; lowered %d); changes per sweep %.2f" % (
    d[0], 16.0 * d[0] / n, d[1], 100.0 * d[1] / max(1, d[0]), d[2], d[2] / n, d[5], d[2] / max(1, d[0])))
print("  bursts %d (%.2f sweeps each), first sweep changed nothing in %d (%.0f %%)" % (d[3], d[0] / max(1, d[3]), d[4], 100.0 * d[4] / max(1, d[3])))
print("  sweeps per burst 1..16+: %s" % [int(d[8 + i]) for i in range(16)])
