"""GPU box.  The reference's second recorded mission (wall-b: a binary bitmap, free cells all cost 1, paths tie) OPEN-LOOP through the engine:
every step starts from the position the LOG printed (six decimals), the engine replans and extracts; printed: how the path cost and the path
length compare with the log's.  (Closed-loop the engine parts from this log in step 1: same path cost, another of several equal-cost way
points -- which one the reference takes depends on the stale values its queue order leaves beyond the start's key, tests/test_reference_mission.py.)
usage: python tools/wallb_open_loop.py [floor=1]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import ufm_amd
from ufm_amd_pkg import harness
import test_reference_mission as trm
kv = dict(a.split("=", 1) for a in sys.argv[1:] if "=" in a)
name = kv.get("log", "wall-b")
g = ufm_amd.Planner(ufm_amd.ALGO_FD, 0, True)
g.set_param("start_cell_floor", float(kv.get("floor", "1")))
worst_c = worst_d = 0.0
same_c = same_d = n = 0
for k, st, tc, td in trm.open_loop(name, g):
    rc, rd = abs(tc - float(st["cost"])) / float(st["cost"]), abs(td - float(st["dist"])) / float(st["dist"])
    worst_c, worst_d = max(worst_c, rc), max(worst_d, rd)
    same_c += trm.g6(tc) == st["cost"]; same_d += trm.g6(td) == st["dist"]
    n += 1
    if trm.g6(tc) != st["cost"] or trm.g6(td) != st["dist"]:
        print("step %2d pos %s: cost %s (log %s) length %s (log %s)" % (k, st["pos"], trm.g6(tc), st["cost"], trm.g6(td), st["dist"]))
print("%s open-loop, %d steps: path cost to the printed digit in %d, path length in %d; worst relative difference cost %.2e length %.2e" % (name, n, same_c, same_d, worst_c, worst_d))
g.close()
