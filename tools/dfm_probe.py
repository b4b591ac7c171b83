import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ufm_amd
size = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
cost = ufm_amd.synth.cost_map(seed, size, size)
start, goal = ufm_amd.synth.start_goal(size, size)
for mi in (32, 128, 512):
    p = ufm_amd.Planner(2, 1)
    p.set_param("max_iters", mi)
    p.set_occupancy_threshold(1); p.set_map(cost); p.set_start(*start); p.set_goal(*goal)
    t = time.time()
    try:
        rc = p.step()
        print("maxit", mi, "rc", rc, "%.1f ms" % ((time.time() - t) * 1e3), "launches", p.stats.launches, "visits", p.stats.tile_visits, flush=True)
    except Exception as e:
        print("maxit", mi, "EXC", e, "%.1f s" % (time.time() - t), flush=True)
    p.close()
