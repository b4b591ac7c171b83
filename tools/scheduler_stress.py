import sys, os, itertools
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/oracle")
import numpy as np, ufm_amd
from helpers import ALGOS, make_pair, check_parity
for algo, lvl, size, seed in (("FD", 1, 1800, 5), ("SG", 2, 1500, 9), ("FD", 1, 3000, 3)):
    cost = ufm_amd.synth.cost_map(seed, size, size)
    start, goal = ufm_amd.synth.start_goal(size, size)
    o, g0 = make_pair(ALGOS[algo], lvl, cost, start, goal)
    assert o.step() == 0
    g0.close()
    ref = None
    variants = [dict(), dict(owned_waves=8), dict(owned_waves=16), dict(owned_flags=32), dict(owned_waves=16, owned_flags=16), dict(owned_waves=16, owned_flags=2),
                dict(owned_band=2.0), dict(owned_waves=8, owned_band=8.0)]
    for rep in range(3):
        for v in variants:
            for full in (0, 1):
                p = ufm_amd.Planner(ALGOS[algo], lvl)
                p.set_param("focused", 0 if full else 1)
                for n_, val in v.items():
                    p.set_param(n_, val)
                p.set_occupancy_threshold(1); p.set_map(cost); p.set_start(*start); p.set_goal(*goal)
                assert p.step() == 0
                assert p.stats.resident_stops == 0, (v, p.stats.resident_stops)
                assert p.check_layout() == (0, 0), v
                if not full:
                    check_parity(o, p, "%s %d %r" % (algo, size, v))
                else:
                    f = p.g()
                    if ref is None: ref = f
                    assert np.array_equal(f, ref), (algo, size, v)
                p.close()
    print(algo, size, "ok: %d runs" % (3 * len(variants) * 2), flush=True)
