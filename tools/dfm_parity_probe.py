"""MS-DFM: engine vs oracle on the trusted set -- max ulp / relative deviation, launches, time.
usage: dfm_parity_probe.py [size[,size..]] [seed[,seed..]] [lvl[,lvl..]]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, ufm_amd, oracle_py as orc
sizes = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1024,2048").split(",")]
seeds = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1000,1005").split(",")]
lvls = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "1").split(",")]
params = dict(kv.split("=") for kv in sys.argv[4:])
for size in sizes:
    for seed in seeds:
        cost = ufm_amd.synth.cost_map(seed, size, size)
        start, goal = ufm_amd.synth.start_goal(size, size)
        for lvl in lvls:
            o = orc.OraclePlanner(orc.ALGO_DFM, lvl, False)
            o.reset(); o.set_occupancy_threshold(1); o.set_map(cost); o.set_start(*start); o.set_goal(*goal)
            rc = o.step()
            p = ufm_amd.Planner(ufm_amd.ALGO_DFM, lvl)
            for k, v in params.items():
                p.set_param(k, float(v))
            p.set_occupancy_threshold(1); p.set_map(cost); p.set_start(*start); p.set_goal(*goal)
            t = time.time(); assert p.step() == 0; dt = time.time() - t
            m = o.trusted_mask(); a, b = p.g()[m], o.g()[m]
            ud = np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32))
            fin = np.isfinite(a)
            rel = np.abs(a[fin].astype(np.float64) - b[fin]) / b[fin].clip(1e-30)
            print("DFM-%d %d^2 seed %d oracle rc %d trusted %d | differ %d max ulp %d max rel %.3g inf %d | %.1f ms launches %d visits %d" % (
                lvl, size, seed, rc, int(m.sum()), int((ud > 0).sum()), int(ud[fin].max()), float(rel.max()), int((~fin).sum()),
                dt * 1e3, p.stats.launches, p.stats.tile_visits), flush=True)
            p.close()
