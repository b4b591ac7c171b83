"""Where does MS-DFM's parity bound come from?  (CPU only, minutes.)

The float fixed point of DFM's update operator is not unique.  This tool measures, on the maps of BASELINE
config 4 (2048^2, seeds 1000..1007), how far apart two evaluation orders of the REFERENCE's own level-1 operator
land: the priority-queue order (oracle/ufm_oracle.c, DFMPlanner<1>::plan) against raster Gauss-Seidel sweeps / a
Jacobi iteration of the same candidates (tools/dfm_fixed_points.c), and what the level-0 planner does there.

usage: dfm_fixed_points.py [size] [seed,seed,..] [gs|jacobi] [--level0]
"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, ufm_amd, oracle_py as orc

args = [a for a in sys.argv[1:] if not a.startswith("--")]
size = int(args[0]) if args else 2048
seeds = [int(v) for v in (args[1] if len(args) > 1 else "1000,1001,1002,1003,1004,1005,1006,1007").split(",")]
mode = {"gs": 1, "jacobi": 0}[args[2] if len(args) > 2 else "gs"]
build = os.path.join(ROOT, "build"); os.makedirs(build, exist_ok=True)
exe = os.path.join(build, "dfm_fixed_points")
subprocess.check_call(["gcc", "-O2", "-fopenmp", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tools", "dfm_fixed_points.c"), "-lm"])
for seed in seeds:
    cost = ufm_amd.synth.cost_map(seed, size, size)
    start, goal = ufm_amd.synth.start_goal(size, size)
    cf = os.path.join(build, "dfm_cost.bin"); cost.tofile(cf)
    o = orc.OraclePlanner(orc.ALGO_DFM, 1, False)
    o.reset(); o.set_occupancy_threshold(1); o.set_map(cost); o.set_start(*start); o.set_goal(*goal)
    assert o.step() == 0
    G0, mask = o.g(), o.trusted_mask()
    of = os.path.join(build, "dfm_field.bin")
    t = time.time()
    subprocess.check_call([exe, str(size), cf, str(int(goal[0])), str(int(goal[1])), "1", str(mode), of], stderr=subprocess.DEVNULL)
    G = np.fromfile(of, np.float32).reshape(size, size)
    a, b = G[mask], G0[mask]
    ud = np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32))
    rel = np.abs(a.astype(np.float64) - b) / b.clip(1e-30)
    line = "DFM-1 %d^2 seed %d: %s of the level-1 candidates vs the oracle's queue order: %d of %d differ, max %d ulp = %.3g rel (%.0f s)" % (
        size, seed, "Gauss-Seidel sweeps" if mode else "Jacobi", int((ud > 0).sum()), int(mask.sum()), int(ud.max()), float(rel.max()), time.time() - t)
    if "--level0" in sys.argv:
        z = orc.OraclePlanner(orc.ALGO_DFM, 0, False)
        z.reset(); z.set_occupancy_threshold(1); z.set_map(cost); z.set_start(*start); z.set_goal(*goal)
        t = time.time(); rc = z.step()
        line += "; level-0 planner: rc %d after %d expansions (%.0f s)" % (rc, z.num_expanded, time.time() - t)
    print(line, flush=True)
