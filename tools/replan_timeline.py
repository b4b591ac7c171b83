"""Per-replan diagnostics of the block-resident replan kernel (ufm_region.h) on the headline workload: the end check's
inputs, deferral / burst / sweep counts and the phase timeline (wall_clock64 inside the kernel).
usage: replan_timeline.py [name=value ...]   (ufm_set_param knobs, e.g. region_tiles=10 region_band=0)"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ufm_amd
size, seed = 4096, 7
cost = ufm_amd.synth.cost_map(seed, size, size)
start, goal = ufm_amd.synth.start_goal(size, size)
script = list(ufm_amd.synth.replan_script(seed, size, size, n_patches=12))
args = dict(kv.split("=") for kv in sys.argv[1:])
if "lib" in args:
    ufm_amd.use_library(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), args.pop("lib")))
p = ufm_amd.Planner(ufm_amd.ALGO_FD, 1)
p.set_param("region_debug", 2)
for k, v in args.items():
    p.set_param(k, float(v))
p.set_occupancy_threshold(1); p.set_map(cost); p.set_start(*start); p.set_goal(*goal)
assert p.step() == 0
for i, (k, s, top, left, patch) in enumerate(script):
    t = time.perf_counter()
    p.patch_map(patch, top, left); p.set_start(*s); assert p.step() == 0
    dt = (time.perf_counter() - t) * 1e6
    dbg = np.zeros(96, np.int32); p.L.ufm_debug_lmax(p.h, C.c_void_p(dbg.ctypes.data), 96); f = dbg.view(np.float32)
    print("replan %2d %4.0f us | B %.0f B0 %.0f rbound %.0f done %d exp %d | deferrals L %d R %d again l %d r %d | bursts raise %d lower %d sweeps raise %d lower %d" % (
        k, dt, f[0], f[1], f[2], dbg[9], dbg[12], dbg[22], dbg[23], dbg[24], dbg[25], dbg[26], dbg[27], dbg[28], dbg[29]), "| us: begin %.1f stage %.1f raise %.1f lower %.1f writeback %.1f check %.1f" % tuple(0.01 * (dbg[31 + i] - (dbg[30 + i] if i else 0)) for i in range(6)),
          "| prologue at us: patch asked %.1f, all asked %.1f, staged %.1f, patch applied %.1f, seeded %.1f, bookkeeping %.1f, seeds read %.1f" % tuple(0.01 * dbg[41 + i] for i in (0, 6, 1, 2, 3, 4, 5)),
          "| bursts that changed nothing: lower %d (all +inf around: %d), raise %d" % (dbg[70], dbg[71], dbg[72]),
          "| SIMD of waves 0..15:", "".join(str(int(v)) for v in dbg[50:66]))
