import os, sys, time, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np, ufm_amd
size, seed = 4096, 7
cost = ufm_amd.synth.cost_map(seed, size, size)
start, goal = ufm_amd.synth.start_goal(size, size)
script = list(ufm_amd.synth.replan_script(seed, size, size, n_patches=100))
p = ufm_amd.Planner(ufm_amd.ALGO_FD, 1)
p.set_param("region_debug", 2)
p.set_occupancy_threshold(1); p.set_map(cost); p.set_start(*start); p.set_goal(*goal)
assert p.step() == 0
for i, (k, s, top, left, patch) in enumerate(script):
    t = time.perf_counter()
    p.patch_map(patch, top, left); p.set_start(*s); assert p.step() == 0
    dt = (time.perf_counter() - t) * 1e6
    dbg = np.zeros(96, np.int32); p.L.ufm_debug_lmax(p.h, C.c_void_p(dbg.ctypes.data), 96); f = dbg.view(np.float32)
    if not dbg[9] or dt > 300:
        print("replan %2d %4.0f us launches %d | B %.1f B0 %.1f rbound %.1f m_r %.1f m_l %.1f nr %d nl %d npr %d npl %d done %d giveup %d sweeps %d exp %d | deferred tiles L %d (min %.1f) R %d (min %.1f) start_in %d start g %s | subrounds %d rounds %d patch (%d,%d) start %s" % (
            k, dt, p.stats.launches, f[0], f[1], f[2], f[3], f[4], dbg[5], dbg[6], dbg[7], dbg[8], dbg[9], dbg[10], dbg[11], dbg[12], dbg[13], f[14], dbg[15], f[16], dbg[17],
            [round(float(f[18 + j]), 1) if dbg[18 + j] != -1 else None for j in range(4)], dbg[24], dbg[25], top, left, s))
