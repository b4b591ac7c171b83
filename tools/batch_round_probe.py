import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, ufm_amd
n, size = 8, 2048
b = ufm_amd.BatchPlanner(n, 2, 1)
b.set_occupancy_threshold(1)
start, goal = ufm_amd.synth.start_goal(size, size)
scripts = []
for i in range(n):
    b.set_map(i, ufm_amd.synth.cost_map(1000 + i, size, size)); b.set_start(i, *start); b.set_goal(i, *goal)
    scripts.append(list(ufm_amd.synth.replan_script(1000 + i, size, size, n_patches=40)))
assert b.step() == 0
for k in range(40):
    t0 = time.perf_counter()
    for i in range(n):
        _, s, top, left, patch = scripts[i][k]
        b.patch_map(i, patch, top, left); b.set_start(i, *s)
    t1 = time.perf_counter()
    assert b.step() == 0
    t2 = time.perf_counter()
    st = b.stats
    print("round %2d: patches %.0f us step %.0f us | launches %d raise %d visits %d u_ms %.3f p_ms %.3f" % (k, (t1-t0)*1e6, (t2-t1)*1e6, st.launches, st.raise_launches, st.tile_visits, st.u_ms, st.p_ms))
