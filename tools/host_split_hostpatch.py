"""Host time per call of a replan on the headline workload, patches handed over from host memory (the bench's `value` leg).
usage: host_split_hostpatch.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, ufm_amd
size, seed = 4096, 7
cost = ufm_amd.synth.cost_map(seed, size, size)
start, goal = ufm_amd.synth.start_goal(size, size)
script = list(ufm_amd.synth.replan_script(seed, size, size, n_patches=100))
h = np.ascontiguousarray(np.stack([s[4] for s in script]))
ptrs = [h.ctypes.data + i * 31 * 31 for i in range(100)]
p = ufm_amd.Planner(ufm_amd.ALGO_FD, 1)
p.set_occupancy_threshold(1); p.set_map(cost); p.set_start(*start); p.set_goal(*goal)
for rep in range(3):
    p.set_map(cost); p.reset(); p.set_start(*start); p.set_goal(*goal); assert p.step() == 0
    t = [0.0] * 4
    pc = time.perf_counter
    kms = 0.0
    for i, (k, s, top, left, patch) in enumerate(script):
        a = pc(); p.patch_map_host(ptrs[i], top, left, 31, 31)
        b = pc(); p.set_start(*s)
        c = pc(); rc = p.step()
        d = pc(); snap = bytes(p.stats)
        e = pc()
        t[0] += b - a; t[1] += c - b; t[2] += d - c; t[3] += e - d
    st = p.stats
    print("per replan us: patch_map (host) %.1f set_start %.1f step %.1f stats copy %.1f | sum %.1f; block kernel (events, every 8th) %.1f us" % tuple([x * 1e4 for x in t] + [sum(t) * 1e4, 1e3 * st.region_kernel_ms / max(1, st.region_timed)]))
