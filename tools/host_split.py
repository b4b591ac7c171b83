import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, ufm_amd, torch
size, seed = 4096, 7
cost = ufm_amd.synth.cost_map(seed, size, size)
start, goal = ufm_amd.synth.start_goal(size, size)
script = list(ufm_amd.synth.replan_script(seed, size, size, n_patches=100))
d_patches = torch.from_numpy(np.stack([s[4] for s in script])).cuda()
ptrs = [d_patches[i].data_ptr() for i in range(100)]
p = ufm_amd.Planner(ufm_amd.ALGO_FD, 1)
p.set_occupancy_threshold(1); p.set_map(cost); p.set_start(*start); p.set_goal(*goal)
for rep in range(3):
    p.set_map(cost); p.reset(); p.set_start(*start); p.set_goal(*goal); assert p.step() == 0
    t = [0.0] * 4
    pc = time.perf_counter
    for i, (k, s, top, left, patch) in enumerate(script):
        a = pc(); p.patch_map_device(ptrs[i], top, left, 31, 31)
        b = pc(); p.set_start(*s)
        c = pc(); rc = p.step()
        d = pc(); snap = bytes(p.stats)
        e = pc()
        t[0] += b - a; t[1] += c - b; t[2] += d - c; t[3] += e - d
    print("per replan us: patch_map_device %.1f set_start %.1f step %.1f stats %.1f | u_ms+p_ms %.1f" % tuple([x * 1e4 for x in t] + [0.0]))
