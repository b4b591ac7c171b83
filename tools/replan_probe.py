"""Replans of the headline workload (FD-1 4096^2, 100 patches): time per replan, block kernel completion rate.
usage: replan_probe.py [size] [algo] [lib=path] [host=1] [name=value ...]   (host=1: the patches handed over as host buffers, ufm_patch_map)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, ufm_amd, torch
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
algo = sys.argv[2] if len(sys.argv) > 2 else "FD"
params = dict(kv.split("=") for kv in sys.argv[3:])
host_patches = params.pop("host", "0") != "0"
if "lib" in params:
    ufm_amd.use_library(os.path.join(ROOT, params.pop("lib")))
A = {"FD": ufm_amd.ALGO_FD, "SG": ufm_amd.ALGO_SG, "DFM": ufm_amd.ALGO_DFM}[algo]
seed = 7
cost = ufm_amd.synth.cost_map(seed, size, size)
start, goal = ufm_amd.synth.start_goal(size, size)
script = list(ufm_amd.synth.replan_script(seed, size, size, n_patches=100))
d_patches = torch.from_numpy(np.stack([s[4] for s in script])).cuda()
for rep in range(3):
    p = ufm_amd.Planner(A, 2 if algo == "SG" else 1)
    for k, v in params.items():
        p.set_param(k, float(v))
    p.set_occupancy_threshold(1); p.set_map(cost); p.set_start(*start); p.set_goal(*goal)
    t = time.perf_counter(); assert p.step() == 0; t_plan = time.perf_counter() - t
    times, cells, visits, evals = [], 0, 0, 0
    for i, (k, s, top, left, patch) in enumerate(script):
        t = time.perf_counter()
        if host_patches: p.patch_map(patch, top, left)
        else: p.patch_map_device(d_patches[i].data_ptr(), top, left, 31, 31)
        p.set_start(*s)
        assert p.step() == 0
        times.append(time.perf_counter() - t); cells += p.stats.expanded; visits += p.stats.tile_visits; evals += p.stats.elem_evals
        if times[-1] > 1e-3: print("   slow replan %d: %.0f us, launches %d (raise %d), region done %d/%d, expanded %d, visits %d, u_ms %.2f p_ms %.2f" % (i, times[-1] * 1e6, p.stats.launches, p.stats.raise_launches, p.stats.region_replans_done, p.stats.region_replans, p.stats.expanded, p.stats.tile_visits, p.stats.u_ms, p.stats.p_ms), flush=True)
    ts = np.array(times) * 1e6
    if rep == 2: print("   slowest:", ", ".join("#%d %.0f us" % (i, ts[i]) for i in np.argsort(-ts)[:10]), "| sum without them %.2f ms" % (np.sort(ts)[:-10].sum() / 1e3), flush=True)
    print("%s %d^2 %s: plan %.2f ms; 100 replans %.2f ms (median %.0f us, p90 %.0f, max %.0f); block kernel %d/%d done; cells %d visits %d patch-sweeps/replan %d launches(last) %d back-pointers %s" % (
        algo, size, params, t_plan * 1e3, ts.sum() / 1e3, np.median(ts), np.percentile(ts, 90), ts.max(),
        p.stats.region_replans_done, p.stats.region_replans, cells, visits, evals // 1600, p.stats.launches, p.check_info() if algo != "DFM" else "-"), flush=True)
    p.close()
