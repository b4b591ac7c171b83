"""GPU box: replans of a batch (every map gets its own patch and start move per step)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ufm_amd
ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=2048)
ap.add_argument("--maps", type=int, default=8)
ap.add_argument("--algo", default="DFM")
ap.add_argument("--patches", type=int, default=30)
ap.add_argument("--param", action="append", default=[], help="name=value for ufm_batch_set_param")
ap.add_argument("--lib", default=None, help="another build of the library (path under the repository)")
a = ap.parse_args()
if a.lib:
    ufm_amd.use_library(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), a.lib))
algo = {"FD": 0, "SG": 1, "DFM": 2}[a.algo]
n, size = a.maps, a.size
b = ufm_amd.BatchPlanner(n, algo, 1 if algo != 1 else 2)
b.set_occupancy_threshold(1)
for kv in a.param:
    b.set_param(kv.split('=')[0], float(kv.split('=')[1]))
start, goal = ufm_amd.synth.start_goal(size, size)
scripts = []
for i in range(n):
    b.set_map(i, ufm_amd.synth.cost_map(1000 + i, size, size)); b.set_start(i, *start); b.set_goal(i, *goal)
    scripts.append(list(ufm_amd.synth.replan_script(1000 + i, size, size, n_patches=a.patches)))
t0 = time.perf_counter(); assert b.step() == 0; t1 = time.perf_counter()
cells = 0
per_round, launches = [], []
for k in range(a.patches):
    tr = time.perf_counter()
    for i in range(n):
        _, s, top, left, patch = scripts[i][k]
        b.patch_map(i, patch, top, left); b.set_start(i, *s)
    assert b.step() == 0
    cells += b.stats.expanded
    per_round.append((time.perf_counter() - tr) * 1e6); launches.append(b.stats.launches)
t2 = time.perf_counter()
print("%s %d x %d^2: plan %.1f ms; %d batch replans %.2f ms each (%d launches in the last step, %.0f cells per step); block kernel finished %d of %d map replans alone" % (
    a.algo, n, size, (t1 - t0) * 1e3, a.patches, (t2 - t1) * 1e3 / a.patches, b.stats.launches, cells / a.patches, b.stats.region_replans_done, b.stats.region_replans))
pr, la = np.array(per_round), np.array(launches)
one = la == 1
print("  rounds the block kernel finished alone: %d, %.0f us each; the others: %d, %.0f us each, launches per round min / median / max %s" % (
    one.sum(), pr[one].mean() if one.any() else 0, (~one).sum(), pr[~one].mean() if (~one).any() else 0,
    (int(la[~one].min()), int(np.median(la[~one])), int(la[~one].max())) if (~one).any() else None))
