"""Plan-only probe (GPU box): full first plan at a given size / planner, three times, with the resident kernel's own time
(HIP events), tile visits and a checksum of the field -- to compare builds of the library (lib=path: ufm_amd.use_library)
and scheduler knobs (name=value: ufm_set_param).
usage: plan_probe.py [size] [algo] [seed] [lib=path] [heur=1] [name=value ...]"""
import os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, ufm_amd
pos = [a for a in sys.argv[1:] if "=" not in a]
kv = dict(a.split("=", 1) for a in sys.argv[1:] if "=" in a)
size = int(pos[0]) if len(pos) > 0 else 4096
algo = pos[1] if len(pos) > 1 else "FD"
seed = int(pos[2]) if len(pos) > 2 else 7
if "lib" in kv:
    ufm_amd.use_library(os.path.join(ROOT, kv.pop("lib")))
heur = int(kv.pop("heur", "0"))
A = {"FD": ufm_amd.ALGO_FD, "SG": ufm_amd.ALGO_SG, "DFM": ufm_amd.ALGO_DFM}[algo]
cost = ufm_amd.synth.cost_map(seed, size, size)
start, goal = ufm_amd.synth.start_goal(size, size)
p = ufm_amd.Planner(A, 2 if algo == "SG" else 1, bool(heur))
if heur:
    p.set_heuristic_multiplier(float(cost.min()))
p.set_occupancy_threshold(1)
p.set_profiling(1)
for k, v in kv.items():
    p.set_param(k, float(v))
p.set_map(cost)
for rep in range(3):
    p.reset(); p.set_start(*start); p.set_goal(*goal)
    t = time.perf_counter(); assert p.step() == 0; dt = time.perf_counter() - t
    s = p.stats
    g = p.g()
    fin = np.isfinite(g)
    print("%s %d^2 seed %d %s tile %d: plan %.2f ms, resident kernel %.2f ms (stops %d), visits %d (resident %d), evals/elem %.1f, launches %d, expanded %d, finite %d, crc %08x, layout %s, back-pointers %s" % (
        algo, size, seed, kv, p.L.ufm_tile_edge(), dt * 1e3, s.resident_kernel_ms, s.resident_stops, s.tile_visits, s.resident_tile_visits, s.elem_evals / max(1, g.size), s.launches,
        s.expanded, int(fin.sum()), zlib.crc32(np.ascontiguousarray(g).tobytes()), p.check_layout(), p.check_info() if algo != "DFM" else "-"), flush=True)
p.close()
