"""Progress of the resident plan kernel over time in the PRODUCT build (no instrumentation): the kernel is told to hand back to the
launch chain after T ms (owned_limit_ms) and reports the tile visits it made until then; T swept.  The step's total time then says
what the launch chain needs for the rest.  (profiles/r4_progress_curve.txt)
usage: progress_probe.py [waves: 8 | 16] [size] [algo]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, ufm_amd
waves = int(sys.argv[1]) if len(sys.argv) > 1 else 8
size = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
algo = sys.argv[3] if len(sys.argv) > 3 else "FD"
A = {"FD": ufm_amd.ALGO_FD, "SG": ufm_amd.ALGO_SG, "DFM": ufm_amd.ALGO_DFM}[algo]
cost = ufm_amd.synth.cost_map(7, size, size)
start, goal = ufm_amd.synth.start_goal(size, size)
p = ufm_amd.Planner(A, 2 if algo == "SG" else 1)
p.set_occupancy_threshold(1); p.set_profiling(1)
p.set_map(cost)
p.set_param("owned_waves", waves)
print("%s %d^2, %d waves per visit" % (algo, size, waves))
for T in (1000, 14, 12, 10, 8, 6, 5, 4, 3, 2, 1.5, 1, 0.5):
    p.set_param("owned_limit_ms", T)
    p.reset(); p.set_start(*start); p.set_goal(*goal)
    t = time.perf_counter(); assert p.step() == 0; dt = time.perf_counter() - t
    s = p.stats
    print("%6g ms limit: step %.1f ms, resident visits %d kernel %.2f ms, total visits %d launches %d" % (T, dt * 1e3, s.resident_tile_visits, s.resident_kernel_ms, s.tile_visits, s.launches), flush=True)
p.close()
