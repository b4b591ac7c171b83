"""GPU box.  Round 4 structural experiment: the resident plan kernel with FIRST VISITS GATED BY AN ARRIVAL ESTIMATE.
Hypothesis: a tile is visited ~4 times per plan because its first visits see immature inputs -- it is activated by the first neighbour the front
reaches it through and visited while its other upwind neighbours are still to come; if a tile's first visit waits until every neighbour that an
(approximate) arrival order puts clearly before it has had ITS first visit, most repeat visits and the ordering band with its hint polling go.
This probe measures the BEST CASE of that idea: the estimate handed in is the true first-arrival value of each tile (smallest value of the tile in
the converged field of an earlier plan), so what is measured is the gating itself, not the quality of a coarse solver.
usage: dag_probe.py [size] [algo] [seed] [lib=path] [name=value ...]   (name=value: ufm_set_param for the gated runs)"""
import ctypes as C, os, sys, time, zlib
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
noise = float(kv.pop("noise", "0"))          # relative noise on the estimate (how good does it have to be?)
coarse = int(kv.pop("coarse", "1"))          # the estimate at a coarser grain: blocks of coarse x coarse tiles share their smallest value
A = {"FD": ufm_amd.ALGO_FD, "SG": ufm_amd.ALGO_SG, "DFM": ufm_amd.ALGO_DFM}[algo]
cost = ufm_amd.synth.cost_map(seed, size, size)
start, goal = ufm_amd.synth.start_goal(size, size)
p = ufm_amd.Planner(A, 2 if algo == "SG" else 1, False)
p.set_occupancy_threshold(1); p.set_profiling(1); p.set_map(cost)


def plan(tag):
    p.reset(); p.set_start(*start); p.set_goal(*goal)
    t = time.perf_counter(); assert p.step() == 0; dt = time.perf_counter() - t
    s = p.stats
    g = p.g()
    print("%-40s plan %.2f ms, resident kernel %.2f ms, visits %d, evals/elem %.1f, crc %08x, back-pointers %s" % (
        tag, dt * 1e3, s.resident_kernel_ms, s.resident_tile_visits, s.elem_evals / max(1, g.size), zlib.crc32(np.ascontiguousarray(g).tobytes()),
        p.check_info() if algo != "DFM" else "-"), flush=True)
    return g


for rep in range(2):
    g = plan("baseline (ordering band)")
T = p.L.ufm_tile_edge()
nx, ny = g.shape
TX, TY = (nx + T - 1) // T, (ny + T - 1) // T
gp = np.full((TX * T, TY * T), np.inf, np.float32)
gp[:nx, :ny] = g
a = gp.reshape(TX, T, TY, T).min(axis=(1, 3)).astype(np.float32)
if coarse > 1:
    cx, cy = (TX + coarse - 1) // coarse, (TY + coarse - 1) // coarse
    ap = np.full((cx * coarse, cy * coarse), np.inf, np.float32); ap[:TX, :TY] = a
    ac = ap.reshape(cx, coarse, cy, coarse).min(axis=(1, 3))
    a = np.repeat(np.repeat(ac, coarse, 0), coarse, 1)[:TX, :TY].copy()
if noise > 0:
    rng = np.random.default_rng(1)
    a = (a * (1.0 + noise * rng.standard_normal(a.shape))).astype(np.float32)
a = np.ascontiguousarray(a, np.float32)
p.L.ufm_debug_set_tile_order.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
rc = p.L.ufm_debug_set_tile_order(p.h, a.ctypes.data, a.size)
assert rc == 0, rc
p.set_param("dag", 1)
for k, v in kv.items():
    p.set_param(k, float(v))
for rep in range(3):
    plan("gated first visits %s" % kv)
p.close()
