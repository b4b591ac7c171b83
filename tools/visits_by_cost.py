"""Diagnostic (GPU box, -DUFM_TIMING build): visits per tile of the resident plan kernel against the tile's mean traversal cost.
Premise to check (round 4): the ordering band is one width in VALUE units (2.5 tile crossings at the map's mean cost), i.e. tens of tiles deep
where the terrain is cheap and about one tile deep where it is expensive -- so the repeat visits should be concentrated in the cheap tiles.
usage: visits_by_cost.py [size] [seed] [lib=build/exp/libufm_timing.so] [name=value ...]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, ufm_amd
pos = [a for a in sys.argv[1:] if "=" not in a]
kv = dict(a.split("=", 1) for a in sys.argv[1:] if "=" in a)
size = int(pos[0]) if pos else 4096
seed = int(pos[1]) if len(pos) > 1 else 7
ufm_amd.use_library(os.path.join(ROOT, kv.pop("lib", "build/exp/libufm_timing.so")))
L = ufm_amd.load_library()
L.ufm_debug_tiles.argtypes = [C.c_void_p, C.c_int]
cost = ufm_amd.synth.cost_map(seed, size, size)
start, goal = ufm_amd.synth.start_goal(size, size)
p = ufm_amd.Planner(ufm_amd.ALGO_FD, 1, False)
p.set_occupancy_threshold(1)
for k, v in kv.items():
    p.set_param(k, float(v))
T = L.ufm_tile_edge()
TX = TY = (size + 1 + T - 1) // T
NT = TX * TY
for rep in range(2):
    p.set_map(cost); p.reset(); p.set_start(*start); p.set_goal(*goal)
    assert p.step() == 0
buf = np.zeros((5, NT), np.uint32)
assert L.ufm_debug_tiles(buf.ctypes.data, NT) == NT
vis = buf[3].reshape(TX, TY).astype(np.float64)
# mean cost of the cells a node tile reads (cells tx*T-1 .. tx*T+T-1): obstacles (255) left out
c = np.where(cost >= 255, np.nan, cost.astype(np.float64))
cp = np.full((TX * T, TY * T), np.nan); cp[:size, :size] = c
cm = np.nanmean(cp.reshape(TX, T, TY, T), axis=(1, 3))
ok = (vis > 0) & np.isfinite(cm)
print("%s: tiles %d, visits %.0f (%.2f per tile)" % (kv, ok.sum(), vis[ok].sum(), vis[ok].mean()))
edges = [1, 10, 20, 35, 50, 70, 90, 110, 130, 150, 175, 201]
print("mean cost of tile   tiles   visits/tile   share of all visits   band depth in tiles (2.5 x 16 x map mean / (16 x tile mean))")
mean_map = np.nanmean(c)
for lo, hi in zip(edges[:-1], edges[1:]):
    s = ok & (cm >= lo) & (cm < hi)
    if s.sum():
        print("  %3d .. %3d       %6d      %5.2f            %4.1f %%                %5.1f" % (lo, hi, s.sum(), vis[s].mean(), 100 * vis[s].sum() / vis[ok].sum(), 2.5 * mean_map / cm[s].mean()))
p.close()
