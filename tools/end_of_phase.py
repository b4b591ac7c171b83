"""Diagnostic (GPU box, -DUFM_TIMING build: build/exp/libufm_timing.so or lib=path): how long after the last tile visit does the resident plan
kernel end (workgroup 0's end-of-phase collect, ufm_relax.h), and what happened around the tiles at the start corner in the last microseconds --
activations, visits (which workgroup, how many sweeps), in-visit refreshes.  FD-1 full plan.
usage: end_of_phase.py size seed reps [lib=path] [events=1] [name=value ...]   (events=1: the event list of every run, not only the last)"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, ufm_amd
size, seed, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
kv = dict(a.split("=", 1) for a in sys.argv[4:])
ufm_amd.use_library(os.path.join(ROOT, kv.pop("lib", "build/exp/libufm_timing.so")))
events = kv.pop("events", "0") != "0"
L = ufm_amd.load_library()
L.ufm_debug_plog.argtypes = [C.c_void_p, C.c_int]
L.ufm_debug_visits.argtypes = [C.c_void_p, C.c_int]
cost = ufm_amd.synth.cost_map(seed, size, size)
start, goal = ufm_amd.synth.start_goal(size, size)
p = ufm_amd.Planner(ufm_amd.ALGO_FD, 1, False)
p.set_occupancy_threshold(1)
for k, v in kv.items():
    p.set_param(k, float(v))
p.set_map(cost)
T = L.ufm_tile_edge(); TY = (size + 1 + T - 1) // T
corner = {0: "(0,0)", 1: "(0,1)", TY: "(1,0)", TY + 1: "(1,1)"}
def dump(tag, events):
    pb = np.zeros((1 << 14, 4), np.uint32); n = L.ufm_debug_plog(pb.ctypes.data, 1 << 14)
    vb = np.zeros((1 << 20, 5), np.uint32); nv = L.ufm_debug_visits(vb.ctypes.data, 1 << 20)
    ev = []
    for t, a, b, c in pb[:n]:
        if b == 0xFFFFFFFF: ev.append((int(t), "WG0 END"))
        elif b == 0xFFFFFFFD: ev.append((int(t), "WG0 look best %s" % ("-" if c == 0xFFFFFFFF else "%.3f" % np.uint32(c).view(np.float32))))
        elif b == 0xFFFFFFFB: ev.append((int(t), "   end of a visit of %s by workgroup %d: converged %d, most sweeps of a wave %d" % (corner.get(int(a), a), c >> 20, c & 1, (c >> 8) & 0xFFF)))
        elif b == 0xFFFFFFFE: ev.append((int(t), "refresh in %s: exchange returned %s" % (corner.get(int(a), a), ("mark" if c >= 0x7F800000 else "%.3f" % np.uint32(c).view(np.float32)))))
        else: ev.append((int(t), "push %s -> %s prio %.3f" % (corner.get(int(a), "tile %d (%d,%d)" % (a, a // TY, a % TY)), corner.get(int(b), b), np.uint32(c).view(np.float32))))
    for gt, s, e, pt, fr in vb[:nv]:
        if int(gt) in corner:
            ev.append((int(s), "visit of %s starts (activation sent at %d by %s)" % (corner[int(gt)], pt, corner.get(int(fr), fr))))
            ev.append((int(e), "visit of %s ends" % corner[int(gt)]))
    allend = int(vb[:nv, 2].max())
    ev.sort(key=lambda x: x[0])
    tend = max(e[0] for e in ev)
    looks = [t for t, s_ in ev if s_.startswith("WG0")]
    print("last visit of any tile ends at %.1f us, WG0's last looks end at %s us; END - last visit = %.1f us" % (allend / 100.0, [round(t / 100.0, 1) for t in looks[-8:]], (max(looks) - allend) / 100.0))
    if not events: return
    print("---- %s: events around the start corner (time in us before the last event)" % tag)
    for t, s in ev:
        if int(tend) - int(t) < 3000000 and not (s.startswith("WG0 look") and int(tend) - int(t) > 30000): print("  %9.2f  %s" % ((int(t) - int(tend)) / 100.0, s))
for rep in range(reps):
    p.reset(); p.set_start(*start); p.set_goal(*goal)
    assert p.step() == 0
    ci = p.check_info()
    assert tuple(ci[1:4]) == (0, 0, 0), ci
    dump("run %d, resident kernel %.2f ms, check_info %s" % (rep, p.stats.resident_kernel_ms, tuple(ci)), events or rep == reps - 1)
