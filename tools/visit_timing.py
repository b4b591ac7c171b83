"""Diagnostic: where a tile visit spends its time (needs a libufm built with -DUFM_TIMING,
UFM_LIB=build/libufm_timing.so).  Headline workload: FD-1, 4096^2 plan (+ optional replans)."""
import argparse
import ctypes as C
import sys
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ufm_amd

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=4096)
ap.add_argument("--algo", default="FD")
ap.add_argument("--replans", type=int, default=0)
ap.add_argument("--param", action="append", default=[], help="name=value for ufm_set_param")
a = ap.parse_args()
algo = {"FD": (ufm_amd.ALGO_FD, 1), "SG": (ufm_amd.ALGO_SG, 2), "DFM": (ufm_amd.ALGO_DFM, 1)}[a.algo]
L = ufm_amd.load_library()
L.ufm_debug_tdiag.argtypes = [C.c_void_p, C.c_int]
cost = ufm_amd.synth.cost_map(7, a.size, a.size)
start, goal = ufm_amd.synth.start_goal(a.size, a.size)


def report(tag, p):
    d = (C.c_ulonglong * 64)()
    L.ufm_debug_tdiag(d, 1)
    v = max(d[3], 1)
    print("%s: launches %d kernel_ms %.2f visits %d | per visit: stage %.2f us  sweeps %.2f us  writeback %.2f us  total %.2f us" % (
        tag, p.stats.launches, p.stats.kernel_ms, d[3], d[0] / v / 100, d[1] / v / 100, d[2] / v / 100, (d[0] + d[1] + d[2]) / v / 100))
    print("   visit-time histogram (2 us bins): " + " ".join("%d" % d[8 + i] for i in range(32)))
    print("   per-wave sweep-count histogram:   " + " ".join("%d" % d[40 + i] for i in range(24)))


L.ufm_debug_trace.argtypes = [C.c_void_p, C.c_int]


def trace_report():
    buf = np.zeros(4 * 16384, np.uint64)
    n = L.ufm_debug_trace(buf.ctypes.data, 16384)
    if n <= 0:
        return
    r = buf[:4 * n].reshape(n, 4)
    k = (r[:, 0] & 0xFFFF).astype(int)
    blk = ((r[:, 0] >> 16) & 0xFFFFFF).astype(int)
    kind = (r[:, 0] >> 40).astype(int)
    for kk in sorted(set(k)):
        sel = k == kk
        t0 = r[sel, 1].min()
        v = sel & (kind == 0)
        b = sel & (kind == 1)
        dur = (r[v, 2] - r[v, 1]) / 100.0
        end = (r[sel, 2].max() - t0) / 100.0
        vis_end = (r[v, 2].max() - t0) / 100.0 if v.any() else 0
        per_blk = np.bincount(blk[v], minlength=1)
        busy = np.bincount(blk[v], weights=dur)
        first_start = (r[v, 1] - t0) / 100.0
        print("  launch %d: %d visits on %d blocks (max %d per block) | span %.1f us (last visit ends %.1f) | visit us: mean %.1f max %.1f | busiest block %.1f us | "
              "visit starts: median %.1f p90 %.1f max %.1f | sweeps of the 5 longest: %s" % (
                  kk, int(v.sum()), int((per_blk > 0).sum()), int(per_blk.max()), end, vis_end, dur.mean(), dur.max(), busy.max(),
                  np.median(first_start), np.percentile(first_start, 90), first_start.max(),
                  list((r[v, 3] & 255)[np.argsort(dur)[-5:]].astype(int))))
        w = r[v, 3]
        sw, seen, hint, n0, n1 = (w & 255).astype(int), ((w >> 8) & 255).astype(int), ((w >> 16) & 255).astype(int), ((w >> 24) & 0xFFFF).astype(int), ((w >> 40) & 0xFFF).astype(int)
        rank = ((w >> 52) & 0xFF).astype(int)
        print("      earlier visits of the same tile in this step (0,1,2,..,>=9): %s | sweeps by earlier visits: %s" % (
            np.bincount(np.minimum(seen, 9), minlength=10).tolist(),
            [round(float(sw[np.minimum(seen, 9) == v].mean()), 1) if (np.minimum(seen, 9) == v).any() else 0 for v in range(10)]))
        first = seen == 0
        for lo_, hi_ in ((0, 64), (64, 128), (128, 192), (192, 256)):
            sel = first & (rank >= lo_) & (rank < hi_)
            if sel.any():
                print("      first visits with band position %3d..%3d: %4d, mean %.1f us, over 14 us: %d" % (lo_, hi_, int(sel.sum()), dur[sel].mean(), int((dur[sel] > 14).sum())))
        long_ = dur > 14.0
        def pr(name, pred):
            tp = int((pred & long_).sum()); fp = int((pred & ~long_).sum()); fn = int((~pred & long_).sum())
            print("      predictor %-34s: flags %4d | long caught %3d missed %3d | false alarms %4d" % (name, int(pred.sum()), tp, fn, fp))
        print("      long visits (>14 us): %d of %d" % (int(long_.sum()), len(dur)))
        pr("first visit or hint>=12 (current)", (seen == 0) | (hint >= 12))
        pr("ninf0 > 0", n0 > 0)
        pr("ninf0 >= 32", n0 >= 32)
        pr("ninf0 >= 32 or hint>=12", (n0 >= 32) | (hint >= 12))
        pr("ninf0 >= 8 or hint>=8", (n0 >= 8) | (hint >= 8))
        missed = long_ & ~((n0 >= 32) | (hint >= 12))
        if missed.any():
            print("      missed examples (sweeps, seen, hint, ninf0, ninf1): " + str([(int(sw[i]), int(seen[i]), int(hint[i]), int(n0[i]), int(n1[i])) for i in np.nonzero(missed)[0][:8]]))


L.ufm_debug_wtrace.argtypes = [C.c_void_p, C.c_void_p]


def wave_report():
    buf = np.zeros(16 * 256 * 2, np.uint64)
    cnt = np.zeros(16, np.uint32)
    if L.ufm_debug_wtrace(buf.ctypes.data, cnt.ctypes.data) != 0 or cnt.max() == 0:
        return
    r = buf.reshape(16, 256, 2)
    t0 = min(int(r[w, 0, 1]) for w in range(16) if cnt[w])
    names = {0: "start", 1: "burst", 2: "done", 3: "idle", 4: "woken", 5: "vote"}
    print("  per-wave timeline of one visit (us since visit start; burst(bits) ... done(total sweeps so far)):")
    for w in range(16):
        ev = []
        for i in range(min(int(cnt[w]), 256)):
            ty, val, t = int(r[w, i, 0]) >> 32, int(r[w, i, 0]) & 0xFFFFFFFF, (int(r[w, i, 1]) - t0) / 100.0
            ev.append("%s%s@%.2f" % (names.get(ty, "?"), "(%d)" % val if ty in (0, 2) else "", t))
        print("   wave %2d (patch %d,%d): %s" % (w, w >> 2, w & 3, " ".join(ev)))


for rep in range(2):
    p = ufm_amd.Planner(*algo)
    p.set_profiling(1)
    for kv in a.param:
        p.set_param(kv.split('=')[0], float(kv.split('=')[1]))
    p.set_occupancy_threshold(1)
    p.set_map(cost)
    p.set_start(*start)
    p.set_goal(*goal)
    L.ufm_debug_tdiag(None, 1)
    assert p.step() == 0
    report("plan", p)
    trace_report()
    if rep == 1:
        wave_report()
    if a.replans and rep == 1:
        for k, s, top, left, patch in ufm_amd.synth.replan_script(7, a.size, a.size, n_patches=a.replans):
            p.patch_map(patch, top, left)
            p.set_start(*s)
            assert p.step() == 0
        report("%d replans" % a.replans, p)
    p.close()
