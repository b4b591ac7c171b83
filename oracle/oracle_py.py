"""ctypes binding of the CPU oracle (oracle/libufm_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product path never imports this module.
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ALGO_FD, ALGO_SG, ALGO_DFM = 0, 1, 2
REV_CURRENT, REV_START_CELL_FLOOR, REV_UPDATE_SKIPS_FAR_BORDER, REV_LOG = 0, 1, 2, 3   # ufm_oracle.h: the reference's sources as they stand / the two differences of the revision that wrote its mission logs / both
CASES = ("FD III (f <= 0)", "FD III (f^2 <= CATH(c,b) [sic])", "FD II (c > b)", "FD I", "FD A (c > b)", "FD B", "FD II (c <= b)", "FD A (c <= b)",
         "SG B", "SG II", "SG A")


def case_counts():
    """(evaluated, won) per case of compute_optimal_cost since the last reset: dicts keyed by CASES"""
    ev, won = (C.c_ulong * len(CASES))(), (C.c_ulong * len(CASES))()
    lib().orc_case_counts(ev, won)
    return dict(zip(CASES, list(ev))), dict(zip(CASES, list(won)))


def case_counts_reset():
    lib().orc_case_counts_reset()


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libufm_oracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        vp, f, i = C.c_void_p, C.c_float, C.c_int
        L.orc_create.restype = vp
        L.orc_create.argtypes = [i, i, i]
        L.orc_destroy.argtypes = [vp]
        L.orc_reset.argtypes = [vp]
        L.orc_set_revision.argtypes = [vp, i]
        L.orc_case_counts.argtypes = [C.POINTER(C.c_ulong), C.POINTER(C.c_ulong)]
        L.orc_set_occupancy_threshold.argtypes = [vp, f]
        L.orc_set_heuristic_multiplier.argtypes = [vp, f]
        L.orc_set_map.argtypes = [vp, C.c_void_p, i, i]
        L.orc_patch_map.argtypes = [vp, C.c_void_p, i, i, i, i]
        L.orc_set_start.argtypes = [vp, f, f]
        L.orc_set_goal.argtypes = [vp, f, f]
        L.orc_step.argtypes = [vp]
        L.orc_step.restype = i
        L.orc_field_dims.argtypes = [vp, C.POINTER(i), C.POINTER(i)]
        for n in ("orc_g", "orc_rhs", "orc_inmap", "orc_bptr"):
            getattr(L, n).restype = C.c_void_p
            getattr(L, n).argtypes = [vp]
        L.orc_track_changes.argtypes = [vp, i]
        for n in ("orc_num_expanded", "orc_num_updated", "orc_map_size", "orc_queue_size", "orc_num_changed"):
            getattr(L, n).restype = C.c_ulong
            getattr(L, n).argtypes = [vp]
        for n in ("orc_u_time_ms", "orc_p_time_ms"):
            getattr(L, n).restype = f
            getattr(L, n).argtypes = [vp]
        L.orc_top_key.argtypes = [vp, C.POINTER(f), C.POINTER(f)]
        pi, pf = C.POINTER(i), C.POINTER(f)
        L.orc_extract_path.restype = i
        L.orc_extract_path.argtypes = [vp, i, i, i, vp, i, vp, i, pi, pf, pf]
        L.orc_extract_path_field.restype = i
        L.orc_extract_path_field.argtypes = [vp, i, i, i, vp, i, i, i, f, f, f, f, i, i, i, vp, i, vp, i, pi, pf, pf]
        L.orc_threshold_uchar.restype = i
        L.orc_threshold_uchar.argtypes = [vp]
        L.orc_min_rhs_info.restype = f
        L.orc_min_rhs_info.argtypes = [vp, i, i, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.orc_load_g.argtypes = [vp, vp]
        L.orc_cost_via.restype = f
        L.orc_cost_via.argtypes = [vp, i, i, i, i, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        _LIB = L
    return _LIB


def _roundf(v):
    """C roundf: half away from zero (Node(Position) / Cell(Position), Node.cpp:14-17)"""
    return int(math.copysign(math.floor(abs(v) + 0.5), v))


class OraclePlanner:
    """Mirror of the reference planner surface (ReplannerBase.h:39-123)."""

    def __init__(self, algo, opt_lvl=0, use_heuristic=False, revision=REV_CURRENT):
        self.L = lib()
        self.h = self.L.orc_create(algo, opt_lvl, int(use_heuristic))
        if not self.h:
            raise ValueError("bad algo/opt_lvl")
        self.L.orc_set_revision(self.h, int(revision))
        self.algo, self.opt_lvl = algo, opt_lvl
        self.use_heuristic = bool(use_heuristic)
        self.hm = 1.0
        self.start = None

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_destroy(self.h)
            self.h = None

    def reset(self):
        self.L.orc_reset(self.h)

    def set_occupancy_threshold(self, t):
        self.L.orc_set_occupancy_threshold(self.h, float(t))

    def set_heuristic_multiplier(self, m):
        self.hm = float(m)
        self.L.orc_set_heuristic_multiplier(self.h, float(m))

    def set_map(self, m):
        m = np.ascontiguousarray(m, dtype=np.uint8)
        length, width = m.shape
        self.L.orc_set_map(self.h, m.ctypes.data, width, length)

    def patch_map(self, patch, x, y):
        patch = np.ascontiguousarray(patch, dtype=np.uint8)
        h, w = patch.shape
        self.L.orc_patch_map(self.h, patch.ctypes.data, int(x), int(y), w, h)

    def set_start(self, x, y):
        self.start = (float(x), float(y))
        self.L.orc_set_start(self.h, float(x), float(y))

    def set_goal(self, x, y):
        self.L.orc_set_goal(self.h, float(x), float(y))

    def step(self):
        return self.L.orc_step(self.h)

    # ---- field views (copies) ----
    def dims(self):
        a, b = C.c_int(), C.c_int()
        self.L.orc_field_dims(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def _arr(self, fn, dtype, mult=1):
        nx, ny = self.dims()
        ptr = fn(self.h)
        if not ptr:
            return None
        n = nx * ny * mult
        buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
        a = np.frombuffer(buf, dtype=dtype).copy()
        return a.reshape(nx, ny) if mult == 1 else a.reshape(nx, ny, mult)

    def g(self):
        return self._arr(self.L.orc_g, np.float32)

    def rhs(self):
        return self._arr(self.L.orc_rhs, np.float32)

    def inmap(self):
        return self._arr(self.L.orc_inmap, np.uint8)

    # ---- path extraction on this planner's RHS field ----
    def extract_path(self, max_steps=20, lookahead=True, allow_indirect=True):
        """LinearInterpolationPathExtractor::extract_path (PathExtraction impl:11-58):
        returns (points[n,2], step_costs[m], total_cost, total_dist)."""
        cap = 3 * max_steps + 1
        pts = np.zeros((cap, 2), np.float32)
        costs = np.zeros(2 * max_steps, np.float32)
        nc, tc, td = C.c_int(), C.c_float(), C.c_float()
        n = self.L.orc_extract_path(self.h, int(lookahead), int(max_steps), int(allow_indirect),
                                    pts.ctypes.data, cap, costs.ctypes.data, costs.size,
                                    C.byref(nc), C.byref(tc), C.byref(td))
        return pts[:n].copy(), costs[:nc.value].copy(), tc.value, td.value

    def load_g(self, g):
        g = np.ascontiguousarray(g, np.float32)
        assert g.shape == self.dims()
        self.L.orc_load_g(self.h, g.ctypes.data)

    def info_field(self):
        """min_rhs<level> back-pointers of every element from the current G field: int32 [nx][ny][2]"""
        nx, ny = self.dims()
        out = np.empty((nx, ny, 2), np.int32)
        a, b = C.c_int32(), C.c_int32()
        for x in range(nx):
            for y in range(ny):
                self.L.orc_min_rhs_info(self.h, x, y, C.byref(a), C.byref(b))
                out[x, y, 0], out[x, y, 1] = a.value, b.value
        return out

    def cost_via(self, x, y, bx, by):
        """(cost, b0, b1) through a given back-pointer on the current G field: node planners the node (bx, by) with its ccw
        neighbour, DFM the level-1 candidate built on the neighbour cell (bx, by)"""
        a, b = C.c_int32(), C.c_int32()
        c = self.L.orc_cost_via(self.h, int(x), int(y), int(bx), int(by), C.byref(a), C.byref(b))
        return float(np.float32(c)), a.value, b.value

    def threshold_uchar(self):
        return self.L.orc_threshold_uchar(self.h)

    @property
    def num_expanded(self):
        return self.L.orc_num_expanded(self.h)

    def track_changes(self, on=True):
        self.L.orc_track_changes(self.h, int(on))

    @property
    def num_changed(self):
        """elements whose G differs after the last step from before it (needs track_changes(); the engine's num_nodes_expanded)"""
        return self.L.orc_num_changed(self.h)

    @property
    def num_updated(self):
        return self.L.orc_num_updated(self.h)

    @property
    def map_size(self):
        return self.L.orc_map_size(self.h)

    @property
    def queue_size(self):
        return self.L.orc_queue_size(self.h)

    @property
    def u_time(self):
        return self.L.orc_u_time_ms(self.h)

    @property
    def p_time(self):
        return self.L.orc_p_time_ms(self.h)

    def top_key(self):
        a, b = C.c_float(), C.c_float()
        self.L.orc_top_key(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def _start_xy(self):
        sx, sy = self.start
        if self.algo == ALGO_DFM:
            sx, sy = float(_roundf(sx)), float(_roundf(sy))
        return sx, sy

    def key1(self, window=None):
        """first key component of every element (of the window (x0, x1, y0, y1)) for its current G: G (+ hm * dist(start, s))"""
        g = self.g()
        x0, x1, y0, y1 = window if window is not None else (0, g.shape[0], 0, g.shape[1])
        g = g[x0:x1, y0:y1]
        if not self.use_heuristic or self.start is None:
            return g
        sx, sy = self._start_xy()
        dx = np.float32(sx) - np.arange(x0, x1, dtype=np.float32)
        dy = np.float32(sy) - np.arange(y0, y1, dtype=np.float32)
        return (g + np.float32(self.hm) * np.hypot(dx[:, None], dy[None, :]).astype(np.float32)).astype(np.float32)

    def start_key(self):
        """the reference's max_start_key (first component): over the start elements that are reached"""
        g, rhs = self.g(), self.rhs()
        if self.start is None:
            return np.inf
        cx, cy = _roundf(self.start[0]), _roundf(self.start[1])
        elems = [(cx, cy)] if self.algo == ALGO_DFM else [(cx, cy), (cx + 1, cy), (cx, cy + 1), (cx + 1, cy + 1)]
        ks = []
        for x, y in elems:
            if 0 <= x < g.shape[0] and 0 <= y < g.shape[1] and np.isfinite(rhs[x, y]):
                # calculate_key uses min(g, rhs): a start corner may end with G = inf and a finite, final RHS
                # (FieldDPlanner_impl.h:165-186, 225-256)
                dist = np.float32(0.0)
                if self.use_heuristic:
                    sx, sy = self._start_xy()
                    dist = np.float32(np.hypot(np.float32(sx) - np.float32(x), np.float32(sy) - np.float32(y)))
                k = np.float32(min(g[x, y], rhs[x, y]) + np.float32(self.hm if self.use_heuristic else 0.0) * dist)
                ks.append(k)
        return max(ks) if ks else np.inf

    def trusted_mask(self, below_start_key=False, window=None):
        """Elements whose value the reference guarantees final after step(): locally consistent
        (G==RHS<inf) and, D*-Lite invariant, with key not beyond the top of the queue.
        below_start_key additionally restricts to keys below the start's key -- the set a
        planner that honours end_condition must have finalised.  window = (x0, x1, y0, y1): the mask of that
        part of the field only (the tests of the largest maps look at the neighbourhood of a replan)."""
        g, rhs = self.g(), self.rhs()
        if window is not None:
            x0, x1, y0, y1 = window
            g, rhs = g[x0:x1, y0:y1], rhs[x0:x1, y0:y1]
        k1t, k2t = self.top_key()
        k1 = self.key1(window)
        m = (g == rhs) & np.isfinite(g) & ((k1 < k1t) | ((k1 == k1t) & (g <= k2t if self.use_heuristic else True)))
        if below_start_key:
            m &= k1 < self.start_key()
        return m


def extract_path_field(rhs, cells, cost_map, thr_uchar, start, goal, max_steps=20, lookahead=True,
                       allow_indirect=True):
    """The oracle's extractor run on an arbitrary dense RHS field (e.g. one read back from the GPU
    engine): isolates extractor parity from field parity."""
    L = lib()
    rhs = np.ascontiguousarray(rhs, np.float32)
    cost_map = np.ascontiguousarray(cost_map, np.uint8)
    cap = 3 * max_steps + 1
    pts = np.zeros((cap, 2), np.float32)
    costs = np.zeros(2 * max_steps, np.float32)
    nc, tc, td = C.c_int(), C.c_float(), C.c_float()
    n = L.orc_extract_path_field(rhs.ctypes.data, rhs.shape[0], rhs.shape[1], int(cells),
                                 cost_map.ctypes.data, cost_map.shape[1], cost_map.shape[0], int(thr_uchar),
                                 float(start[0]), float(start[1]), float(goal[0]), float(goal[1]),
                                 int(lookahead), int(max_steps), int(allow_indirect),
                                 pts.ctypes.data, cap, costs.ctypes.data, costs.size,
                                 C.byref(nc), C.byref(tc), C.byref(td))
    return pts[:n].copy(), costs[:nc.value].copy(), tc.value, td.value
