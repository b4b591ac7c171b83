"""ctypes binding of include/ufm.h (libufm.so).  No CPU fallback."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ALGO_FD, ALGO_SG, ALGO_DFM = 0, 1, 2
LOOP_OK, LOOP_FAILURE_NO_GRAPH, LOOP_FAILURE_NO_GOAL = 0, -1, -2

# every symbol include/ufm.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = [
    "ufm_create", "ufm_destroy", "ufm_reset", "ufm_set_occupancy_threshold",
    "ufm_set_heuristic_multiplier", "ufm_set_map", "ufm_patch_map", "ufm_set_start",
    "ufm_set_goal", "ufm_step", "ufm_set_map_device", "ufm_patch_map_device",
    "ufm_field_dims", "ufm_read_field", "ufm_read_map", "ufm_set_param", "ufm_set_profiling", "ufm_stream",
    "ufm_version", "ufm_tile_edge", "ufm_batch_create", "ufm_batch_destroy", "ufm_batch_size",
    "ufm_batch_set_occupancy_threshold", "ufm_batch_set_map", "ufm_batch_patch_map",
    "ufm_batch_set_start", "ufm_batch_set_goal", "ufm_batch_reset", "ufm_batch_step",
    "ufm_batch_read_field", "ufm_extract_path", "ufm_batch_extract_path", "ufm_read_info", "ufm_read_info_derived",
    "ufm_check_layout", "ufm_batch_check_layout", "ufm_batch_set_param", "ufm_check_info", "ufm_batch_check_info",
    "ufm_batch_create_sharded", "ufm_batch_shards", "ufm_batch_set_heuristic_multiplier", "ufm_batch_set_map_device",
    "ufm_batch_patch_map_device", "ufm_batch_read_map", "ufm_batch_set_profiling", "ufm_batch_stream", "ufm_read_queue",
]


class UfmError(RuntimeError):
    pass


class Stats(C.Structure):
    _fields_ = [
        ("u_ms", C.c_float), ("p_ms", C.c_float),
        ("updated", C.c_uint64), ("expanded", C.c_uint64),
        ("tile_visits", C.c_uint64), ("tile_iters", C.c_uint64), ("elem_evals", C.c_uint64),
        ("launches", C.c_uint32), ("raise_launches", C.c_uint32),
        ("kernel_ms", C.c_float),
        ("crit_sweeps", C.c_uint64),
        ("raise_tile_visits", C.c_uint64),
        ("raise_kernel_ms", C.c_float),
        ("queued_lower", C.c_uint32),
        ("queued_raise", C.c_uint32),
        ("timed_launches", C.c_uint32),
        ("timed_raise_launches", C.c_uint32),
        ("graphs_instantiated", C.c_uint32),
        ("region_replans", C.c_uint32),
        ("region_replans_done", C.c_uint32),
        ("resident_launches", C.c_uint32),
        ("resident_kernel_ms", C.c_float),
        ("resident_stops", C.c_uint32),
        ("resident_tile_visits", C.c_uint64),
        ("region_launches", C.c_uint32),
        ("region_timed", C.c_uint32),
        ("region_kernel_ms", C.c_float),
        ("region_tiles", C.c_uint32),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class PathInfo(C.Structure):
    """ufm_path_info (include/ufm.h)"""
    _fields_ = [
        ("n_points", C.c_int32), ("n_costs", C.c_int32),
        ("total_cost", C.c_float), ("total_dist", C.c_float),
        ("steps", C.c_int32), ("e_ms", C.c_float),
    ]


_LIB_PATH = os.path.join(_HERE, "libufm.so")


def library_path():
    return _LIB_PATH


def use_library(path):
    """Bind another build of the same library (tools/: kernel tuning experiments, diagnostic builds).  Explicit, per
    process, before the first planner is created -- the product, the tests and bench.py never call this."""
    global _LIB_PATH
    if _LIB is not None:
        raise UfmError("use_library() after the library has been loaded")
    _LIB_PATH = os.path.abspath(path)


def build_library():
    """hipcc cross-compiles gfx950 without a GPU (seconds)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def load_library():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = library_path()
    if not os.path.exists(so):
        raise UfmError("libufm.so is not built (run `make -C unige-tasi-path-planners_amd` "
                       "or __graft_entry__.build()); there is no CPU fallback")
    L = C.CDLL(so)
    vp, f, i = C.c_void_p, C.c_float, C.c_int
    L.ufm_create.argtypes = [C.POINTER(vp), i, i, i, i]
    L.ufm_destroy.argtypes = [vp]
    L.ufm_reset.argtypes = [vp]
    L.ufm_set_occupancy_threshold.argtypes = [vp, f]
    L.ufm_set_heuristic_multiplier.argtypes = [vp, f]
    L.ufm_set_map.argtypes = [vp, vp, i, i]
    L.ufm_set_map_device.argtypes = [vp, vp, i, i]
    L.ufm_patch_map.argtypes = [vp, vp, i, i, i, i]
    L.ufm_patch_map_device.argtypes = [vp, vp, i, i, i, i]
    L.ufm_set_start.argtypes = [vp, f, f]
    L.ufm_set_goal.argtypes = [vp, f, f]
    L.ufm_step.argtypes = [vp, C.POINTER(Stats)]
    L.ufm_field_dims.argtypes = [vp, C.POINTER(i), C.POINTER(i)]
    L.ufm_read_field.argtypes = [vp, i, i, i, i, vp, vp]
    L.ufm_read_map.argtypes = [vp, vp]
    L.ufm_check_layout.argtypes = [vp, vp, vp]
    L.ufm_batch_check_layout.argtypes = [vp, vp, vp]
    L.ufm_check_info.argtypes = [vp, vp]
    L.ufm_read_queue.argtypes = [vp, i, vp, vp, vp]
    L.ufm_batch_check_info.argtypes = [vp, vp]
    L.ufm_batch_set_param.argtypes = [vp, C.c_char_p, C.c_double]
    L.ufm_set_param.argtypes = [vp, C.c_char_p, C.c_double]
    L.ufm_set_profiling.argtypes = [vp, i]
    L.ufm_stream.argtypes = [vp]
    L.ufm_stream.restype = vp
    L.ufm_version.restype = C.c_char_p
    L.ufm_batch_create.argtypes = [C.POINTER(vp), i, i, i, i, i]
    L.ufm_batch_create_sharded.argtypes = [C.POINTER(vp), i, i, i, i, C.POINTER(i), i]
    L.ufm_batch_shards.argtypes = [vp]
    L.ufm_batch_set_heuristic_multiplier.argtypes = [vp, f]
    L.ufm_batch_set_map_device.argtypes = [vp, i, vp, i, i]
    L.ufm_batch_patch_map_device.argtypes = [vp, i, vp, i, i, i, i]
    L.ufm_batch_read_map.argtypes = [vp, i, vp]
    L.ufm_batch_set_profiling.argtypes = [vp, i]
    L.ufm_batch_stream.argtypes = [vp, i]
    L.ufm_batch_stream.restype = vp
    L.ufm_batch_destroy.argtypes = [vp]
    L.ufm_batch_size.argtypes = [vp]
    L.ufm_batch_set_occupancy_threshold.argtypes = [vp, f]
    L.ufm_batch_set_map.argtypes = [vp, i, vp, i, i]
    L.ufm_batch_patch_map.argtypes = [vp, i, vp, i, i, i, i]
    L.ufm_batch_set_start.argtypes = [vp, i, f, f]
    L.ufm_batch_set_goal.argtypes = [vp, i, f, f]
    L.ufm_batch_reset.argtypes = [vp, i]
    L.ufm_batch_step.argtypes = [vp, C.POINTER(Stats)]
    L.ufm_batch_read_field.argtypes = [vp, i, i, i, i, i, vp, vp]
    L.ufm_read_info.argtypes = [vp, i, i, i, i, vp]
    L.ufm_read_info_derived.argtypes = [vp, i, i, i, i, vp]
    L.ufm_extract_path.argtypes = [vp, i, i, i, vp, i, vp, i, C.POINTER(PathInfo)]
    L.ufm_batch_extract_path.argtypes = [vp, i, i, i, vp, i, vp, i, C.POINTER(PathInfo)]
    _LIB = L
    return L


def _chk(rc, what):
    if rc != 0:
        raise UfmError("%s failed with code %d" % (what, rc))


class Planner:
    """Mirror of the reference planner surface (ReplannerBase.h:39-123):
    reset / set_occupancy_threshold / set_heuristic_multiplier / set_map /
    patch_map / set_start / set_goal / step, public stats u_time, p_time,
    num_nodes_updated, num_nodes_expanded, and a dense field view in place of
    ExpandedMap::get_g / get_rhs."""

    def __init__(self, algo, opt_lvl=0, use_heuristic=False, device=0):
        self.L = load_library()
        h = C.c_void_p()
        _chk(self.L.ufm_create(C.byref(h), algo, opt_lvl, int(use_heuristic), device), "ufm_create")
        self.h = h
        self.algo = algo
        self.stats = Stats()
        self.u_time = 0.0
        self.p_time = 0.0
        self.num_nodes_updated = 0
        self.num_nodes_expanded = 0

    def close(self):
        if getattr(self, "h", None):
            self.L.ufm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        _chk(self.L.ufm_reset(self.h), "ufm_reset")

    def set_occupancy_threshold(self, t):
        _chk(self.L.ufm_set_occupancy_threshold(self.h, float(t)), "ufm_set_occupancy_threshold")

    def set_heuristic_multiplier(self, m):
        _chk(self.L.ufm_set_heuristic_multiplier(self.h, float(m)), "ufm_set_heuristic_multiplier")

    def set_map(self, m):
        m = np.ascontiguousarray(m, dtype=np.uint8)
        length, width = m.shape
        _chk(self.L.ufm_set_map(self.h, m.ctypes.data, width, length), "ufm_set_map")

    def set_map_device(self, dev_ptr, width, length):
        _chk(self.L.ufm_set_map_device(self.h, dev_ptr, width, length), "ufm_set_map_device")

    def patch_map(self, patch, x, y):
        patch = np.ascontiguousarray(patch, dtype=np.uint8)
        h, w = patch.shape
        _chk(self.L.ufm_patch_map(self.h, patch.ctypes.data, int(x), int(y), w, h), "ufm_patch_map")

    def patch_map_device(self, dev_ptr, x, y, w, h):
        _chk(self.L.ufm_patch_map_device(self.h, dev_ptr, int(x), int(y), int(w), int(h)), "ufm_patch_map_device")

    def patch_map_host(self, host_ptr, x, y, w, h):
        """ufm_patch_map with the address of a host buffer (a caller that keeps its patches in one array: no numpy view per call)"""
        _chk(self.L.ufm_patch_map(self.h, host_ptr, int(x), int(y), int(w), int(h)), "ufm_patch_map")

    def set_start(self, x, y):
        _chk(self.L.ufm_set_start(self.h, float(x), float(y)), "ufm_set_start")

    def set_goal(self, x, y):
        _chk(self.L.ufm_set_goal(self.h, float(x), float(y)), "ufm_set_goal")

    def set_param(self, name, value):
        _chk(self.L.ufm_set_param(self.h, name.encode(), float(value)), "ufm_set_param")

    def stream_ptr(self):
        """hipStream_t the engine's kernels run on (e.g. for torch.cuda.ExternalStream)"""
        return int(self.L.ufm_stream(self.h) or 0)

    def set_profiling(self, on):
        _chk(self.L.ufm_set_profiling(self.h, int(on)), "ufm_set_profiling")

    def step(self):
        rc = self.L.ufm_step(self.h, C.byref(self.stats))
        if rc in (LOOP_FAILURE_NO_GRAPH, LOOP_FAILURE_NO_GOAL):
            return rc
        _chk(rc, "ufm_step")
        self.u_time, self.p_time = self.stats.u_ms, self.stats.p_ms
        self.num_nodes_updated = self.stats.updated
        self.num_nodes_expanded = self.stats.expanded
        return rc

    def dims(self):
        a, b = C.c_int(), C.c_int()
        _chk(self.L.ufm_field_dims(self.h, C.byref(a), C.byref(b)), "ufm_field_dims")
        return a.value, b.value

    def read_field(self, x0=0, y0=0, nx=None, ny=None):
        ex, ey = self.dims()
        nx = ex - x0 if nx is None else nx
        ny = ey - y0 if ny is None else ny
        g = np.empty((nx, ny), dtype=np.float32)
        rhs = np.empty((nx, ny), dtype=np.float32)
        _chk(self.L.ufm_read_field(self.h, x0, y0, nx, ny, g.ctypes.data, rhs.ctypes.data), "ufm_read_field")
        return g, rhs

    def g(self):
        return self.read_field()[0]

    def read_info(self, x0=0, y0=0, nx=None, ny=None, derived=False):
        """back-pointers (the reference's INFO of level-1/2 planners): int32 [nx][ny][2] -- the stored ones, or
        (derived=True) those min_rhs<level>() derives from the field alone"""
        ex, ey = self.dims()
        nx = ex - x0 if nx is None else nx
        ny = ey - y0 if ny is None else ny
        out = np.empty((nx, ny, 2), np.int32)
        fn = self.L.ufm_read_info_derived if derived else self.L.ufm_read_info
        _chk(fn(self.h, x0, y0, nx, ny, out.ctypes.data), "ufm_read_info")
        return out

    def extract_path(self, max_steps=20, lookahead=True, allow_indirect=True):
        """LinearInterpolationPathExtractor::extract_path on the device:
        returns (points[n,2], step_costs[m], total_cost, total_dist); the call's info is kept in
        self.path_info."""
        cap_p, cap_c = 3 * max_steps + 1, 2 * max_steps
        pts = np.zeros((cap_p, 2), np.float32)
        costs = np.zeros(cap_c, np.float32)
        self.path_info = PathInfo()
        _chk(self.L.ufm_extract_path(self.h, int(max_steps), int(lookahead), int(allow_indirect),
                                     pts.ctypes.data, cap_p, costs.ctypes.data, cap_c,
                                     C.byref(self.path_info)), "ufm_extract_path")
        pi = self.path_info
        return pts[:pi.n_points].copy(), costs[:pi.n_costs].copy(), pi.total_cost, pi.total_dist

    def read_map(self, width, length):
        m = np.empty((length, width), dtype=np.uint8)
        _chk(self.L.ufm_read_map(self.h, m.ctypes.data), "ufm_read_map")
        return m

    def check_layout(self):
        """(ring entries, cost-window bytes) that differ from the values they copy; (0, 0) when sound"""
        bad = (C.c_uint64 * 2)()
        _chk(self.L.ufm_check_layout(self.h, C.addressof(bad), C.addressof(bad) + 8), "ufm_check_layout")
        return int(bad[0]), int(bad[1])


    def check_info(self):
        """stored back-pointers (node planners): (elements with a value, without a back-pointer, whose parent triangle does not
        give the value but a larger one, whose dependence bits are off, whose parent gives a smaller value -- waiting to be
        lowered, beyond the start's key --, unsupported ones at / beyond the start's key: queued invalidations); [1:4] are 0 when sound"""
        out = (C.c_uint64 * 6)()
        _chk(self.L.ufm_check_info(self.h, C.addressof(out)), "ufm_check_info")
        return tuple(int(v) for v in out)

    def read_queue(self, cap=None):
        """the reference's priority_queue as a caller could observe it between two steps: the elements that are not consistent.
        Returns (xy int32 [n, 2], g float32 [n], rhs float32 [n], total); n = min(total, cap), cap None: all of them"""
        total = C.c_int(0)
        if cap is None:
            _chk(self.L.ufm_read_queue(self.h, 0, None, None, C.addressof(total)), "ufm_read_queue")
            cap = total.value
        xy = np.zeros((max(cap, 1), 2), np.int32)
        gr = np.zeros((max(cap, 1), 2), np.float32)
        _chk(self.L.ufm_read_queue(self.h, cap, xy.ctypes.data, gr.ctypes.data, C.addressof(total)), "ufm_read_queue")
        n = min(total.value, cap)
        return xy[:n], gr[:n, 0].copy(), gr[:n, 1].copy(), total.value


class BatchPlanner:
    """Batch of independent, equally sized map instances: on one device, or (devices=[...]) spread over several
    in contiguous blocks, one engine per device inside the one handle."""

    def __init__(self, n_maps, algo, opt_lvl=0, use_heuristic=False, device=0, devices=None):
        self.L = load_library()
        h = C.c_void_p()
        if devices is None:
            _chk(self.L.ufm_batch_create(C.byref(h), n_maps, algo, opt_lvl, int(use_heuristic), device), "ufm_batch_create")
        else:
            arr = (C.c_int * len(devices))(*devices)
            _chk(self.L.ufm_batch_create_sharded(C.byref(h), n_maps, algo, opt_lvl, int(use_heuristic), arr, len(devices)),
                 "ufm_batch_create_sharded")
        self.h = h
        self.n = n_maps
        self.algo = algo
        self.stats = Stats()

    def close(self):
        if getattr(self, "h", None):
            self.L.ufm_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_occupancy_threshold(self, t):
        _chk(self.L.ufm_batch_set_occupancy_threshold(self.h, float(t)), "ufm_batch_set_occupancy_threshold")

    def set_map(self, i, m):
        m = np.ascontiguousarray(m, dtype=np.uint8)
        length, width = m.shape
        _chk(self.L.ufm_batch_set_map(self.h, i, m.ctypes.data, width, length), "ufm_batch_set_map")
        self._dims = (length + (0 if self.algo == ALGO_DFM else 1), width + (0 if self.algo == ALGO_DFM else 1))

    def set_map_device(self, i, dev_ptr, width, length):
        _chk(self.L.ufm_batch_set_map_device(self.h, i, dev_ptr, width, length), "ufm_batch_set_map_device")
        self._dims = (length + (0 if self.algo == ALGO_DFM else 1), width + (0 if self.algo == ALGO_DFM else 1))

    def patch_map(self, i, patch, x, y):
        patch = np.ascontiguousarray(patch, dtype=np.uint8)
        h, w = patch.shape
        _chk(self.L.ufm_batch_patch_map(self.h, i, patch.ctypes.data, int(x), int(y), w, h), "ufm_batch_patch_map")

    def patch_map_device(self, i, dev_ptr, x, y, w, h):
        _chk(self.L.ufm_batch_patch_map_device(self.h, i, dev_ptr, int(x), int(y), int(w), int(h)), "ufm_batch_patch_map_device")

    def set_heuristic_multiplier(self, m):
        _chk(self.L.ufm_batch_set_heuristic_multiplier(self.h, float(m)), "ufm_batch_set_heuristic_multiplier")

    def set_profiling(self, on):
        _chk(self.L.ufm_batch_set_profiling(self.h, int(on)), "ufm_batch_set_profiling")

    def stream_ptr(self, shard=0):
        return int(self.L.ufm_batch_stream(self.h, shard) or 0)

    def shards(self):
        return self.L.ufm_batch_shards(self.h)

    def read_map(self, i, width, length):
        m = np.empty((length, width), dtype=np.uint8)
        _chk(self.L.ufm_batch_read_map(self.h, i, m.ctypes.data), "ufm_batch_read_map")
        return m

    def set_start(self, i, x, y):
        _chk(self.L.ufm_batch_set_start(self.h, i, float(x), float(y)), "ufm_batch_set_start")

    def set_goal(self, i, x, y):
        _chk(self.L.ufm_batch_set_goal(self.h, i, float(x), float(y)), "ufm_batch_set_goal")

    def reset(self, i):
        _chk(self.L.ufm_batch_reset(self.h, i), "ufm_batch_reset")

    def step(self):
        rc = self.L.ufm_batch_step(self.h, C.byref(self.stats))
        if rc in (LOOP_FAILURE_NO_GRAPH, LOOP_FAILURE_NO_GOAL):
            return rc
        _chk(rc, "ufm_batch_step")
        return rc

    def extract_paths(self, max_steps=20, lookahead=True, allow_indirect=True):
        """One launch for all maps: list of (points, step_costs, total_cost, total_dist)."""
        n = self.L.ufm_batch_size(self.h)
        cap_p, cap_c = 3 * max_steps + 1, 2 * max_steps
        pts = np.zeros((n, cap_p, 2), np.float32)
        costs = np.zeros((n, cap_c), np.float32)
        info = (PathInfo * n)()
        _chk(self.L.ufm_batch_extract_path(self.h, int(max_steps), int(lookahead), int(allow_indirect),
                                           pts.ctypes.data, cap_p, costs.ctypes.data, cap_c, info),
             "ufm_batch_extract_path")
        self.path_info = info
        return [(pts[k, :info[k].n_points].copy(), costs[k, :info[k].n_costs].copy(),
                 info[k].total_cost, info[k].total_dist) for k in range(n)]

    def read_field(self, i):
        nx, ny = self._dims
        g = np.empty((nx, ny), dtype=np.float32)
        _chk(self.L.ufm_batch_read_field(self.h, i, 0, 0, nx, ny, g.ctypes.data, None), "ufm_batch_read_field")
        return g

    def set_param(self, name, value):
        _chk(self.L.ufm_batch_set_param(self.h, name.encode(), float(value)), "ufm_batch_set_param")

    def check_info(self):
        out = (C.c_uint64 * 6)()
        _chk(self.L.ufm_batch_check_info(self.h, C.addressof(out)), "ufm_batch_check_info")
        return tuple(int(v) for v in out)

    def check_layout(self):
        bad = (C.c_uint64 * 2)()
        _chk(self.L.ufm_batch_check_layout(self.h, C.addressof(bad), C.addressof(bad) + 8), "ufm_batch_check_layout")
        return int(bad[0]), int(bad[1])
