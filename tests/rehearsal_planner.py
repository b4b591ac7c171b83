"""CPU stand-in planners for tests/bench_rehearsal.py (tests only): the oracle behind
the Python surface of ufm_amd.Planner / ufm_amd.BatchPlanner, so that the launcher, the process group, the
broadcasts and the reductions of bench.py can run end to end without a GPU."""
import numpy as np

import oracle_py as orc

_ALGO = {"FD": orc.ALGO_FD, "SG": orc.ALGO_SG, "DFM": orc.ALGO_DFM}


class Single:
    def __init__(self, algo, lvl, heuristic):
        self.o = orc.OraclePlanner(_ALGO[algo], lvl, heuristic)
        self.cost = None
        self.num_nodes_expanded = 0

    def reset(self): self.o.reset()
    def set_occupancy_threshold(self, t): self.o.set_occupancy_threshold(t)
    def set_heuristic_multiplier(self, m): self.o.set_heuristic_multiplier(m)

    def set_map(self, m):
        self.cost = np.array(m, dtype=np.uint8)
        self.o.set_map(self.cost)

    def patch_map(self, patch, x, y):
        patch = np.asarray(patch, dtype=np.uint8)
        self.cost[x:x + patch.shape[0], y:y + patch.shape[1]] = patch
        self.o.patch_map(patch, x, y)

    def set_start(self, x, y): self.o.set_start(x, y)
    def set_goal(self, x, y): self.o.set_goal(x, y)

    def step(self):
        rc = self.o.step()
        self.num_nodes_expanded = self.o.num_expanded
        return rc

    def read_map(self, width, length):
        return self.cost.copy()


class Batch:
    def __init__(self, algo, lvl, heuristic, n):
        self.maps = [Single(algo, lvl, heuristic) for _ in range(n)]
        self.dirty = [True] * n
        self.num_nodes_expanded = 0

    def set_occupancy_threshold(self, t):
        for p in self.maps: p.set_occupancy_threshold(t)

    def set_heuristic_multiplier(self, m):
        for p in self.maps: p.set_heuristic_multiplier(m)

    def set_map(self, i, m): self.maps[i].set_map(m)
    def patch_map(self, i, patch, x, y): self.maps[i].patch_map(patch, x, y)
    def set_start(self, i, x, y): self.maps[i].set_start(x, y)
    def set_goal(self, i, x, y): self.maps[i].set_goal(x, y)
    def reset(self, i): self.maps[i].reset()

    def step(self):
        self.num_nodes_expanded = 0
        for p in self.maps:
            rc = p.step()
            if rc != 0:
                return rc
            self.num_nodes_expanded += p.num_nodes_expanded
        return 0

    def read_map(self, i, width, length):
        return self.maps[i].read_map(width, length)


def make(kind, algo, lvl, heuristic, n_maps):
    return Single(algo, lvl, heuristic) if kind == "single" else Batch(algo, lvl, heuristic, n_maps)
