"""Helper of tests/test_strict_fences.py (run as a process of its own: a process binds one build of the library): FD-1 plans and replans
through the resident kernel's scheduler forms on the library given as argv[1]; prints one line per case with the SHA-1 of the whole
field (full-field mode: every element is final, so the field is one deterministic array)."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import ufm_amd  # noqa: E402

ufm_amd.use_library(sys.argv[1])
print("version", ufm_amd.load_library().ufm_version().decode())
size, seed = 768, 5
cost = ufm_amd.synth.cost_map(seed, size, size)
start, goal = ufm_amd.synth.start_goal(size, size)
script = list(ufm_amd.synth.replan_script(seed, size, size, n_patches=4))
for params in ({"owned_waves": 16}, {"owned_waves": 8}, {"owned_waves": 16, "owned_flags": 2 + 16}, {"owned_waves": 8, "owned_flags": 32}, {"owned_waves": 16, "owned_band": 1.0}):
    p = ufm_amd.Planner(ufm_amd.ALGO_FD, 1)
    p.set_param("focused", 0)
    for k, v in params.items():
        p.set_param(k, v)
    p.set_occupancy_threshold(1); p.set_map(cost); p.set_start(*start); p.set_goal(*goal)
    assert p.step() == 0 and p.stats.resident_launches == 1 and p.stats.resident_stops == 0
    h = hashlib.sha1(np.ascontiguousarray(p.g()).tobytes()).hexdigest()
    for k, s, top, left, patch in script:
        p.patch_map(patch, top, left); p.set_start(*s)
        assert p.step() == 0
    h2 = hashlib.sha1(np.ascontiguousarray(p.g()).tobytes()).hexdigest()
    assert p.check_layout() == (0, 0) and p.check_info()[1:4] == (0, 0, 0)
    print("case", sorted(params.items()), h, h2)
    p.close()
