"""The header-only C++ mirror of the reference planner surface (unige-tasi-path-planners_amd/include)
compiles against include/ufm.h (CPU) and, on a GPU, a driver written like the reference's
Tests/Planners/FDSTAR/main.cpp produces the same numbers as the C-ABI called from Python."""
import os
import re
import struct
import subprocess

import numpy as np
import pytest

import ufm_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "unige-tasi-path-planners_amd")
EXE = os.path.join(ROOT, "build", "drop_in_driver")


def _build():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    if not os.path.exists(ufm_amd.library_path()):
        ufm_amd.build_library()
    subprocess.check_call([
        "g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
        "-I" + os.path.join(PKG, "include"), os.path.join(ROOT, "tests", "cpp", "drop_in_driver.cpp"),
        "-o", EXE, "-L" + PKG, "-lufm", "-Wl,-rpath," + PKG])


def test_driver_compiles_against_the_surface():
    _build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
@pytest.mark.parametrize("algo", ["FD", "SG", "DFM"])
def test_driver_matches_python_binding(tmp_path, algo):
    _build()
    size, seed, n = 160, 13, 4
    cost = ufm_amd.synth.cost_map(seed, size, size)
    script = list(ufm_amd.synth.replan_script(seed, size, size, n_patches=n))
    (tmp_path / "map.raw").write_bytes(cost.tobytes())
    with open(tmp_path / "patches.raw", "wb") as f:
        for k, s, top, left, patch in script:
            f.write(struct.pack("<4i", top, left, int(s[0]), int(s[1])))
            f.write(patch.tobytes())
    out = subprocess.check_output([EXE, algo, str(size), str(tmp_path / "map.raw"), str(n), str(tmp_path / "patches.raw")], text=True)
    lines = out.strip().splitlines()

    ids = {"FD": (ufm_amd.ALGO_FD, 1), "SG": (ufm_amd.ALGO_SG, 2), "DFM": (ufm_amd.ALGO_DFM, 1)}[algo]
    p = ufm_amd.Planner(*ids)
    p.reset(); p.set_occupancy_threshold(1); p.set_map(cost)
    start, goal = ufm_amd.synth.start_goal(size, size)
    p.set_start(*start); p.set_goal(*goal)
    assert p.step() == 0
    m = re.match(r"plan expanded (\d+) g_start (\S+) cost_start (\S+)", lines[0])
    assert int(m.group(1)) == p.num_nodes_expanded
    assert np.float32(m.group(2)) == p.g()[8, 8]
    cur = cost.copy()
    for i, (k, s, top, left, patch) in enumerate(script):
        p.patch_map(patch, top, left); p.set_start(*s)
        cur[top:top + 31, left:left + 31] = patch
        assert p.step() == 0
        m = re.match(r"replan (\d+) updated (\d+) rhs_start (\S+) interp (\S+) consistent (\d) patched_cost (\S+)", lines[1 + i])
        assert int(m.group(2)) == p.num_nodes_updated
        g = p.g()
        sx, sy = int(s[0]), int(s[1])
        assert np.float32(m.group(3)) == g[sx, sy]
        if algo != "DFM":
            assert np.float32(m.group(4)) == g[sx, sy]
        assert float(m.group(6)) == float(cur[top, left])     # the host-side Graph mirror followed the patch
    # the queue view (ReplannerBase::priority_queue, the mirror's PriorityQueue.h): ordered by key, every entry inconsistent and holding
    # the field's G, nothing below the start's key (DFM: within its tolerance)
    m = re.match(r"queue size (\d+) empty (\d) top_key (\S+) start_key (\S+) ordered (\d) inconsistent (\d) field_g (\d) top_is_first (\d) err (-?\d+)", lines[-2])
    assert m, lines[-2]
    assert (int(m.group(1)) == 0) == (int(m.group(2)) == 1)
    assert m.group(5) == m.group(6) == m.group(7) == m.group(8) == "1" and int(m.group(9)) == 0
    if int(m.group(1)):
        assert float(m.group(3)) >= float(m.group(4)) * (1 - (2e-6 if algo == "DFM" else 0.0))
    # the `tof` dump: map.size() == elements iterated over map.buckets == elements with a value when the field is probed through
    # get_g, in the driver's own process (elements beyond the start's key are not final: which of them hold a value is not
    # the same from run to run, so another run's count is only a sanity bound)
    m = re.match(r"dump size (\d+) iterated (\d+) sum_g (\S+) probed (\d+) probed_sum (\S+)", lines[-1])
    g = p.g()
    assert int(m.group(1)) == int(m.group(2)) == int(m.group(4))
    assert abs(float(m.group(3)) - float(m.group(5))) <= 1e-9 * float(m.group(3))
    assert abs(int(m.group(1)) - int(np.isfinite(g).sum())) <= 0.1 * int(m.group(1))
    p.close()


@pytest.mark.parametrize("driver", ["FDSTAR", "SGDFM", "DFM"])
@pytest.mark.parametrize("define", [[], ["-DNO_HEURISTIC"]])
def test_reference_drivers_compile_unmodified_against_the_mirror(driver, define):
    """The drop-in claim itself: the reference's own driver sources (Tests/Planners/*/main.cpp) are
    parsed and type-checked against the mirrored headers, unmodified (g++ -fsyntax-only: nothing is
    built or copied).  Only where the reference tree is mounted (not on the GPU box)."""
    src = os.path.join("/root/reference/Tests/Planners", driver, "main.cpp")
    if not os.path.exists(src):
        pytest.skip("reference tree not present")
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-Wall"] + define +
                          ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "include"), src])
