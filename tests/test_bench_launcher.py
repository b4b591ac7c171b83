"""CPU: `bench.py --gpus 2` starts its two ranks itself (no torchrun), forms a gloo process group, broadcasts
every patch / every replan round from rank 0 and prints ONE line with n_gpus == 2.  The planner is the oracle
behind the Python planner surface (tests/rehearsal_planner.py), handed to bench.main() by tests/bench_rehearsal.py
-- the shipped benchmark has no import hook --; the same code path runs the HIP planner on RCCL on the GPU boxes.
A rank that fails ends the run within seconds, with its exit code and its traceback."""
import json
import os
import subprocess
import sys
import time

import numpy as np

import ufm_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), env.get("PYTHONPATH", "")])
    env.update(env_extra or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "bench_rehearsal.py"), "--backend", "gloo",
                          "--steps", "1", "--warmup", "1", *extra], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def _oracle_cells(algo, lvl, seed_map, seed_script, size, patches):
    import oracle_py as orc
    p = orc.OraclePlanner({"FD": orc.ALGO_FD, "DFM": orc.ALGO_DFM}[algo], lvl, False)
    cost = ufm_amd.synth.cost_map(seed_map, size, size)
    start, goal = ufm_amd.synth.start_goal(size, size)
    p.set_occupancy_threshold(1); p.set_map(cost); p.reset(); p.set_start(*start); p.set_goal(*goal)
    assert p.step() == 0
    n = p.num_expanded
    for k, s, top, left, patch in ufm_amd.synth.replan_script(seed_script, size, size, n_patches=patches):
        p.patch_map(patch, top, left); p.set_start(*s)
        assert p.step() == 0
        n += p.num_expanded
    return n


def test_self_launcher_two_ranks_single_map():
    size, patches = 96, 5
    d = _bench("--gpus", "2", "--size", str(size), "--patches", str(patches))
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "rehearsal" in d and "roofline" not in d and "cpu_baseline" not in d
    assert d["config"]["broadcasts_per_rank_per_episode"] == patches            # one broadcast per replan on every rank
    assert d["config"]["maps_total"] == 2
    # whole-job aggregate: rank r plans map seed 7 + 1000 r under rank 0's patch stream
    expect = sum(_oracle_cells("FD", 1, 7 + 1000 * r, 7, size, patches) for r in range(2))
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - expect) < 1e-6 * expect


def test_self_launcher_two_ranks_batch_with_per_map_patch_streams():
    size, patches, M = 80, 4, 3
    d = _bench("--gpus", "2", "--size", str(size), "--patches", str(patches), "--algo", "DFM", "--batch", str(M), "--no-pipeline")
    assert d["n_gpus"] == 2 and d["config"]["maps_per_gpu"] == M and d["config"]["maps_total"] == 2 * M
    assert d["config"]["broadcasts_per_rank_per_episode"] == patches            # one packed round per replan round
    # map g (global id) has the map AND the patch stream of seed 1000 + g
    expect = sum(_oracle_cells("DFM", 1, 1000 + g, 1000 + g, size, patches) for g in range(2 * M))
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - expect) < 1e-6 * expect


def test_a_failing_rank_ends_the_run_with_its_code_and_traceback():
    """Rank 1's planner cannot be made: it dies before the first collective.  The launcher must not sit in rank 0's
    communicate() while rank 0 waits in a broadcast for the backend's timeout (up to 30 min with gloo): it stops the
    other rank and exits non-zero within seconds, with rank 1's traceback on stderr."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "bench_rehearsal.py"), "--backend", "gloo", "--steps", "1", "--warmup", "0",
                          "--gpus", "2", "--size", "64", "--patches", "2", "--fail-rank", "1"], capture_output=True, text=True, timeout=120, env=env)
    dt = time.time() - t0
    assert out.returncode != 0
    assert dt < 20.0, "the launcher took %.1f s to notice the dead rank" % dt
    assert "rank 1 exited with code" in out.stderr and "fails on purpose" in out.stderr, out.stderr[-2000:]
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]      # no result line from a failed run


def test_the_launcher_timeout_stops_the_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "bench_rehearsal.py"), "--backend", "gloo", "--steps", "50", "--warmup", "0",
                          "--gpus", "2", "--size", "512", "--patches", "100", "--timeout", "3"], capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 124 and time.time() - t0 < 30.0
    assert "timeout after 3 s" in out.stderr


def test_one_rank_needs_no_process_group():
    d = _bench("--size", "64", "--patches", "3")
    assert d["n_gpus"] == 1 and d["config"]["broadcasts_per_rank_per_episode"] == 0


def test_pack_round_layout():
    ep = ufm_amd.episode
    patch = np.arange(9, dtype=np.uint8).reshape(3, 3)
    r = ep.pack_round([(5, 10, 20, patch), (6, 11, 21, patch + 1)], 3)
    assert r.shape == (2, 32) and r.dtype == np.uint8
    assert r[1, :16].view(np.int32).tolist() == [6, 11, 21, 3]
    assert np.array_equal(r[0, 16:25], patch.reshape(-1))
