"""-m gpu: bench.py's output contract on a small instance of the headline workload (the driver runs the
full-size one): one JSON line with the metric, the roofline object of the dominant kernel and the CPU
baseline, and the same line with the collectives switched on (one rank on RCCL)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
    base = ["--size", "1024", "--patches", "6", "--steps", "1", "--warmup", "1", "--cpu-size", "256"]
    for i in range(0, len(extra), 1):       # an explicit option overrides the default of the same name
        if extra[i] in base:
            j = base.index(extra[i]); del base[j:j + 2]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *base, *extra], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_single_gpu():
    d = _run()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["unit"] == "cells/s" and d["scaling"] == "weak" and d["dtype"] == "f32"
    assert d["value"] > 0 and d["vs_baseline"] is None
    # N = 1: `value` is SURVEY 8(d)'s metric -- the patches come from host memory through ufm_patch_map inside the timed region --, the
    # device-resident form (the N > 1 data path) is reported next to it
    assert d["config"]["patch_inputs"] == "host" and d["value_device_inputs"] > 0 and d["ms_per_step_device_inputs"] > 0
    assert d["phases"]["replans_ms_device_inputs"] > 0
    assert 0.5 < d["value_device_inputs"] / d["value"] < 2.0
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["avg_launch_us"] > 0 and r["timed_launches"] > 0
    assert "traffic_source" in r                      # the PMC traffic is a separately collected number and says so
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["cores_used"] == 1 and c["cores_host"] >= 1 and c["value"] > 0 and "sample" in c
    # the split that settles "GPU against one core" for the plan and for the replans from the line alone
    ph, cph = d["phases"], c["phases"]
    assert ph["plan_ms"] > 0 and ph["replans_ms"] > 0 and ph["set_map_ms"] > 0 and ph["replans"] == 6
    assert abs(ph["plan_ms"] + ph["replans_ms"] + ph["set_map_ms"] - d["ms_per_step"]) < 0.05 * d["ms_per_step"] + 0.2
    assert ph["plan_cells"] > 0 and ph["replans_cells"] > 0 and cph["plan_ms"] > 0 and cph["replans_ms"] > 0
    # the replans' cells in BOTH definitions on the CPU leg: the reference's queue pops, and elements whose G differs after the step (the GPU leg's count)
    assert cph["replans_cells"] > 0 and cph["replans_cells_changed"] > 0
    # ... and the replans' kernel has a roofline entry of its own
    rr = d["roofline_replans"]
    assert rr["bound"] == "hbm" and rr["launches"] == 6 and rr["timed_launches"] >= 1 and rr["avg_launch_us"] > 0
    assert abs(rr["frac"] - rr["achieved"] / rr["peak"]) < 1e-12 and rr["region_replans"] >= rr["region_replans_done"] > 0


def test_bench_line_with_collectives_one_rank():
    d = _run("--force-dist", "--no-cpu-baseline")        # RCCL: init, broadcast per replan (one ahead), all_reduce, barrier
    assert d["n_gpus"] == 1 and d["value"] > 0
    assert d["config"]["patch_inputs"] == "device" and "value_device_inputs" not in d      # with a process group the patches arrive in HBM
    d = _run("--force-dist", "--no-pipeline", "--no-cpu-baseline")
    assert d["value"] > 0


def test_bench_self_launcher_two_ranks_on_one_gpu():
    """`--gpus 2` without a launcher around it: bench.py starts both ranks itself; here they share the one GPU
    of the box and talk over gloo (RCCL needs a device per rank), the planner is the HIP engine."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--size", "1024", "--patches", "6",
                          "--steps", "1", "--warmup", "1"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and "cpu_baseline" not in d
    assert d["config"]["broadcasts_per_rank_per_episode"] == 6 and d["config"]["maps_total"] == 2


def test_bench_batch_config4_shape_small():
    """config 4's shape on one GPU at a small size: --batch maps in one handle, a patch stream per map, the
    process-parallel CPU leg with the host's core count in the line"""
    d = _run("--algo", "DFM", "--batch", "3", "--size", "512", "--cpu-size", "512", "--patches", "4")
    assert d["config"]["maps_per_gpu"] == 3 and d["value"] > 0 and d["roofline"]["frac"] > 0
    c = d["cpu_baseline"]
    assert c["cores_used"] == min(3, c["cores_host"]) and c["value"] > 0
