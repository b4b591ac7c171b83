#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): grid cells updated / second over a full
plan + 100 replans on a 4096x4096 map, Field D* (level 1), per MI355X.

One "step" = one episode = set the map, plan from scratch, then 100 x (apply a
31x31 cost patch, move the start, replan).  Inputs (map, patches) are resident
in HBM before the timed region.  With N > 1 every rank owns an independent map
instance (weak scaling); the patch stream lives on rank 0 and reaches the other
ranks by an RCCL broadcast before every replan (the only exchange the path has).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

def bytes_per_tile_visit(tile):
    """9 B per element relaxation (4 B G read + 1 B cost + 4 B G write) + the halo"""
    return 9 * tile * tile + (4 * tile + 4) * 4
HBM_PEAK_GBS = 8000.0                                         # MI355X_MICROARCH.md: HBM3E peak


def cpu_baseline(size, seed, n_patches, algo_name="FD", heuristic=False):
    """The oracle (a C port of the reference's D*-Lite planners; FD-1 for the headline) timed on one
    host core on a bounded sample of the same workload.  Checker/baseline only."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as orc
    import ufm_amd
    try:
        os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[0]})   # reference pins to one core (main.cpp:36-40)
    except Exception:
        pass
    cost = ufm_amd.synth.cost_map(seed, size, size)
    start, goal = ufm_amd.synth.start_goal(size, size)
    oalgo = {"FD": orc.ALGO_FD, "SG": orc.ALGO_SG, "DFM": orc.ALGO_DFM}[algo_name]
    p = orc.OraclePlanner(oalgo, 2 if algo_name == "SG" else 1, heuristic)
    p.reset(); p.set_occupancy_threshold(1)
    if heuristic:
        p.set_heuristic_multiplier(float(cost.min()))
    p.set_map(cost); p.set_start(*start); p.set_goal(*goal)
    exp, ms = 0, 0.0
    assert p.step() == 0
    exp += p.num_expanded; ms += p.u_time + p.p_time
    for k, s, top, left, patch in ufm_amd.synth.replan_script(seed, size, size, n_patches=n_patches):
        p.patch_map(patch, top, left); p.set_start(*s)
        assert p.step() == 0
        exp += p.num_expanded; ms += p.u_time + p.p_time
    return {"value": exp / (ms * 1e-3), "unit": "cells/s", "cores": 1, "kind": "port",
            "sample": "%s-%d%s %dx%d seed %d, full plan + %d replans, %d expansions in %.1f s" % (
                algo_name, 2 if algo_name == "SG" else 1, " heuristic keys" if heuristic else "", size, size, seed, n_patches, exp, ms * 1e-3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--patches", type=int, default=100)
    ap.add_argument("--seed", type=int, default=7, help="seed of the synthetic map and patch script (SURVEY 8d: 7 for the headline, 42 for config 5)")
    ap.add_argument("--algo", default="FD", choices=["FD", "SG", "DFM"])
    ap.add_argument("--heuristic", action="store_true",
                    help="heuristic keys with hm = the map's smallest cost (planners built without -DNO_HEURISTIC; BASELINE config 5 with --size 8192)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-size", type=int, default=4096, help="map size of the CPU baseline sample (4096 = the identical workload, ~5 s)")
    ap.add_argument("--no-profile", action="store_true", help="skip per-launch HIP event timing")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1 (nccl = RCCL; gloo only to rehearse the control flow, staging through host memory)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="broadcast each patch right before its replan instead of one replan ahead")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and run the collectives even with one rank (rehearses the RCCL path on a one-GPU box)")
    args = ap.parse_args()

    import torch
    import ufm_amd
    BYTES_PER_TILE_VISIT = bytes_per_tile_visit(ufm_amd.load_library().ufm_tile_edge())

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # one process per GPU; (a gloo rehearsal on a single-GPU box folds the ranks onto the devices there are)
    dev_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", "29511"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group("gloo")
    else:
        dist = None
    dev = torch.device("cuda", dev_index)

    size, seed = args.size, args.seed
    algo = {"FD": ufm_amd.ALGO_FD, "SG": ufm_amd.ALGO_SG, "DFM": ufm_amd.ALGO_DFM}[args.algo]
    # independent map instance per rank; rank 0's instance is the BASELINE seed
    cost = ufm_amd.synth.cost_map(seed + 1000 * rank, size, size)
    start, goal = ufm_amd.synth.start_goal(size, size)
    script = list(ufm_amd.synth.replan_script(seed, size, size, n_patches=args.patches))
    d_cost = torch.from_numpy(cost).to(dev)
    psz = script[0][4].shape[0] if script else 31
    if rank == 0 and script:
        d_patches = torch.from_numpy(np.stack([s[4] for s in script])).to(dev)
    else:
        d_patches = torch.empty((max(1, len(script)), psz, psz), dtype=torch.uint8, device=dev)
    d_recv = [torch.empty((psz, psz), dtype=torch.uint8, device=dev) for _ in range(2)]

    planner = ufm_amd.Planner(algo, 1 if algo != ufm_amd.ALGO_SG else 2, bool(args.heuristic), device=dev_index)
    if args.heuristic:
        planner.set_heuristic_multiplier(float(cost.min()))
    planner.set_occupancy_threshold(1)
    planner.set_profiling(not args.no_profile)
    # HIP events on every 16th launch of a plan (~110 timed launches per run): the event packets cost ~4 us each,
    # with every 4th launch timed the episode was 3 % slower than untimed
    planner.set_param("profile_stride", 16)

    ep = ufm_amd.episode
    # The engine's HIP stream becomes torch's current stream: the collective's completion is then a
    # dependency of the engine's own stream (RCCL chains its internal stream to the current one), with no
    # further event hop and no host block between the broadcast and the patch kernel.  (Measured with one
    # rank on RCCL: 38 us per replan with a separate wait_stream() hop, 21 us this way.)
    try:
        eng_stream = torch.cuda.ExternalStream(planner.stream_ptr(), device=dev)
        torch.cuda.set_stream(eng_stream)
        patch_ready = None
    except Exception:   # no external-stream support in this torch build: block the host instead
        patch_ready = lambda: torch.cuda.current_stream().synchronize()
    stream = ep.PatchStream(d_patches if rank == 0 else None, d_recv, dist=dist, rank=rank, sync=patch_ready,
                            pipeline=not args.no_pipeline, count=len(script))
    meta = [(k, s, top, left) for (k, s, top, left, _) in script]

    ptr_cache = {}

    def apply_patch(p, buf, top, left):
        key = id(buf)
        ptr = ptr_cache.get(key)
        if ptr is None:
            ptr = ptr_cache[key] = buf.data_ptr()      # the per-patch views live as long as the stream
        p.patch_map_device(ptr, top, left, psz, psz)

    def step_stats(p):
        st = p.stats
        d = {"cells": st.expanded, "visits": st.tile_visits, "launches": st.launches, "kernel_ms": st.kernel_ms, "evals": st.elem_evals,
             "lower_visits": 0, "lower_launches": 0, "lower_kernel_ms": 0.0, "lower_timed": 0}
        timed = st.timed_launches - st.timed_raise_launches
        if timed > 0:   # steps whose lowering launches carry HIP events: the plans (a sample of their launches);
            #             the replans are replayed from a captured graph, whose nodes HIP events cannot time
            d.update(lower_visits=st.tile_visits - st.raise_tile_visits, lower_launches=st.launches - st.raise_launches,
                     lower_kernel_ms=st.kernel_ms - st.raise_kernel_ms, lower_timed=timed)
        return d

    def run_one():
        return ep.run_episode(
            planner,
            set_map=lambda p: p.set_map_device(d_cost.data_ptr(), size, size),
            start=start, goal=goal, script=meta, stream=stream,
            apply_patch=apply_patch,
            read_stats=step_stats)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    dt, per_step = ep.timed_episodes(run_one, args.steps, args.warmup, barrier)
    tot = [sum(d[k] for d in per_step) for k in ("cells", "visits", "launches", "kernel_ms", "evals")]
    low = [sum(d[k] for d in per_step) for k in ("lower_visits", "lower_launches", "lower_kernel_ms", "lower_timed")]

    # self-check of the patch path (outside the timed region): every rank's raster must now be its own
    # map with all the broadcast patches applied -- a collective that delivered late or wrong data fails here
    expect = cost.copy()
    for (_k, _s, top, left, patch) in script:
        expect[top:top + patch.shape[0], left:left + patch.shape[1]] = patch
    got = planner.read_map(size, size)
    if not np.array_equal(got, expect):
        bad = np.argwhere(got != expect)
        raise RuntimeError("rank %d: raster on the device differs from map + patches in %d cells, first at %r (device %d, expected %d)" % (
            rank, len(bad), tuple(bad[0]), got[tuple(bad[0])], expect[tuple(bad[0])]))

    cells, visits, launches, kms, evals = tot
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        c = torch.tensor([cells, visits, launches, evals], dtype=torch.float64, device=dev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        cells_all = float(c[0].item())
    else:
        cells_all = float(cells)

    if rank == 0:
        out = {
            "metric": "grid cells updated/sec (full plan + 100 replans), %d^2 map" % size,
            "value": cells_all / dt,
            "unit": "cells/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / max(1, args.steps),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "%s, %dx%d cost map (seed %d generator, SURVEY 8d), full plan + %d 31x31 patches with moving start, one map instance per GPU" % (
                    {"FD": "Field D* level-1", "SG": "Shifted-Grid FM level-2", "DFM": "MS-DFM level-1"}[args.algo], size, size, seed, len(script)),
                "algo": args.algo, "size": size, "patches": len(script), "heuristic_keys": bool(args.heuristic),
                "cells_per_step_rank0": cells / max(1, args.steps),
                "relax_launches_per_step_rank0": launches / max(1, args.steps),
                "tile_visits_per_step_rank0": visits / max(1, args.steps),
                "elem_evals_per_step_rank0": evals / max(1, args.steps),
            },
        }
        lvis, llaunch, lkms, ltimed = low
        if lkms > 0 and llaunch > 0 and ltimed > 0:
            # dominant kernel: k_relax<algo, LOWER>.  Algorithmic bytes = tile visits x (9 B per element
            # + halo) per SURVEY.md 8(d); duration = HIP events on the engine's stream around its launches.
            # (the events bracket a sample of the launches -- every 16th of a plan --
            # so that they can stay on inside the timed region: ltimed of the llaunch launches)
            avg_launch_s = lkms * 1e-3 / ltimed
            achieved = (lvis / llaunch) * BYTES_PER_TILE_VISIT / avg_launch_s / 1e9
            traffic = None
            tj = os.path.join(ROOT, "profiles", "r1_traffic.json")
            if os.path.exists(tj) and args.algo == "FD" and size == 4096:
                traffic = json.load(open(tj)).get("traffic_bytes_per_launch")   # rocprofv3 PMC, same command
            out["roofline"] = {
                "bound": "hbm", "kernel": "k_relax<%s,LOWER,cursor> (the lowering kernel as launched by the plans)" % args.algo, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "avg_launch_us": 1e6 * avg_launch_s, "timed_launches": ltimed, "launches": llaunch,
                "algorithmic_bytes_per_launch": lvis * BYTES_PER_TILE_VISIT / llaunch,
                "tile_visits_per_launch": lvis / llaunch,
                "kernel_time_share": avg_launch_s * llaunch / dt,
                "note": "latency-bound (dependent in-LDS sweeps along the wavefront), not bandwidth-bound: see DESIGN.md",
            }
        if not args.no_cpu_baseline and world == 1:   # the CPU baseline is reported at N = 1 only
            out["cpu_baseline"] = cpu_baseline(args.cpu_size, seed, args.patches, args.algo, bool(args.heuristic))
        print(json.dumps(out))
    stream.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
