#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): grid cells updated / second over a full
plan + 100 replans on a 4096x4096 map, Field D* (level 1), per MI355X.

One "step" = one episode = set the map, plan from scratch, then 100 x (apply a
31x31 cost patch, move the start, replan).  Inputs (map, patches) are resident
in HBM before the timed region.  With N > 1 every rank owns independent map
instances (weak scaling); the patch streams live on rank 0 and reach the other
ranks by one RCCL broadcast per replan (the only exchange the path has).

  python bench.py                                  config 3 (the headline), one GPU
  python bench.py --gpus 8                         the same on 8 GPUs: starts 8 ranks itself
  python bench.py --gpus 8 --algo DFM --size 2048 --batch 8 --seed 1000     config 4: 64 maps, 8 per GPU,
                                                   every map with a patch stream of its own
  python bench.py --size 8192 --seed 42 --heuristic                          config 5 (this GPU's replica)

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process starts N rank processes (fresh
interpreters, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set) before anything touches a GPU, watches all of
them -- the first one to fail ends the others within seconds, its stderr tail is relayed and its exit code is
this process's; `--timeout` bounds the whole run --, relays rank 0's line.  Under
`python -m torch.distributed.run` the ranks exist already and each process is one of them.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def bytes_per_tile_visit(tile):
    """9 B per element relaxation (4 B G read + 1 B cost + 4 B G write) + the halo"""
    return 9 * tile * tile + (4 * tile + 4) * 4


HBM_PEAK_GBS = 8000.0                                         # MI355X_MICROARCH.md: HBM3E peak
ALGO_LABEL = {"FD": "Field D* level-1", "SG": "Shifted-Grid FM level-2", "DFM": "MS-DFM level-1"}


def opt_level(algo_name):
    return 2 if algo_name == "SG" else 1


# ---------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (a C port of the reference's D*-Lite planners) on the host cores.  Checker /
# baseline only.  One map = one process pinned to one core, like the reference driver (main.cpp:36-40);
# a batch = one such process per map, side by side on as many cores as there are maps (or cores).
# ---------------------------------------------------------------------------------------------------------
def cpu_worker(spec):
    """Runs in a fresh process (`bench.py --cpu-worker JSON`): one map's episode on the oracle, one core."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as orc
    import ufm_amd
    try:
        cores = sorted(os.sched_getaffinity(0))
        os.sched_setaffinity(0, {cores[spec["core"] % len(cores)]})
    except Exception:
        pass
    size, seed, algo_name = spec["size"], spec["seed"], spec["algo"]
    cost = ufm_amd.synth.cost_map(seed, size, size)
    start, goal = ufm_amd.synth.start_goal(size, size)
    oalgo = {"FD": orc.ALGO_FD, "SG": orc.ALGO_SG, "DFM": orc.ALGO_DFM}[algo_name]
    p = orc.OraclePlanner(oalgo, opt_level(algo_name), spec["heuristic"])
    p.reset(); p.set_occupancy_threshold(1)
    if spec["heuristic"]:
        p.set_heuristic_multiplier(float(cost.min()))
    p.set_map(cost); p.set_start(*start); p.set_goal(*goal)
    exp, ms = 0, 0.0
    t0 = time.perf_counter()
    assert p.step() == 0
    exp += p.num_expanded; ms += p.u_time + p.p_time
    plan_exp, plan_ms = exp, ms
    # (the replans also count the elements whose G differs after the step -- the engine's definition of a cell updated, next to the reference's
    #  own count of queue pops: both are in the line.  The bookkeeping is one compare per G assignment, inside the replans' clock.)
    p.track_changes(True)
    changed = 0
    for k, s, top, left, patch in ufm_amd.synth.replan_script(seed, size, size, n_patches=spec["patches"]):
        p.patch_map(patch, top, left); p.set_start(*s)
        assert p.step() == 0
        exp += p.num_expanded; ms += p.u_time + p.p_time
        changed += p.num_changed
    print(json.dumps({"expanded": exp, "ms": ms, "wall_s": time.perf_counter() - t0, "plan_ms": plan_ms, "plan_expanded": plan_exp,
                      "replans_ms": ms - plan_ms, "replans_expanded": exp - plan_exp, "replans_changed": changed}))


def cpu_baseline(size, seed, n_patches, algo_name, heuristic, n_maps):
    cores_host = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    used = max(1, min(n_maps, cores_host))
    specs = [{"size": size, "seed": seed + m, "algo": algo_name, "heuristic": heuristic, "patches": n_patches, "core": m} for m in range(n_maps)]
    results, t0 = [], time.perf_counter()
    for lo in range(0, n_maps, used):       # waves of `used` processes when there are more maps than cores
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", json.dumps(sp)],
                                  stdout=subprocess.PIPE, text=True) for sp in specs[lo:lo + used]]
        for pr in procs:
            out, _ = pr.communicate()
            if pr.returncode != 0:
                raise RuntimeError("cpu baseline worker failed")
            results.append(json.loads(out.strip().splitlines()[-1]))
    wall = time.perf_counter() - t0
    exp = sum(r["expanded"] for r in results)
    if n_maps == 1:     # the reference's own accounting: sum of expansions / sum of (u_time + p_time)
        value, how = exp / (results[0]["ms"] * 1e-3), "%d expansions in %.1f s" % (exp, results[0]["ms"] * 1e-3)
    else:               # a batch: all maps' expansions over the time the slowest process needed
        slow = max(r["ms"] for r in results) * 1e-3
        value, how = exp / slow, "%d expansions, slowest process %.1f s (wall incl. start-up %.1f s)" % (exp, slow, wall)
    # the same split as the GPU line's "phases": the reference's own clocks (u_time + p_time) of the first step and of the replans,
    # the slowest process of a batch
    phases = {"plan_ms": max(r["plan_ms"] for r in results), "replans_ms": max(r["replans_ms"] for r in results),
              "plan_cells": sum(r["plan_expanded"] for r in results), "replans_cells": sum(r["replans_expanded"] for r in results),
              "replans_cells_changed": sum(r["replans_changed"] for r in results),
              "note": "cells = the reference's num_nodes_expanded (queue pops); replans_cells_changed = elements whose G differs after the step, the definition of the GPU line's phases.replans_cells"}
    return {"value": value, "unit": "cells/s", "cores": used, "cores_used": used, "cores_host": cores_host, "kind": "port", "phases": phases,
            "sample": "%s-%d%s %dx%d, %s, full plan + %d replans each: %s" % (
                algo_name, opt_level(algo_name), " heuristic keys" if heuristic else "", size, size,
                "seed %d" % seed if n_maps == 1 else "%d maps (seeds %d..%d), one pinned process per map on %d of %d host cores" % (
                    n_maps, seed, seed + n_maps - 1, used, cores_host), n_patches, how)}


# ---------------------------------------------------------------------------------------------------------
# launcher
# ---------------------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, script, argv, timeout_s, _attempt=0):
    """Start n rank processes of `script` (one per GPU) and watch ALL of them.  Nothing in this process has touched torch or the
    GPU; the children are fresh interpreters.  Rank 0's stdout is collected (the one JSON line), every rank's stderr goes to a
    file of its own.  The first rank that exits with a non-zero code -- or the timeout -- ends the others (terminate, then kill):
    a rank that died before a collective must not leave the rest waiting in it until the backend's own timeout.  Returns that
    code (124 for the timeout) after relaying the tail of the failing rank's stderr.  The rendezvous port is picked by binding port 0 and
    closing it again, which another launch on the machine can win: a rendezvous that fails with "address already in use" is started once
    more on another port (fresh children; not when MASTER_PORT was given)."""
    port_given = bool(os.environ.get("MASTER_PORT"))          # (set and not empty: one test for both decisions below)
    port = os.environ["MASTER_PORT"] if port_given else str(free_port())
    logdir = tempfile.mkdtemp(prefix="bench_ranks_")
    procs, errs = [], []
    out0 = open(os.path.join(logdir, "rank0.out"), "w+")
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        err = open(os.path.join(logdir, "bench_rank%d.err" % r), "w+")
        errs.append(err)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(script)] + list(argv), env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL, stderr=err))
    rc, failed = 0, None
    deadline = time.monotonic() + timeout_s
    live = set(range(n))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc, failed = code, r
        if rc != 0 or time.monotonic() > deadline:
            break
        time.sleep(0.05)
    if live:                                         # a rank failed, or the timeout: end the others
        if rc == 0:
            rc, failed = 124, None
        for r in live:
            procs[r].terminate()
        t_kill = time.monotonic() + 3.0
        for r in live:
            try:
                procs[r].wait(timeout=max(0.1, t_kill - time.monotonic()))
            except subprocess.TimeoutExpired:
                procs[r].kill()
                procs[r].wait()
    out0.seek(0)
    for line in out0.read().splitlines():      # the contract is ONE JSON line on stdout; whatever else rank 0 printed goes to stderr
        print(line, file=sys.stdout if line.startswith("{") and rc == 0 else sys.stderr)
    sys.stdout.flush()
    if rc != 0 and failed is not None and _attempt == 0 and not port_given:
        # (every rank's stderr: the rank that holds the EADDRINUSE message -- rank 0, which binds -- need not be the first one to exit)
        taken = False
        for e in errs:
            e.seek(0)
            taken = taken or "address already in use" in e.read().lower()
        if taken:
            for f in errs + [out0]:
                f.close()
            shutil.rmtree(logdir, ignore_errors=True)
            print("bench.py: rendezvous port %s was taken, starting the ranks once more on another one" % port, file=sys.stderr)
            return launch_ranks(n, script, argv, max(1.0, deadline - time.monotonic()), _attempt=1)
    if rc != 0:
        print("bench.py: %s; per-rank stderr in %s" % (
            "rank %d exited with code %d, the other ranks were stopped" % (failed, rc) if failed is not None else "timeout after %.0f s" % timeout_s, logdir),
            file=sys.stderr)
        for r in ([failed] if failed is not None else range(n)):
            errs[r].seek(0)
            tail = errs[r].read().splitlines()[-40:]
            print("---- rank %d stderr (tail) ----" % r, file=sys.stderr)
            print("\n".join(tail), file=sys.stderr)
    else:
        for r in range(n):                     # warnings of a good run: rank by rank, not interleaved
            errs[r].seek(0)
            text = errs[r].read().strip()
            if text:
                print("---- rank %d stderr ----\n%s" % (r, text), file=sys.stderr)
    for f in errs + [out0]:
        f.close()
    if rc == 0:
        shutil.rmtree(logdir, ignore_errors=True)     # (a failed run keeps its per-rank files: the message above names the directory)
    return rc


# ---------------------------------------------------------------------------------------------------------
# one rank
# ---------------------------------------------------------------------------------------------------------
def run_rank(args, planner_factory=None, factory_label=None):
    """planner_factory(kind, algo, lvl, heuristic, n_maps): a CPU stand-in planner instead of the HIP one -- the rehearsal of the
    multi-rank control flow in tests/bench_rehearsal.py (there is no command-line way to get here)."""
    import torch
    import ufm_amd
    ep = ufm_amd.episode

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    rehearsal = planner_factory is not None
    if rehearsal:
        factory = planner_factory
        dev, dev_index = torch.device("cpu"), 0
        if args.backend == "nccl":
            raise SystemExit("a stand-in planner runs on the CPU: use --backend gloo")
    else:
        # one process per GPU; (a gloo rehearsal on a single-GPU box folds the ranks onto the devices there are)
        dev_index = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(dev_index)
        dev = torch.device("cuda", dev_index)
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", "29511"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    else:
        dist = None

    size, seed = args.size, args.seed
    algo = {"FD": ufm_amd.ALGO_FD, "SG": ufm_amd.ALGO_SG, "DFM": ufm_amd.ALGO_DFM}[args.algo]
    lvl = opt_level(args.algo)
    M = args.batch
    n_local = max(1, M)
    start, goal = ufm_amd.synth.start_goal(size, size)
    tile = 16 if rehearsal else ufm_amd.load_library().ufm_tile_edge()
    BYTES_PER_TILE_VISIT = bytes_per_tile_visit(tile)

    def on_dev(a):
        t = torch.from_numpy(np.ascontiguousarray(a))
        return t if rehearsal else t.to(dev)

    def step_stats(p):
        """inside the timed region: a byte copy of the step's statistics struct (evaluated by stats_dict() afterwards)"""
        if rehearsal:
            return {"cells": p.num_nodes_expanded}
        return bytes(p.stats)

    def stats_dict(snapshot):
        st = ufm_amd.capi.Stats.from_buffer_copy(snapshot)
        d = {"cells": st.expanded, "visits": st.tile_visits, "launches": st.launches, "kernel_ms": st.kernel_ms, "evals": st.elem_evals,
             "lower_visits": 0, "lower_launches": 0, "lower_kernel_ms": 0.0, "lower_timed": 0,
             "res_visits": 0, "res_launches": 0, "res_kernel_ms": 0.0,
             "reg_launches": st.region_launches, "reg_timed": st.region_timed, "reg_kernel_ms": st.region_kernel_ms, "reg_tiles": st.region_tiles,
             "reg_cells": st.expanded if st.region_launches else 0}
        if st.resident_launches and st.resident_kernel_ms > 0:   # a plan: its lowering phase was one launch of the resident kernel
            d.update(res_visits=st.resident_tile_visits, res_launches=st.resident_launches, res_kernel_ms=st.resident_kernel_ms)
        timed = st.timed_launches - st.timed_raise_launches
        if timed > 0:   # steps whose lowering launches carry HIP events: the plans (a sample of their launches);
            #             the replans are replayed from a captured graph, whose nodes HIP events cannot time
            d.update(lower_visits=st.tile_visits - st.raise_tile_visits, lower_launches=st.launches - st.raise_launches,
                     lower_kernel_ms=st.kernel_ms - st.raise_kernel_ms, lower_timed=timed)
        return d

    def adopt_stream(ptr):
        # The engine's HIP stream becomes torch's current stream: the collective's completion is then a
        # dependency of the engine's own stream (RCCL chains its internal stream to the current one), with no
        # further event hop and no host block between the broadcast and the patch kernel.  (Measured with one
        # rank on RCCL: 38 us per replan with a separate wait_stream() hop, 21 us this way.)
        try:
            torch.cuda.set_stream(torch.cuda.ExternalStream(ptr, device=dev))
            return None
        except Exception:   # no external-stream support in this torch build: block the host instead
            return lambda: torch.cuda.current_stream().synchronize()

    if M == 0:
        # ---- one map instance per rank (configs 2, 3, 5); rank 0's instance is the BASELINE seed ----
        cost = ufm_amd.synth.cost_map(seed + 1000 * rank, size, size)
        script = list(ufm_amd.synth.replan_script(seed, size, size, n_patches=args.patches))
        psz = script[0][4].shape[0] if script else 31
        d_cost = on_dev(cost)
        if rank == 0 and script:
            d_patches = on_dev(np.stack([s[4] for s in script]))
        else:
            d_patches = torch.empty((max(1, len(script)), psz, psz), dtype=torch.uint8, device=dev)
        d_recv = [torch.empty((psz, psz), dtype=torch.uint8, device=dev) for _ in range(2)]
        if rehearsal:
            planner = factory("single", args.algo, lvl, bool(args.heuristic), 1)
        else:
            planner = ufm_amd.Planner(algo, lvl, bool(args.heuristic), device=dev_index)
        if args.heuristic:
            planner.set_heuristic_multiplier(float(cost.min()))
        planner.set_occupancy_threshold(1)
        patch_ready = None
        if not rehearsal:
            for kv in args.param:
                planner.set_param(kv.split("=")[0], float(kv.split("=")[1]))
            planner.set_profiling(not args.no_profile)
            # HIP events on every 16th launch of a plan (~110 timed launches per run): the event packets cost ~4 us
            # each, with every 4th launch timed the episode was 3 % slower than untimed
            planner.set_param("profile_stride", 16)
            patch_ready = adopt_stream(planner.stream_ptr())
        stream = ep.PatchStream(d_patches if rank == 0 else None, d_recv, dist=dist, rank=rank, sync=patch_ready,
                                pipeline=not args.no_pipeline, count=len(script))
        meta = [(k, s, top, left) for (k, s, top, left, _) in script]
        ptr_cache = {}

        def apply_patch(p, buf, top, left):
            if rehearsal:
                p.patch_map(buf.numpy(), top, left)
                return
            key = id(buf)
            ptr = ptr_cache.get(key)
            if ptr is None:
                ptr = ptr_cache[key] = buf.data_ptr()      # the per-patch views live as long as the stream
            p.patch_map_device(ptr, top, left, psz, psz)

        def set_map(p):
            if rehearsal:
                p.set_map(cost)
            else:
                p.set_map_device(d_cost.data_ptr(), size, size)

        # N = 1 (no process group): the patches are handed over FROM HOST MEMORY through ufm_patch_map inside the timed region -- SURVEY 8(d)'s
        # metric includes the patch upload -- and that episode is `value`; the device-resident form below (the receive buffer of a broadcast:
        # the N > 1 data path) is timed as a second leg and reported as `value_device_inputs`.
        h_patches = np.ascontiguousarray(np.stack([s[4] for s in script])) if script else np.zeros((1, psz, psz), np.uint8)
        h_ptrs = [h_patches.ctypes.data + i * psz * psz for i in range(len(script))]

        class HostPatches:
            broadcasts = 0

            def fetch(self, i):
                return i

            def close(self):
                pass

        def apply_patch_host(p, i, top, left):
            p.patch_map_host(h_ptrs[i], top, left, psz, psz)

        legs = {"device": (stream, apply_patch)}
        if dist is None and not rehearsal and args.patch_inputs != "device":
            legs["host"] = (HostPatches(), apply_patch_host)
        leg = ["host" if "host" in legs else "device"]

        def run_one():
            st_, ap_ = legs[leg[0]]
            return ep.run_episode(planner, set_map=set_map, start=start, goal=goal, script=meta, stream=st_,
                                  apply_patch=ap_, read_stats=step_stats, phases=phases)

        def self_check():
            # every rank's raster must now be its own map with all the broadcast patches applied -- a collective
            # that delivered late or wrong data fails here
            expect = cost.copy()
            for (_k, _s, top, left, patch) in script:
                expect[top:top + patch.shape[0], left:left + patch.shape[1]] = patch
            return [(0, planner.read_map(size, size), expect)]
        n_rounds = len(script)
    else:
        # ---- M independent maps per rank, every map with its own patch stream (config 4) ----
        first = rank * M
        n_global = world * M
        costs = [ufm_amd.synth.cost_map(seed + first + m, size, size) for m in range(M)]
        n_rounds = args.patches
        d_costs = [on_dev(c) for c in costs]
        starts, rounds, psz = [], None, 31
        scripts = {}
        if n_rounds:
            gen0 = list(ufm_amd.synth.replan_script(seed, size, size, n_patches=n_rounds))
            starts = [s for (_k, s, _t, _l, _p) in gen0]       # every map's robot advances alike
            psz = gen0[0][4].shape[0]
        if rank == 0 and n_rounds:
            gens = [list(ufm_amd.synth.replan_script(seed + g, size, size, n_patches=n_rounds)) for g in range(n_global)]
            rounds = on_dev(np.stack([ep.pack_round([(g, gens[g][i][2], gens[g][i][3], gens[g][i][4]) for g in range(n_global)], psz)
                                      for i in range(n_rounds)]))
        for m in range(M):      # for the self-check only
            scripts[m] = list(ufm_amd.synth.replan_script(seed + first + m, size, size, n_patches=n_rounds))
        rec = (ep.REC_HDR + psz * psz + 15) // 16 * 16
        d_recv = [torch.empty((n_global, rec), dtype=torch.uint8, device=dev) for _ in range(2)]
        if rehearsal:
            planner = factory("batch", args.algo, lvl, bool(args.heuristic), M)
        else:
            planner = ufm_amd.BatchPlanner(M, algo, lvl, bool(args.heuristic), device=dev_index)
        if args.heuristic:
            planner.set_heuristic_multiplier(float(min(c.min() for c in costs)))
        planner.set_occupancy_threshold(1)
        if not rehearsal:
            # (opt-in: a round's patches sit in the round's receive buffer, which is only written again two rounds later -- so they may be applied
            #  by one launch at the step that consumes them instead of one launch per map at the call: include/ufm.h)
            planner.set_param("defer_patches", 1)
            for kv in args.param:
                planner.set_param(kv.split("=")[0], float(kv.split("=")[1]))
            planner.set_profiling(not args.no_profile)
            planner.set_param("profile_stride", 16)
            adopt_stream(planner.stream_ptr(0))
        stream = ep.RoundStream(rounds, d_recv, n_rounds, dist=dist, rank=rank, pipeline=not args.no_pipeline)
        # The record headers (map id, top, left, edge: the planner surface takes positions as plain ints) are a recording like the
        # patch bytes: rank 0 hands every rank a host copy of all rounds' headers ONCE, before the episodes -- reading them back
        # from each received buffer was a device-to-host copy and a host synchronisation per replan round on every rank.  The bytes
        # of a round still arrive by that round's broadcast; the engine reads them in stream order behind it.
        if n_rounds:
            if rank == 0:
                hdr_all = torch.from_numpy(np.ascontiguousarray(rounds.cpu().numpy()[:, :, :ep.REC_HDR]).view(np.int32).reshape(n_rounds, n_global, 4).copy())
            else:
                hdr_all = torch.empty((n_rounds, n_global, 4), dtype=torch.int32)
            if dist is not None:
                hbuf = hdr_all if rehearsal or args.backend == "gloo" else hdr_all.to(dev)
                dist.broadcast(hbuf, src=0)
                hdr_all = hbuf.cpu()
            hdr_all = hdr_all.numpy()

        def headers_of(i, buf):
            return hdr_all[i]

        def apply_record(b, m, buf, g, top, left, edge):
            if rehearsal:
                b.patch_map(m, buf[g, ep.REC_HDR:ep.REC_HDR + edge * edge].numpy().reshape(edge, edge), top, left)
            else:
                b.patch_map_device(m, buf.data_ptr() + g * buf.stride(0) + ep.REC_HDR, top, left, edge, edge)

        def set_maps(b):
            for m in range(M):
                if rehearsal:
                    b.set_map(m, costs[m])
                else:
                    b.set_map_device(m, d_costs[m].data_ptr(), size, size)

        def run_one():
            return ep.run_batch_episode(planner, M, first, set_maps, start, goal, starts, stream, headers_of, apply_record, step_stats, phases=phases)

        def self_check():
            out = []
            for m in (0, M - 1):
                expect = costs[m].copy()
                for (_k, _s, top, left, patch) in scripts[m]:
                    expect[top:top + patch.shape[0], left:left + patch.shape[1]] = patch
                out.append((m, planner.read_map(m, size, size), expect))
            return out

    phases = {}         # host wall seconds of the episodes' parts, summed over warm-up and timed episodes (run_episode)
    if M != 0:
        legs, leg = {"device": None}, ["device"]

    def barrier():
        if not rehearsal:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        if not rehearsal:
            torch.cuda.synchronize()

    def run_warm():     # (only the timed episodes count for the phase split)
        r = run_one()
        phases.clear()
        return r
    for _ in range(args.warmup):
        run_warm()
    dt, per_step = ep.timed_episodes(run_one, args.steps, 0, barrier)
    device_leg = None
    if leg[0] == "host" and args.patch_inputs == "both":            # the second leg: the same episodes with the patches resident in HBM (ufm_patch_map_device)
        phases_host = dict(phases)
        leg[0] = "device"
        run_warm()
        dt_d, per_step_d = ep.timed_episodes(run_one, args.steps, 0, barrier)
        cells_d = sum(stats_dict(sn)["cells"] for snaps in per_step_d for sn in snaps)
        device_leg = {"dt": dt_d, "cells": cells_d, "phases": dict(phases)}
        phases.clear(); phases.update(phases_host)
        leg[0] = "host"
    last_snapshot = None
    if not rehearsal:       # the snapshots of every step of an episode -> one summed dict per episode
        summed = []
        if per_step and per_step[-1]:
            last_snapshot = per_step[-1][-1]
        for snaps in per_step:
            tot_ = {}
            for sn in snaps:
                for k_, v_ in stats_dict(sn).items():
                    tot_[k_] = tot_.get(k_, 0) + v_
            summed.append(tot_)
        per_step = summed
    keys = ("cells", "visits", "launches", "kernel_ms", "evals")
    tot = [sum(d.get(k, 0) for d in per_step) for k in keys]
    low = [sum(d.get(k, 0) for d in per_step) for k in ("lower_visits", "lower_launches", "lower_kernel_ms", "lower_timed")]
    res = [sum(d.get(k, 0) for d in per_step) for k in ("res_visits", "res_launches", "res_kernel_ms")]
    reg = [sum(d.get(k, 0) for d in per_step) for k in ("reg_launches", "reg_timed", "reg_kernel_ms", "reg_tiles", "reg_cells")]

    # self-check of the patch path (outside the timed region)
    for m, got, expect in self_check():
        if not np.array_equal(got, expect):
            bad = np.argwhere(got != expect)
            raise RuntimeError("rank %d map %d: raster on the device differs from map + patches in %d cells, first at %r (device %d, expected %d)" % (
                rank, m, len(bad), tuple(bad[0]), got[tuple(bad[0])], expect[tuple(bad[0])]))

    cells, visits, launches, kms, evals = tot
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        c = torch.tensor([cells, visits, launches, evals, stream.broadcasts], dtype=torch.float64, device=dev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        cells_all = float(c[0].item())
        bcast_all = int(c[4].item())
    else:
        cells_all = float(cells)
        bcast_all = 0

    if rank == 0:
        episodes = args.steps + args.warmup
        maps_desc = "one map instance per GPU" if M == 0 else "%d independent maps per GPU (seeds %d + global map id), every map with a patch stream of its own" % (M, seed)
        out = {
            "metric": "grid cells updated/sec (full plan + %d replans), %d^2 map" % (n_rounds, size),
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
                "workload": "%s, %dx%d cost map (seed %d generator, SURVEY 8d), full plan + %d 31x31 patches with moving start, %s; %s" % (
                    ALGO_LABEL[args.algo], size, size, seed, n_rounds, maps_desc,
                    "patches handed over from host memory inside the timed region (ufm_patch_map)" if leg[0] == "host" else
                    "patches resident in HBM (ufm_patch_map_device: the receive buffer of the patch broadcast)"),
                "patch_inputs": leg[0],
                "algo": args.algo, "size": size, "patches": n_rounds, "heuristic_keys": bool(args.heuristic),
                "maps_per_gpu": n_local, "maps_total": n_local * world,
                # one broadcast per replan (round) per rank, warm-up episodes included; 0 without a process group
                "broadcasts_per_rank_per_episode": (bcast_all / world / episodes) if (dist is not None and episodes) else 0,
                "cells_per_step_rank0": cells / max(1, args.steps),
                "relax_launches_per_step_rank0": launches / max(1, args.steps),
                "tile_visits_per_step_rank0": visits / max(1, args.steps),
                "elem_evals_per_step_rank0": evals / max(1, args.steps),
            },
        }
        # where the episode's time goes (rank 0's host clock around the calls, per timed episode): set_map (untimed by the metric's
        # definition in the reference, inside the timed region here), the plan step, the replans (patch + set_start + step each)
        steps_ = max(1, args.steps)
        out["phases"] = {"set_map_ms": 1e3 * phases.get("set_map", 0.0) / steps_, "plan_ms": 1e3 * phases.get("plan", 0.0) / steps_,
                         "replans_ms": 1e3 * phases.get("replans", 0.0) / steps_, "replans": n_rounds,
                         "plan_cells": (cells - reg[4]) / steps_ if not rehearsal else None,
                         "replans_cells": reg[4] / steps_ if not rehearsal else None,
                         "note": "host wall per episode on rank 0; cells of the replans = those of the steps the block kernel ran (all of them here)"}
        if device_leg is not None:
            out["value_device_inputs"] = device_leg["cells"] / device_leg["dt"]
            out["ms_per_step_device_inputs"] = 1e3 * device_leg["dt"] / steps_
            out["phases"]["replans_ms_device_inputs"] = 1e3 * device_leg["phases"].get("replans", 0.0) / steps_
        if rehearsal:
            out["rehearsal"] = "CPU stand-in planner %s over %s: control flow only, not a measurement" % (factory_label, args.backend)
        lvis, llaunch, lkms, ltimed = low
        rvis, rlaunch, rkms = res
        traffic_ok = args.algo == "FD" and size == 4096 and M == 0 and not args.heuristic
        if not rehearsal and rlaunch > 0 and rkms > 0:
            # dominant kernel: the resident lowering kernel, k_relax<algo, LOWER, false, 1 | 2> -- ONE launch per plan runs the whole
            # lowering phase.  Algorithmic bytes = its tile visits x (9 B per element + halo) per SURVEY.md 8(d); duration = HIP events
            # attached to that dispatch on the engine's stream (every launch of it inside the timed region is timed).
            avg_launch_s = rkms * 1e-3 / rlaunch
            achieved = (rvis / rlaunch) * BYTES_PER_TILE_VISIT / avg_launch_s / 1e9
            traffic, traffic_source = None, None
            tj = os.path.join(ROOT, "profiles", args.traffic_json)
            if os.path.exists(tj) and traffic_ok:
                tjd = json.load(open(tj))
                if "resident" in tjd.get("kernel", ""):
                    traffic = tjd.get("traffic_bytes_per_launch")
                    traffic_source = "profiles/%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, collected separately (not in this run)" % args.traffic_json
            out["roofline"] = {
                "bound": "hbm", "kernel": "k_relax<%s,LOWER,resident> (one launch per plan: the whole lowering phase)" % args.algo, "achieved": achieved,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                "avg_launch_us": 1e6 * avg_launch_s, "timed_launches": rlaunch, "launches": rlaunch,
                "algorithmic_bytes_per_launch": rvis * BYTES_PER_TILE_VISIT / rlaunch,
                "tile_visits_per_launch": rvis / rlaunch,
                "kernel_time_share": avg_launch_s * rlaunch / dt,
                "note": "latency-bound (dependent in-LDS sweeps along the wavefront), not bandwidth-bound: see DESIGN.md",
            }
        elif not rehearsal and lkms > 0 and llaunch > 0 and ltimed > 0:
            # dominant kernel: k_relax<algo, LOWER>.  Algorithmic bytes = tile visits x (9 B per element
            # + halo) per SURVEY.md 8(d); duration = HIP events on the engine's stream around its launches.
            # (the events bracket a sample of the launches -- every 16th of a plan --
            # so that they can stay on inside the timed region: ltimed of the llaunch launches)
            avg_launch_s = lkms * 1e-3 / ltimed
            achieved = (lvis / llaunch) * BYTES_PER_TILE_VISIT / avg_launch_s / 1e9
            traffic, traffic_source = None, None
            tj = os.path.join(ROOT, "profiles", args.traffic_json)
            if os.path.exists(tj) and traffic_ok and "resident" not in json.load(open(tj)).get("kernel", ""):
                traffic = json.load(open(tj)).get("traffic_bytes_per_launch")
                traffic_source = "profiles/%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, collected separately (not in this run)" % args.traffic_json
            out["roofline"] = {
                "bound": "hbm", "kernel": "k_relax<%s,LOWER,cursor> (the lowering kernel as launched by the plans)" % args.algo, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                "avg_launch_us": 1e6 * avg_launch_s, "timed_launches": ltimed, "launches": llaunch,
                "algorithmic_bytes_per_launch": lvis * BYTES_PER_TILE_VISIT / llaunch,
                "tile_visits_per_launch": lvis / llaunch,
                "kernel_time_share": avg_launch_s * llaunch / dt,
                "note": "latency-bound (dependent in-LDS sweeps along the wavefront), not bandwidth-bound: see DESIGN.md",
            }
        rl_, rt_, rk_, rtiles_, _rc = reg
        if not rehearsal and rl_ > 0 and rt_ > 0 and rk_ > 0:
            # the replans' kernel: k_replan_region, ONE workgroup per map runs a replan's invalidation and lowering on a block of tiles in
            # LDS.  Units per SURVEY 8(d): the tiles it stages (block edge^2) x (9 B per element + halo); duration = HIP events attached
            # to every 8th of its dispatches inside the timed region.
            last = ufm_amd.capi.Stats.from_buffer_copy(last_snapshot) if last_snapshot is not None else None
            avg_s = rk_ * 1e-3 / rt_
            tiles_per = rtiles_ / rl_
            ach = tiles_per * BYTES_PER_TILE_VISIT / avg_s / 1e9
            rtraffic = None
            tj = os.path.join(ROOT, "profiles", args.traffic_json)
            if os.path.exists(tj) and traffic_ok:
                rtraffic = json.load(open(tj)).get("region_traffic_bytes_per_launch")
            out["roofline_replans"] = {
                "bound": "hbm", "kernel": "k_replan_region<%s> (one launch per replan: invalidation + lowering of a block of tiles in one workgroup's LDS)" % args.algo,
                "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": rtraffic,
                "avg_launch_us": 1e6 * avg_s, "timed_launches": rt_, "launches": rl_, "tile_visits_per_launch": tiles_per,
                "algorithmic_bytes_per_launch": tiles_per * BYTES_PER_TILE_VISIT,
                "region_replans": last.region_replans if last else None, "region_replans_done": last.region_replans_done if last else None,
                "kernel_time_share": avg_s * rl_ / dt,
                "note": "one CU of 256: a replan changes ~1 k elements; latency-bound chain of dependent patch sweeps (DESIGN.md 4.6)",
            }
        if not rehearsal and not args.no_cpu_baseline and world == 1:   # the CPU baseline is reported at N = 1 only
            out["cpu_baseline"] = cpu_baseline(args.cpu_size, seed, n_rounds, args.algo, bool(args.heuristic), n_local)
        print(json.dumps(out))
    stream.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main(argv=None, planner_factory=None, factory_label=None, script=None):
    """argv: the command line (default sys.argv[1:]); planner_factory / factory_label / script: the CPU rehearsal of
    tests/bench_rehearsal.py (a stand-in planner, and the file the rank processes are started from)."""
    argv = list(sys.argv[1:] if argv is None else argv)
    if len(argv) >= 2 and argv[0] == "--cpu-worker":
        cpu_worker(json.loads(argv[1]))
        return 0
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--patches", type=int, default=100)
    ap.add_argument("--seed", type=int, default=None, help="seed of the synthetic map(s) and patch script(s) (SURVEY 8d: 7 for the headline, 1234 "
                    "config 2, 42 config 5; with --batch: 1000 + global map id)")
    ap.add_argument("--algo", default="FD", choices=["FD", "SG", "DFM"])
    ap.add_argument("--batch", type=int, default=0, help="M > 0: M independent maps per rank in one batch handle, every map with its own patch "
                    "stream (BASELINE config 4: --gpus 8 --algo DFM --size 2048 --batch 8)")
    ap.add_argument("--heuristic", action="store_true",
                    help="heuristic keys with hm = the map's smallest cost (planners built without -DNO_HEURISTIC; BASELINE config 5 with --size 8192)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-size", type=int, default=None, help="map size of the CPU baseline sample (default: the identical workload)")
    ap.add_argument("--no-profile", action="store_true", help="skip per-launch HIP event timing")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1 (nccl = RCCL; gloo only to rehearse the control flow, staging through host memory)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="broadcast each patch right before its replan instead of one replan ahead")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and run the collectives even with one rank (rehearses the RCCL path on a one-GPU box)")
    ap.add_argument("--timeout", type=float, default=1500.0, help="--gpus N self-launch: seconds after which the rank processes are stopped (exit code 124)")
    ap.add_argument("--param", action="append", default=[], metavar="NAME=VALUE",
                    help="scheduler knob for the engine (ufm_set_param / ufm_batch_set_param), e.g. owned_waves=16; for experiments -- the defaults are the product")
    ap.add_argument("--patch-inputs", default="both", choices=["both", "host", "device"],
                    help="N = 1, one map per GPU: where the patches come from.  both (default): the episodes with patches from host memory are `value`, a second leg "
                    "with the patches resident in HBM is `value_device_inputs`; host / device: that leg only (profiles).  With a process group the patches arrive in HBM")
    ap.add_argument("--traffic-json", default="r4_traffic.json", help="file under profiles/ holding the separately collected PMC traffic of the headline run")
    args = ap.parse_args(argv)
    if args.seed is None:
        args.seed = 1000 if args.batch > 0 else 7
    if args.cpu_size is None:
        args.cpu_size = args.size
    if args.gpus < 1 or args.batch < 0:
        raise SystemExit("--gpus >= 1, --batch >= 0")
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and env_world is None:
        return launch_ranks(args.gpus, script or __file__, argv, args.timeout)      # nothing above has imported torch or touched a GPU
    if env_world is not None and int(env_world) != args.gpus and int(env_world) > 1:
        print("bench.py: --gpus %d but WORLD_SIZE=%s; the process group decides" % (args.gpus, env_world), file=sys.stderr)
    run_rank(args, planner_factory, factory_label)
    return 0


if __name__ == "__main__":
    sys.exit(main())
