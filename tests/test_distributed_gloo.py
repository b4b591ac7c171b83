"""CPU, world_size 2 over gloo: the multi-rank form of the benchmark episode -- independent map
instances per rank, the patch stream owned by rank 0 and broadcast before every replan, sum /
max reductions of the results -- with the CPU oracle standing in for the planner."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import ufm_amd

SIZE, SEED, NP = 96, 5, 6


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _episode(rank, world, use_dist, pipeline=False):
    import oracle_py as orc
    cost = ufm_amd.synth.cost_map(SEED + 1000 * rank, SIZE, SIZE)
    start, goal = ufm_amd.synth.start_goal(SIZE, SIZE)
    script = list(ufm_amd.synth.replan_script(SEED, SIZE, SIZE, n_patches=NP))
    patches = [torch.from_numpy(s[4].copy()) for s in script] if rank == 0 else None
    recv = [torch.empty((31, 31), dtype=torch.uint8) for _ in range(2)]
    ep = ufm_amd.episode
    stream = ep.PatchStream(patches, recv, dist=dist if use_dist else None, rank=rank, pipeline=pipeline, count=NP)
    planner = orc.OraclePlanner(orc.ALGO_FD, 1, False)
    planner.set_occupancy_threshold(1)
    seen = []

    def apply_patch(p, buf, top, left):
        seen.append(buf.numpy().copy())
        p.patch_map(buf.numpy(), top, left)

    tot = ep.run_episode(planner, lambda p: p.set_map(cost), start, goal,
                         [(k, s, t, l) for (k, s, t, l, _) in script], stream, apply_patch,
                         lambda p: {"cells": p.num_expanded})
    stream.close()
    return tot["cells"], seen, stream.broadcasts


def _worker(rank, world, port, q, pipeline):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cells, seen, nb = _episode(rank, world, True, pipeline)
    cells2, seen2, nb2 = _episode(rank, world, True, pipeline)      # a second episode on the same process group
    assert cells2 == cells and nb2 == nb
    ref = [s[4] for s in ufm_amd.synth.replan_script(SEED, SIZE, SIZE, n_patches=NP)]
    ok = all(np.array_equal(a, b) for a, b in zip(seen, ref)) and nb == NP
    t = torch.tensor([float(cells), 1.0 if ok else 0.0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    if rank == 0:
        q.put((t[0].item(), t[1].item()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("pipeline", [False, True])
def test_two_ranks_broadcast_patches_and_aggregate(pipeline):
    """pipeline: the broadcast of patch i+1 is issued by a helper thread while replan i runs"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, pipeline)) for r in range(world)]
    for p in procs:
        p.start()
    total, oks = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert oks == world                       # every rank received rank 0's patches bit-exactly
    # rank r must have done exactly what a stand-alone run with the same map and the same
    # (rank 0) patch stream does
    import oracle_py as orc
    expect = 0
    for r in range(world):
        cost = ufm_amd.synth.cost_map(SEED + 1000 * r, SIZE, SIZE)
        start, goal = ufm_amd.synth.start_goal(SIZE, SIZE)
        p = orc.OraclePlanner(orc.ALGO_FD, 1, False)
        p.set_occupancy_threshold(1); p.set_map(cost); p.reset(); p.set_start(*start); p.set_goal(*goal)
        assert p.step() == 0
        expect += p.num_expanded
        for k, s, top, left, patch in ufm_amd.synth.replan_script(SEED, SIZE, SIZE, n_patches=NP):
            p.patch_map(patch, top, left); p.set_start(*s)
            assert p.step() == 0
            expect += p.num_expanded
    assert total == expect
