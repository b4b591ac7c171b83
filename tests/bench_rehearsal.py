"""CPU rehearsal of bench.py's multi-rank control flow (tests only; the shipped benchmark has no way to import a planner by name):
bench.main() with the oracle behind the Python planner surface (tests/rehearsal_planner.py) instead of the HIP planner, so that
the self-launcher, the process group, the broadcasts and the reductions run end to end without a GPU.  The rank processes of
`--gpus N` are started from THIS file.

  python tests/bench_rehearsal.py --gpus 2 --backend gloo --size 96 --patches 5 [--fail-rank R]

--fail-rank R: the planner factory of rank R raises (the launcher must end the other ranks and report R's code and traceback)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)

import bench  # noqa: E402


def main():
    argv = list(sys.argv[1:])
    fail_rank = None
    if "--fail-rank" in argv:
        i = argv.index("--fail-rank")
        fail_rank = int(argv[i + 1])
        # (the flag stays in argv: the launcher hands the same command line to the ranks; bench's parser never sees it)
    bench_argv = [a for j, a in enumerate(argv) if not (a == "--fail-rank" or (j > 0 and argv[j - 1] == "--fail-rank"))]

    def factory(kind, algo, lvl, heuristic, n_maps):
        import rehearsal_planner
        if fail_rank is not None and int(os.environ.get("RANK", "0")) == fail_rank:
            raise RuntimeError("rehearsal: the planner of rank %d fails on purpose" % fail_rank)
        return rehearsal_planner.make(kind, algo, lvl, heuristic, n_maps)

    if "WORLD_SIZE" not in os.environ:
        # the parent: parse with bench's parser, start the ranks from this file with the full command line
        import argparse
        ap = argparse.ArgumentParser(add_help=False)
        ap.add_argument("--gpus", type=int, default=1)
        ap.add_argument("--timeout", type=float, default=1500.0)
        known, _ = ap.parse_known_args(bench_argv)
        if known.gpus > 1:
            return bench.launch_ranks(known.gpus, __file__, argv, known.timeout)
    return bench.main(bench_argv, planner_factory=factory, factory_label="tests/rehearsal_planner.make", script=__file__)


if __name__ == "__main__":
    sys.exit(main())
