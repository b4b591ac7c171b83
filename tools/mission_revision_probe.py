"""CPU only.  Which revision of the reference wrote its two recorded mission logs (Tests/Results/{noise-trap,wall-b}/planner_opt0.log)?
The logs print lines the current sources have commented out (FieldDPlanner_impl.h:65,139), and with the sources as they stand the
restatement reproduces noise-trap's paths but not all of its counts, and parts from wall-b in its second step.  The pattern of the
differences named the candidates -- "nodes expanded" is off exactly where a start coordinate's fraction is >= 0.5 (roundf against floor),
"nodes updated" exactly where a changed cell touches the map's bottom row / right column -- and this probe replays both logs closed-loop
under each combination (oracle: orc_set_revision), printing per log: steps whose position / path cost / path length agree to the printed
digit, first step that does not, steps whose "nodes updated" / "nodes expanded" agree.  Then the ablations of FD's compute_optimal_cost and
the case counters (what the logs pin of the operator), and the shifted-grid planner in FD's place.
usage: python tools/mission_revision_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import oracle_py as orc
import ufm_amd
import test_reference_mission as trm


def run(name, planner):
    ok = upd = exp = nu = ne = 0
    first = None
    k = -1
    for k, st, got in trm.replay(name, planner, trm.o_counts):
        same = (got["pos"], got["cost"], got["dist"]) == (st["pos"], st["cost"], st["dist"])
        ok += same
        if not same and first is None:
            first = k
        if "updated" in st:
            nu += 1; upd += got["updated"] == st["updated"]
        if "expanded" in st:
            ne += 1; exp += got["expanded"] == st["expanded"]
    return "%3d steps, paths agree in %3d (first that does not: %s), nodes updated %3d/%3d, nodes expanded %3d/%3d" % (k + 1, ok, first, upd, nu, exp, ne)


print("== revisions (Field D* level 0, heuristic keys)")
for rev, what in ((orc.REV_CURRENT, "current sources (start cell = roundf, update() takes all four corners)"),
                  (orc.REV_START_CELL_FLOOR, "start cell = floor"), (orc.REV_UPDATE_SKIPS_FAR_BORDER, "update() without the far-border corner nodes"),
                  (orc.REV_LOG, "both")):
    for name in ("noise-trap", "wall-b"):
        print("%-72s %-10s %s" % (what, name, run(name, orc.OraclePlanner(ufm_amd.ALGO_FD, 0, True, revision=rev))))
print("== other planners in the logs' revision")
for algo, lvl, what in ((ufm_amd.ALGO_FD, 1, "Field D* level 1"), (ufm_amd.ALGO_SG, 0, "shifted grid level 0"), (ufm_amd.ALGO_SG, 2, "shifted grid level 2")):
    for name in ("noise-trap", "wall-b"):
        print("%-72s %-10s %s" % (what, name, run(name, orc.OraclePlanner(algo, lvl, True, revision=orc.REV_LOG))))
print("== ablations of FD's compute_optimal_cost (FD impl:292-319), logs' revision")
L = orc.lib()
for mask, what in ((1, "without the f^2 <= CATH(c,b) clause of Type III"), (8, "Type III pays c instead of b"), (2, "without Type I"), (4, "without the c > b chain")):
    L.orc_set_fd_ablation(mask)
    for name in ("noise-trap", "wall-b"):
        print("%-72s %-10s %s" % (what, name, run(name, orc.OraclePlanner(ufm_amd.ALGO_FD, 0, True, revision=orc.REV_LOG))))
L.orc_set_fd_ablation(0)
print("== which cases the missions take (evaluated / gave the minimum of a min_rhs call)")
for name in ("noise-trap", "wall-b"):
    orc.case_counts_reset()
    run(name, orc.OraclePlanner(ufm_amd.ALGO_FD, 0, True, revision=orc.REV_LOG))
    ev, won = orc.case_counts()
    for c in orc.CASES[:8]:
        print("   %-10s %-34s %9d %9d" % (name, c, ev[c], won[c]))
