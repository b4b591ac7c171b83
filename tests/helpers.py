"""Shared helpers for the parity tests (HIP engine vs CPU oracle)."""
import numpy as np

import oracle_py as orc
import ufm_amd

ALGOS = {"FD": 0, "SG": 1, "DFM": 2}


def make_pair(algo, opt_lvl, cost, start, goal, thr=1.0, heuristic=False, hm=1.0):
    """Configure an oracle planner and a HIP planner the way the reference's
    demo drivers do (Tests/Planners/FDSTAR/main.cpp:82-88)."""
    o = orc.OraclePlanner(algo, opt_lvl, heuristic)
    g = ufm_amd.Planner(algo, opt_lvl, heuristic)
    for p in (o, g):
        p.reset()
        p.set_occupancy_threshold(thr)
        p.set_heuristic_multiplier(hm)
        p.set_map(cost)
        p.set_start(*start)
        p.set_goal(*goal)
    return o, g


def ulp_diff(a, b):
    """distance in units in the last place between float32 arrays (finite, same sign)"""
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    return np.abs(ia - ib)


def check_parity(o, g, what="", below_start_key=False):
    """Compare the HIP field with the oracle on the set of elements whose value
    the reference guarantees final (consistent and not beyond the queue top).
    Target: bit-equal (FD / SG are asserted bit-equal by the callers).  Acceptance bound
    (SURVEY.md 8d): |dG| <= max(1e-6*G, 2 ulp); for DFM 2e-6*G: the float fixed point of its upwind
    quadratic is not unique and the engine cuts ulp-level creep short --
    measured worst case 13 ulp = 1.2e-6 on 2048^2."""
    og, orhs = o.g(), o.rhs()
    mask = o.trusted_mask(below_start_key=below_start_key)
    gg, grhs = g.read_field()
    assert gg.shape == og.shape, (gg.shape, og.shape)
    n = int(mask.sum())
    assert n > 0, "oracle produced an empty consistent set " + what
    a, b = gg[mask], og[mask]
    finite = np.isfinite(a)
    assert finite.all(), "%s: %d trusted elements are +inf on the device" % (what, int((~finite).sum()))
    nbad = int((a != b).sum())
    if nbad:
        ud = ulp_diff(a, b)
        rel = np.abs(a.astype(np.float64) - b) / np.maximum(b, 1e-30)
        rtol = 2e-6 if o.algo == orc.ALGO_DFM else 1e-6
        assert (ud <= 2).all() or (rel <= rtol).all(), "%s: max ulp %d, max rel %.3g over %d/%d differing" % (
            what, int(ud.max()), float(rel.max()), nbad, n)
    # RHS view: equals G at the fixed point; must agree with the oracle's RHS wherever that is final
    assert np.array_equal(grhs[mask], gg[mask])
    # the engine's redundant copies (neighbour rings, cost windows) agree with their originals
    assert g.check_layout() == (0, 0), "%s: layout self-check %r" % (what, g.check_layout())
    return n, nbad
