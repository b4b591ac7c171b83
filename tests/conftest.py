import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ref_bitmaps():
    import numpy as np
    d = np.load(os.path.join(ROOT, "tests", "golden", "ref_bitmaps.npz"))
    out = {}
    for k in d.files:
        if k.endswith("_cost"):
            name = k[:-5]
            out[name] = (d[k], tuple(float(v) for v in d[name + "_startgoal"]))
    return out


def pytest_terminal_summary(terminalreporter):
    """MS-DFM: the worst deviation from the oracle each test saw, next to the bound it was held to (helpers.note_margin) -- the bound is
    self-derived (ufm_amd.tolerances), so the margin under it is printed with every run"""
    try:
        import helpers
    except Exception:
        return
    if not helpers.MARGINS:
        return
    terminalreporter.write_sep("-", "MS-DFM deviations from the oracle (worst per test; bound = what the test asserts)")
    for test, m in sorted(helpers.MARGINS.items()):
        terminalreporter.write_line("%-70s worst rel %.3g (%d ulp), bound %.1e, %d of %d values differ, %d comparisons" % (
            test[:70], m["rel"], m["ulp"], m["bound"], m["differing"], m["n"], m["cmp"]))
