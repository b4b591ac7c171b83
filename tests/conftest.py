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
