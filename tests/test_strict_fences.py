"""-m gpu: the resident plan kernel's in-launch memory protocol (sc1 stores and loads, s_waitcnt vmcnt(0), relaxed
agent-scope atomics -- argued, not release / acquire) against a checking build of the same source with the textbook fences
(-DUFM_STRICT_FENCES: agent-scope release in front of every activation and lock release, acquire behind every take;
unige-tasi-path-planners_amd/libufm_strict.so).  Same inputs, same scheduler forms (both wave counts, early hand-off and in-visit
refresh on / off, helping on / off, a narrow band), whole fields compared bit for bit.  A one-off cross-check of the argument.
The checking build also keeps only two tiles in the LDS copy of the block replan kernel's write-back (-DUFM_REGION_NG0=2: the product keeps 46), so the
four replans of every case run the path of the tiles beyond the copy against the product's."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "unige-tasi-path-planners_amd")


def _probe(lib):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "strict_probe.py"), os.path.join(PKG, lib)],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = out.stdout.strip().splitlines()
    return lines[0], [l for l in lines if l.startswith("case")]


def test_product_build_equals_the_strict_fence_build_bit_for_bit():
    assert os.path.exists(os.path.join(PKG, "libufm_strict.so")), "make -C unige-tasi-path-planners_amd (or __graft_entry__.build()) builds the checking library"
    v0, prod = _probe("libufm.so")
    v1, strict = _probe("libufm_strict.so")
    assert "strict-fence" in v1 and "strict-fence" not in v0
    assert len(prod) == 5 and prod == strict, "\n".join(["product:"] + prod + ["strict:"] + strict)
    assert len({l.split()[-2] for l in prod}) == 1 and len({l.split()[-1] for l in prod}) == 1      # ... and the scheduler forms agree with each other
