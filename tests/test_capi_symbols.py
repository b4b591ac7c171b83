"""CPU: libufm.so loads and exports every entry point include/ufm.h declares.  No compute."""
import ctypes
import os
import re

import pytest

import ufm_amd
from ufm_amd_pkg import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "ufm.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ufm_[a-z_0-9]+)\s*\(", txt)))


def test_header_and_binding_list_agree():
    assert _declared() == sorted(capi.SYMBOLS)


def test_header_tolerance_is_the_tested_one():
    """the MS-DFM bound the queue view uses (include/ufm.h) is the one the parity tests hold the planner to"""
    txt = open(os.path.join(ROOT, "include", "ufm.h")).read()
    m = re.search(r"#define\s+UFM_DFM_RTOL\s+([0-9.eE+-]+)f", txt)
    assert m and float(m.group(1)) == ufm_amd.tolerances.DFM_RTOL


def test_library_exports_every_symbol():
    if not os.path.exists(ufm_amd.library_path()):
        ufm_amd.build_library()
    lib = ctypes.CDLL(ufm_amd.library_path())
    for name in _declared():
        assert hasattr(lib, name), name


def test_no_cpu_fallback():
    """Without a GPU creating a planner must fail loudly (negative code -> UfmError), never fall
    back to a CPU path.  With a GPU it simply succeeds."""
    try:
        p = ufm_amd.Planner(ufm_amd.ALGO_FD, 0)
    except ufm_amd.UfmError as e:
        assert "ufm_create" in str(e)
        return
    p.close()


def test_bad_arguments_rejected():
    lib = ufm_amd.load_library()
    h = ctypes.c_void_p()
    assert lib.ufm_create(ctypes.byref(h), 7, 0, 0, 0) == -22          # unknown planner family
    assert lib.ufm_create(ctypes.byref(h), ufm_amd.ALGO_FD, 2, 0, 0) == -22   # only SG has level 2
    assert lib.ufm_version().startswith(b"ufm-gfx950")


def test_header_is_plain_c(tmp_path):
    """include/ufm.h is the drop-in boundary: it must compile as C (no C++ in the signatures) and a C
    program must link against libufm.so"""
    import subprocess
    src = tmp_path / "abi.c"
    src.write_text('#include "ufm.h"\n#include <stdio.h>\n'
                   'int main(void) { ufm_stats s; ufm_path_info pi; (void)s; (void)pi;\n'
                   '  printf("%s tile %d\\n", ufm_version(), ufm_tile_edge()); return 0; }\n')
    exe = tmp_path / "abi"
    pkg = os.path.dirname(ufm_amd.library_path())
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           str(src), "-o", str(exe), "-L" + pkg, "-lufm", "-Wl,-rpath," + pkg])
    out = subprocess.check_output([str(exe)], text=True)
    assert out.startswith("ufm-gfx950")
