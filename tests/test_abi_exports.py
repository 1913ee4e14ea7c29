"""CPU-side checks of the drop-in boundary: the HIP library loads without a GPU, exports every symbol that
include/vislam_ba.h declares, and fails loudly (no CPU fallback) when no device is present."""
import ctypes as C
import os
import re

import pytest

from mc_slam_amd import backend, abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "vislam_ba.h")).read()
    return sorted(set(re.findall(r"^\s*(?:int|void|const char \*)\s*\*?(vba_\w+)\s*\(", txt, flags=re.M)))


def test_header_symbols_are_exported():
    names = _declared()
    assert set(names) == set(backend.EXPORTS), (names, backend.EXPORTS)
    lib = backend.load_library()
    for n in names:
        assert getattr(lib, n) is not None


def test_shipped_library_exports_exactly_the_header():
    """the test / diagnostic hooks (vba_debug_*) exist only in the hooks flavour of the library"""
    import subprocess
    def exported(path):
        out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
        return sorted(set(l.split()[-1] for l in out.splitlines() if l.split() and l.split()[-1].startswith("vba_")))
    assert exported(backend.LIB_PATH) == _declared()
    hooks = exported(backend.HOOKS_LIB_PATH)
    assert set(_declared()) < set(hooks) and any(n.startswith("vba_debug_") for n in hooks)


def test_struct_layout_matches_header():
    # sizes the C compiler gives the same structs (checked once with gcc -> constants below); guards ctypes drift
    import subprocess, tempfile, textwrap
    src = textwrap.dedent('''
        #include <stdio.h>
        #include "vislam_ba.h"
        int main(){printf("%zu %zu %zu %zu %zu\\n", sizeof(vba_problem), sizeof(vba_result), sizeof(vba_profile), sizeof(vba_frame_problem), sizeof(vba_frame_result));return 0;}''')
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "s.c"); exe = os.path.join(td, "s")
        open(c, "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        sizes = tuple(map(int, subprocess.check_output([exe]).split()))
    assert sizes == (C.sizeof(abi.vba_problem), C.sizeof(abi.vba_result), C.sizeof(abi.vba_profile),
                     C.sizeof(abi.vba_frame_problem), C.sizeof(abi.vba_frame_result))


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="no usable HIP device"):
        backend.LocalBA(0)
