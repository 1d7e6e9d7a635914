"""CPU-side checks of the drop-in boundary: the shared library loads and exports
every symbol include/spdp.h declares; the ctypes mirrors match the C layouts."""
import ctypes as C
import os
import re
import subprocess

import pytest

from spaln_amd import abi, engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "spdp.h")).read()
    return sorted(set(re.findall(r"\b(spdp_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_all_declared_symbols():
    lib = C.CDLL(engine.LIB_PATH)
    missing = [s for s in _declared() if not hasattr(lib, s)]
    assert not missing, missing
    assert set(engine.EXPORTS) <= set(_declared())


def test_struct_layout_matches_header(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "spdp.h"\n'
                   'int main(){printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(SpdpScoring), sizeof(SpdpProblem),'
                   'offsetof(SpdpScoring, gop), offsetof(SpdpScoring, qm_len), offsetof(SpdpProblem, a_left),'
                   'sizeof(SpdpAlignment));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(abi.Scoring), C.sizeof(abi.Problem), abi.Scoring.gop.offset,
            abi.Scoring.qm_len.offset, abi.Problem.a_left.offset, C.sizeof(abi.Alignment)]
    assert got == want


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        engine.Engine(0)
