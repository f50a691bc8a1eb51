"""The boundary is a C ABI: include/relmc.h compiles as C11 and as C++17, and a plain-C client (tests/c/abi_smoke.c,
no Python, no C++ types) drives the library."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")
CSRC = os.path.join(ROOT, "powersystemsreliabilityassessment_amd", "csrc")


def _build_client(tmp_path):
    from powersystemsreliabilityassessment_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()                              # hipcc cross-compiles without a GPU
    exe = str(tmp_path / "abi_smoke")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-I", INC, os.path.join(ROOT, "tests", "c", "abi_smoke.c"),
                           "-o", exe, "-L", CSRC, "-lrelmc", "-Wl,-rpath," + CSRC])
    return exe


def test_header_is_c_and_cpp(tmp_path):
    src = tmp_path / "hdr.c"
    src.write_text('#include "relmc.h"\nint main(void) { relmc_acc a; relmc_acc_zero(&a); return (int)a.n; }\n')
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-pedantic", "-fsyntax-only", "-I", INC, str(src)])
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c++", "-I", INC, str(src)])
    _build_client(tmp_path)                       # links against every symbol it uses without a GPU


@pytest.mark.gpu
def test_plain_c_client(tmp_path, case):
    exe = _build_client(tmp_path)
    f = tmp_path / "case.bin"
    with open(f, "wb") as fh:
        fh.write(np.array([case.nb, case.ng, case.nl, case.nd, case.ref_bus], dtype=np.int32).tobytes())
        fh.write(np.array([case.base_mva, case.total_load], dtype=np.float64).tobytes())
        for a, t in ((case.bus_pd, np.float64), (case.inj_bus, np.int32), (case.inj_pmin, np.float64), (case.inj_pmax, np.float64),
                     (case.inj_cost, np.float64), (case.br_from, np.int32), (case.br_to, np.int32), (case.br_b, np.float64),
                     (case.br_rate, np.float64), (case.unavail, np.float64), (case.always_up, np.uint8)):
            fh.write(np.ascontiguousarray(a, dtype=t).tobytes())
    out = subprocess.run([exe, str(f)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    n, n_fail, sum_dns, nd = out.stdout.strip().splitlines()[-1].split()[:4]      # RCCL prints a banner before it
    from powersystemsreliabilityassessment_amd import api
    eng = api.Engine(case)
    acc = eng.nsq_accumulate(1, 0, 100000)
    assert (int(n), int(n_fail)) == (acc.n, acc.n_fail) and float(sum_dns) == pytest.approx(acc.sum_dns, rel=1e-12)
    assert 0 < int(nd) < 100000
    eng.close()
