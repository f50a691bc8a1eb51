import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def case():
    from powersystemsreliabilityassessment_amd import case24
    return case24.rts24()


@pytest.fixture(scope="session")
def oracle(case):
    """CPU restatement (test infrastructure, oracle/relmc_oracle.c)."""
    from oracle import coracle
    return coracle.Oracle(case)


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(GOLDEN, "nsq_golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def states_fixture(case):
    with open(os.path.join(GOLDEN, "states_fixture.json")) as f:
        d = json.load(f)
    st = np.zeros((len(d["states"]), case.ncomp), dtype=np.uint8)
    for i, x in enumerate(d["states"]):
        st[i, x["failed"]] = 1
    d["matrix"] = st
    return d


@pytest.fixture(scope="session")
def nsq_fixture():
    with open(os.path.join(GOLDEN, "nsq_seed1_1e5.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def engine(case):
    """The product path: HIP library through the C ABI.  Only for @pytest.mark.gpu tests."""
    from powersystemsreliabilityassessment_amd import api
    eng = api.Engine(case, device=0)
    yield eng
    eng.close()
