"""What the device returns TODAY for the 51 RTS-96 states of tests/golden/rts96_numfail_fixture.json through the production entry point
(relmc_mc_simulation: primary order, then the further static orders, then the dense pivoted solve), per policy and per state: status,
iterations, dns.  Run on the GPU box; tests/golden/make_golden.py --numfail96-device <this JSON> records it in the fixture, so the
parity test pins the iteration count of every retried state instead of a blanket tolerance.
    python tests/tools/numfail96_device.py > gpurun_out/numfail96_device.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from powersystemsreliabilityassessment_amd import api, case96, _lib

d = json.load(open(os.path.join(ROOT, "tests", "golden", "rts96_numfail_fixture.json")))
case = case96.rts96()
st = np.zeros((len(d["states"]), case.ncomp), dtype=np.uint8)
for i, x in enumerate(d["states"]):
    st[i, x["failed"]] = 1
eng = api.Engine(case)
out = {"code_object_sha256": _lib.code_object_sha256(), "version": _lib.load().relmc_version().decode(), "n_states": len(d["states"])}
for name, pol in (("emulate", api.REFERENCE_EMULATE), ("physical", api.PHYSICAL)):
    runs = []
    for rep in range(2):                                   # twice: the result must not depend on what ran before
        before = eng.retry_stats()
        dns, nodal, info = eng.mc_simulation(st, mpopt=api.mpoption(pol), return_info=True)
        after = eng.retry_stats()
        runs.append((info["status"].tolist(), info["iters"].tolist(), [float(v) for v in dns]))
    assert runs[0][:2] == runs[1][:2], "device results of two identical calls differ"
    out[name] = {"status": runs[0][0], "iters": runs[0][1], "dns": runs[0][2], "retried_units": after[0] - before[0], "retried_converged": after[1] - before[1]}
print(json.dumps(out))
