"""Which of the tuned candidate orders keep the parity pins?  For every order file: kernel ms per 1e6, fixture states whose iteration count
moves, and the per-bus nodal sums of 3e5 sampled states against the C oracle (developer tool; uses the oracle, hence lives with the
developer tools under tests/ and is not part of the product):  python tests/tools/order_select.py rts24 <order files...>"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np
from powersystemsreliabilityassessment_amd import api, case24, case96
from oracle import coracle
name = sys.argv[1]
c = case24.rts24() if name == "rts24" else case96.rts96()
orc = coracle.Oracle(c)
d = json.load(open(os.path.join(ROOT, "tests/golden", "states_fixture.json" if name == "rts24" else "rts96_states_fixture.json")))
fx = np.zeros((len(d["states"]), c.ncomp), np.uint8)
for i, x in enumerate(d["states"]): fx[i, x["failed"]] = 1
fit = np.array([x["emulate"]["iters"] for x in d["states"]])
N = 300_000
ref = None
for f in sys.argv[2:]:
    order = None if f == "rule" else np.array([int(v) for v in open(f).read().strip().split(",")], np.int32)
    eng = api.Engine(c, elim_order=order)
    _, _, info = eng.mc_simulation(fx, return_info=True)
    flips = int((info["iters"] != fit).sum())
    st = eng.mc_sampling(None, N, seed=1, first_index=0)
    dns, nodal, info2 = eng.mc_simulation(st, return_info=True)
    if ref is None: ref = orc.mc_simulation(st, 0, nthreads=16)
    nd, no = nodal.sum(0), ref["nodal"].sum(0); m = no > 0
    rel = np.zeros(c.nb); rel[m] = np.abs(nd[m] - no[m]) / no[m]
    eng.nsq_accumulate(1, 0, 200000); ts = []
    for k in range(3): eng.nsq_accumulate(1, 1000000 * (k + 1), 1000000); ts.append(eng.last_kernel_ms())
    print("%-34s ms %.3f  fixture iteration moves %d  iterations != oracle %d of %d  nodal sums rel diff max %.2e (bus %d)" % (
        os.path.basename(f), min(ts), flips, int((info2["iters"] != ref["iters"]).sum()), N, rel.max(), int(np.argmax(rel)) + 1), flush=True)
    eng.close()
