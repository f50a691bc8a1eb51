"""Aggregate per-bus nodal curtailment over the RTS-96 fixture states (317 sampled + 67 device-numfail states), device against the
C oracle, both policies: how tight can the per-bus pin of tests/test_rts96.py be?  (developer tool, needs a GPU and the oracle)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np
from powersystemsreliabilityassessment_amd import api, case96, _abi
from oracle import coracle
c = case96.rts96(); eng = api.Engine(c); orc = coracle.Oracle(c)
mats = []
for fn in ("rts96_states_fixture.json", "rts96_numfail_fixture.json"):
    d = json.load(open(os.path.join(ROOT, "tests/golden", fn)))
    st = np.zeros((len(d["states"]), c.ncomp), np.uint8)
    for i, x in enumerate(d["states"]): st[i, x["failed"]] = 1
    mats.append(st)
st = np.vstack(mats)
for name, pol in (("emulate", 0), ("physical", 1)):
    dns, nodal, info = eng.mc_simulation(st, mpopt=api.mpoption(pol), return_info=True)
    r = orc.mc_simulation(st, pol, nthreads=16)
    g, o = nodal.sum(0), r["nodal"].sum(0)
    m = o > 0
    rel = np.abs(g[m] - o[m]) / o[m]
    print(name, "states", len(st), "total", g.sum(), o.sum(), "per-bus rel diff: max %.3e median %.3e; abs max %.3f MW of bus sum %.1f" % (rel.max(), np.median(rel), np.abs(g - o).max(), o[np.argmax(np.abs(g - o))]))
    per_state = np.abs(nodal - r["nodal"]).max(1)
    print("   per-state max per-bus diff: max %.2f MW, mean %.3f MW; states with > 1 MW: %d" % (per_state.max(), per_state.mean(), int((per_state > 1).sum())))
    conv = (info["status"] == 0) & (r["status"] == 0)
    g2, o2 = nodal[conv].sum(0), r["nodal"][conv].sum(0); m2 = o2 > 0
    print("   both converged (%d states): per-bus rel diff max %.3e" % (int(conv.sum()), (np.abs(g2[m2] - o2[m2]) / o2[m2]).max()))
n = 20000
import time; t0 = time.time()
acc = eng.nsq_accumulate(7, 123456, n); ref = orc.nsq_accumulate(7, 123456, n, 0)
ad, rd = acc.to_arrays()[1], ref.to_arrays()[1]
m = rd[2:] > 0
print("accumulate n=%d: oracle %.1f s; per-bus rel diff max %.3e abs max %.3f" % (n, time.time() - t0, (np.abs(ad[2:][m] - rd[2:][m]) / rd[2:][m]).max(), np.abs(ad[2:] - rd[2:]).max()))
