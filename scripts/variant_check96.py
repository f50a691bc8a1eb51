"""Timing + parity of kernel build variants on RTS-96:  python scripts/variant_check96.py <lib in csrc/ablate | base> ..."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, json; sys.path.insert(0, %r)
import numpy as np
from powersystemsreliabilityassessment_amd import api, case96
c = case96.rts96(); eng = api.Engine(c)
d = json.load(open(%r + "/tests/golden/rts96_states_fixture.json"))
st = np.zeros((len(d["states"]), c.ncomp), np.uint8)
for i, x in enumerate(d["states"]): st[i, x["failed"]] = 1
res = []
for name, pol in (("emulate", 0), ("physical", 1)):
    dns, nodal, info = eng.mc_simulation(st, mpopt=api.mpoption(pol), return_info=True)
    it = np.array([x[name]["iters"] for x in d["states"]]); ss = np.array([x[name]["status"] for x in d["states"]]); dn = np.array([x[name]["dns"] for x in d["states"]])
    res.append((int((np.abs(info["iters"] - it) > 0).sum()), int((info["status"] != ss).sum()), float(np.abs(dns - dn).max())))
eng.nsq_accumulate(1, 0, 100000)
ts = []
for k in range(3):
    acc = eng.nsq_accumulate(1, 1000000 * (k + 1), 1000000); ts.append(eng.last_kernel_ms())
print("ms %%.3f  fixture(iters!=, status!=, max|ddns|) %%s  sum_iters %%d sum_dns %%.9f nfail %%d nc %%d" %% (min(ts), res, acc.sum_iters, acc.sum_dns, acc.n_fail, acc.n_nonconverged))
''' % (ROOT, ROOT)
for v in sys.argv[1:]:
    env = dict(os.environ)
    if v != "base":
        env["RELMC_LIB_PATH"] = os.path.join(ROOT, "powersystemsreliabilityassessment_amd/csrc/ablate", v + ".so")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print(v, out.stdout.strip(), out.stderr.strip()[-300:], flush=True)
