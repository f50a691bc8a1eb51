"""Non-converged units of the primary order, before any retry, under the elimination order the case ships (argv[3] = "rule": the library's rule, or a file with a comma-separated order):
python scripts/order_soak.py rts96 1e8   (developer tool)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powersystemsreliabilityassessment_amd import api, case24, case96
name = sys.argv[1]; n = int(float(sys.argv[2]))
import numpy as np
order = "case" if len(sys.argv) < 4 else (None if sys.argv[3] == "rule" else np.array([int(v) for v in open(sys.argv[3]).read().strip().split(",")], np.int32))
e = api.Engine(case24.rts24() if name == "rts24" else case96.rts96(), elim_order=order)
for pol in (0, 1):
    u0 = e.retry_stats(); d0 = e.retry_dense_stats(); t = time.time()
    acc = e.nsq_accumulate(3, 0, n, api.mpoption(pol)); dt = time.time() - t
    u1 = e.retry_stats(); d1 = e.retry_dense_stats()
    print("%s policy %d: %d samples in %.1f s (%.2f M/s): to the further orders %d (converged there %d), dense %d (%d), non-converged %d, EDNS %.6f, primary order %s" % (
        name, pol, n, dt, n / dt / 1e6, u1[0] - u0[0], u1[1] - u0[1], d1[0] - d0[0], d1[1] - d0[1], acc.n_nonconverged, acc.sum_dns / acc.n, e.case_order()), flush=True)
