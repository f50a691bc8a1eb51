"""Non-converged scenarios before and after the second elimination order: RTS-96, first 1e8 samples of seed 1 (67 without it)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powersystemsreliabilityassessment_amd import api, case96, case24
N96 = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
for name, case, n in (("RTS-96", case96.rts96(), N96), ("RTS-24", case24.rts24(), 1_000_000_000)):
    e = api.Engine(case)
    for pol in (0, 1):
        t = time.perf_counter(); acc = e.nsq_accumulate(1, 0, n, api.mpoption(pol)); dt = time.perf_counter() - t
        print("%s policy %d: %d samples in %.1f s, non-converged %d, second attempts (units, converged) %s, dense last resort (units, converged) %s, EDNS %.6f" % (
            name, pol, n, dt, acc.n_nonconverged, e.retry_stats(), e.retry_dense_stats(), acc.sum_dns / acc.n), flush=True)
    e.close()
