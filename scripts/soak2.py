"""Round-2 soak: 1e10 RTS-24 samples every sample solved and through the state database, 1e9 RTS-96 samples (emulate policy)."""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from powersystemsreliabilityassessment_amd import api, case96, dist, _abi
e = api.Engine()
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000_000
t = time.time(); tot = _abi.Acc()
for lo in range(0, N, 2_000_000_000):
    tot = dist.merge(tot, e.nsq_accumulate(3, lo, min(2_000_000_000, N - lo)))
dt = time.time() - t
ix = dist.indices_from_acc(tot, 24, 71)
print("rts24 per-sample: n %d in %.1f s (%.1f M/s) EDNS %.5f beta %.6f PLC %.6f nonconverged %d singular %d second attempts (units, converged) %s dense (units, converged) %s" % (tot.n, dt, tot.n / dt / 1e6, ix["edns"], ix["beta"], ix["plc"], tot.n_nonconverged, tot.n_singular, e.retry_stats(), e.retry_dense_stats()), flush=True)
t = time.time(); r = e.nsqMain(beta_limit=0.0, max_iterations=N, samples_per_batch=50_000_000, seed=3, distinct_states="database"); dt = time.time() - t
di, dd = r.acc.to_arrays(); ti, td = tot.to_arrays()
print("rts24 database: n %d in %.1f s (%.1f M/s) rows %d ints equal %s max rel diff %.2e" % (r.current_iteration, dt, N / dt / 1e6, r.database_row_count, np.array_equal(di, ti), np.max(np.abs(dd - td) / np.maximum(np.abs(td), 1e-300))), flush=True)
e.close()
e = api.Engine(case96.rts96()); N96 = N // 10
t = time.time(); a = e.nsq_accumulate(3, 0, N96); dt = time.time() - t
print("rts96 per-sample: n %d in %.1f s (%.1f M/s) EDNS %.5f PLC %.6f nonconverged %d (%.2e) second attempts (units, converged) %s dense (units, converged) %s" % (a.n, dt, a.n / dt / 1e6, a.sum_dns / a.n, a.n_fail / a.n, a.n_nonconverged, a.n_nonconverged / a.n, e.retry_stats(), e.retry_dense_stats()), flush=True)
