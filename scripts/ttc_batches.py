"""Wall time to EENS CoV < 1 % as a function of the checkpoint spacing (the reference checks every 100 samples, nsqMain.m:60,299-312)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powersystemsreliabilityassessment_amd import api
eng = api.Engine()
eng.nsqMain(beta_limit=0.01, max_iterations=5_000_000, samples_per_batch=100_000, seed=1)
for batch in (100, 1000, 8192, 20000, 50000, 100000, 300000):
    ts = []
    for rep in range(5):
        t = time.perf_counter(); r = eng.nsqMain(beta_limit=0.01, max_iterations=5_000_000, samples_per_batch=batch, seed=1); ts.append(time.perf_counter() - t)
    print("batch %6d: %.2f ms (min of 5; median %.2f), stops at %d samples, beta %.5f, EDNS %.4f" % (batch, 1e3 * min(ts), 1e3 * sorted(ts)[2], r.current_iteration, r.current_beta, r.accumulated_edns), flush=True)
for batch in (100, 8192, 100000):
    ts = []
    for rep in range(5):
        t = time.perf_counter(); r = eng.nsqMain(beta_limit=0.01, max_iterations=5_000_000, samples_per_batch=batch, seed=1, distinct_states="database"); ts.append(time.perf_counter() - t)
    print("database, batch %6d: %.2f ms (min of 5; median %.2f), stops at %d samples, beta %.5f, rows %d" % (batch, 1e3 * min(ts), 1e3 * sorted(ts)[2], r.current_iteration, r.current_beta, r.database_row_count), flush=True)
