"""State-database path (nsqMain.m:220-278 on the device): samples/s per batch as the database fills, time to beta limits.
  python scripts/db_rate.py [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powersystemsreliabilityassessment_amd import api, case24, case96
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
for name, case in (("rts24", case24.rts24()), ("rts96", case96.rts96())):
    eng = api.Engine(case)
    eng.nsq_accumulate(1, 0, 65536)
    eng.db_reset(); eng.nsq_db_batch(1, 0, B); eng.db_reset()           # warm-up (allocations, rocprim temp storage)
    for k in range(8 if name == "rts24" else 3):
        t = time.perf_counter(); acc, st = eng.nsq_db_batch(1, k * B, B); dt = time.perf_counter() - t
        print(f"{name} batch {k}: {dt*1e3:8.3f} ms  {B/dt/1e6:9.1f} M samples/s  rows {st.rows} new {st.new_rows} distinct {st.batch_distinct}", flush=True)
    t = time.perf_counter(); a = eng.nsq_accumulate(1, 0, B); dt = time.perf_counter() - t
    print(f"{name} per-sample path: {dt*1e3:.3f} ms wall, kernel {eng.last_kernel_ms():.3f} ms")
    if name == "rts24":
        for lim, batch in ((0.01, 100_000), (0.0017, 1_000_000)):
            for mode in (0, 1, "database"):
                t = time.perf_counter(); r = eng.nsqMain(beta_limit=lim, max_iterations=50_000_000, samples_per_batch=batch, seed=1, distinct_states=mode); dt = time.perf_counter() - t
                print(f"  beta<{lim}: mode {mode!s:9} {dt*1e3:9.2f} ms  samples {r.current_iteration} beta {r.current_beta:.6f} edns {r.accumulated_edns:.6f} rows {r.database_row_count}", flush=True)
    eng.close()
