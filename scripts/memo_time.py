import sys; sys.path.insert(0,'/root/repo')
import time
from powersystemsreliabilityassessment_amd import api, case96
for name, eng in (("rts24", api.Engine()), ("rts96", api.Engine(case96.rts96()))):
    for n in (100000, 1000000, 10000000):
        eng.nsq_accumulate_distinct(1, 0, n)
        t=time.time(); acc, nd = eng.nsq_accumulate_distinct(1, n, n); dt=time.time()-t
        print(name, n, "distinct", nd, "%.4f"%(nd/n), "wall ms %.2f"%(dt*1e3), "reported ms %.2f"%eng.last_kernel_ms(), "scen/s %.3e"%(n/dt))
    t=time.time(); r=eng.nsqMain(beta_limit=0.01, max_iterations=20_000_000, samples_per_batch=100_000, distinct_states=True); print(name, "to 1% distinct:", time.time()-t, r.current_iteration, r.current_beta)
    t=time.time(); r=eng.nsqMain(beta_limit=0.01, max_iterations=20_000_000, samples_per_batch=100_000); print(name, "to 1% per-sample:", time.time()-t, r.current_iteration, r.current_beta)
