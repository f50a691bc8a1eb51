"""nsqMain with the reference's own batch of 100 samples (nsqMain.m:60): many checkpoints per launch (default) against one
launch per checkpoint (python scripts/small_batch.py off: the context's diagnosis switch nsq_no_stretch, relmc_debug_set)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powersystemsreliabilityassessment_amd import api
e = api.Engine()
OFF = len(sys.argv) > 1 and sys.argv[1] == "off"
e.debug_set("nsq_no_stretch", OFF)
e.nsq_accumulate(1, 0, 100000)
for n, b in ((100000, 100), (100000, 100), (1000000, 100), (1000000, 1000), (1000000, 10000)):
    t = time.perf_counter(); r = e.nsqMain(beta_limit=0.0, max_iterations=n, samples_per_batch=b, seed=1); dt = time.perf_counter() - t
    print("stretch=%s n=%d batch=%d: %.2f ms wall (library %.2f ms, kernels %.2f ms), %d checkpoints, EDNS %.6f beta %.5f" % (
        "off" if OFF else "on", n, b, dt * 1e3, r.elapsed_time * 1e3, r.kernel_seconds * 1e3, len(r.beta_history), r.accumulated_edns, r.current_beta))
for mode in ("database", True):
    for n, b in ((100000, 100), (100000, 100), (1000000, 1000)):
        t = time.perf_counter(); r = e.nsqMain(beta_limit=0.0, max_iterations=n, samples_per_batch=b, seed=1, distinct_states=mode); dt = time.perf_counter() - t
        print("distinct_states=%s n=%d batch=%d: %.2f ms wall (library %.2f ms, kernels %.2f ms), %d checkpoints, EDNS %.6f beta %.5f" % (
            mode, n, b, dt * 1e3, r.elapsed_time * 1e3, r.kernel_seconds * 1e3, len(r.beta_history), r.accumulated_edns, r.current_beta))
