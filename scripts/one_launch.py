"""One launch of the fused kernel after a warm-up (the program rocprofv3's PMC passes profile: scripts/pmc.sh, scripts/pmc_classes.sh):
    python scripts/one_launch.py [samples] [rts24|rts96] [max_it]
With max_it (below the 9 iterations the fastest state needs) every scenario row runs exactly max_it Newton steps: the difference of the counters of
two such launches is the instruction count of ONE trip of the interior-point loop, as the hardware counted it (scripts/isa_budget.py).
Prints kernel time, mean interior-point iterations per scenario, and the mean over scenario groups of the group's LARGEST iteration count
(a wavefront iterates until the slowest of its rows has converged; groups = consecutive 4 scenarios of the hard / easy ordered windows on the
16-lane tile are not reproduced here: the figure printed is over consecutive groups of the sampled order and is an upper bound on the kernel's)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from powersystemsreliabilityassessment_amd import api, case24, case96
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
case = case96.rts96() if len(sys.argv) > 2 and sys.argv[2] == "rts96" else case24.rts24()
max_it = int(sys.argv[3]) if len(sys.argv) > 3 else 0
eng = api.Engine(case, debug_switches=("no_retry",) if max_it else ())
opts = api.mpoption(max_it=max_it) if max_it else api.mpoption()
eng.nsq_accumulate(1, 0, 65536, opts)
acc = eng.nsq_accumulate(1, 1000000, n, opts)
ms = eng.last_kernel_ms()
if max_it:
    print("kernel_ms", ms, "scen/s", n / ms * 1e3, "iters", acc.sum_iters / acc.n, "mean_max_iters", float(max_it), "nonconverged", acc.n_nonconverged, "singular", acc.n_singular)
    sys.exit(0)
# per-group maximum of the iteration counts: from the materialised path on a slice of the same samples
m = min(n, 200000)
st = eng.mc_sampling(None, m, seed=1, first_index=1000000)
_, _, info = eng.mc_simulation(st, return_info=True)
rows = 4 if case.nb <= 32 else 1
it = info["iters"][: m // rows * rows].reshape(-1, rows)
print("kernel_ms", ms, "scen/s", n / ms * 1e3, "iters", acc.sum_iters / acc.n, "mean_max_iters", float(it.max(1).mean()))
