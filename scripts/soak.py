import sys, time; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powersystemsreliabilityassessment_amd import api, case96, dist
for name, case, n in (("rts24", None, 1_000_000_000), ("rts96", case96.rts96(), 100_000_000)):
    e = api.Engine(case) if case is not None else api.Engine()
    for pol, pn in ((api.REFERENCE_EMULATE, "emulate"), (api.PHYSICAL, "physical")):
        t=time.time(); acc = e.nsq_accumulate(3, 0, n, api.mpoption(pol)); dt=time.time()-t
        ix = dist.indices_from_acc(acc, e.case.nb, e.case.ncomp)
        print(name, pn, "n=%d wall %.1fs rate %.2f M/s EDNS %.4f PLC %.6f beta %.6f iters %.4f sing %d infeas %d nc %d"%(n, dt, n/dt/1e6, ix["edns"], ix["plc"], ix["beta"], ix["mean_iters"], acc.n_singular, acc.n_infeasible, acc.n_nonconverged), flush=True)
