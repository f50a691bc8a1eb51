"""Upper bound of removing the 16-lane tile's fourth injection slot (it holds 2 of RTS-24's 50 injections): RTS-24 with the loads of
buses 19 and 20 moved onto bus 16's load (48 injections), on the shipped tile (4 slots) and on a 3-slot build (RELMC_LIB_PATH).
Developer tool; results of the modified case mean nothing, only the time does."""
import os, sys, dataclasses
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from powersystemsreliabilityassessment_amd import api, case24
c = case24.rts24()
keep = np.ones(c.ninj, bool); keep[-2:] = False            # the last two virtual generators (loads at buses 19, 20)
pd = c.bus_pd.copy(); moved = pd[[18, 19]].sum(); pd[[18, 19]] = 0.0; pd[15] += moved
inj_pmin = c.inj_pmin[keep].copy(); j16 = c.ng + list(np.flatnonzero(c.bus_pd != 0)).index(15); inj_pmin[j16] -= moved
c2 = dataclasses.replace(c, nd=c.nd - 2, bus_pd=pd, inj_bus=c.inj_bus[keep], inj_pmin=inj_pmin, inj_pmax=c.inj_pmax[keep], inj_cost=c.inj_cost[keep])
eng = api.Engine(c2)
eng.nsq_accumulate(1, 0, 200000)
ts = []
for k in range(3):
    acc = eng.nsq_accumulate(1, 1000000 * (k + 1), 1000000); ts.append(eng.last_kernel_ms())
print("ninj", c2.ng + c2.nd, "ms %.3f" % min(ts), "mean iters %.3f" % (acc.sum_iters / acc.n), "edns %.4f" % (acc.sum_dns / acc.n))
