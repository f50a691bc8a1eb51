import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from powersystemsreliabilityassessment_amd import api, case24, case96
for name, case, n in (("rts24", case24.rts24(), 1_000_000), ("rts96", case96.rts96(), 300_000)):
    eng = api.Engine(case)
    for pol in (0, 1):
        a = eng.nsq_accumulate(1, 0, n, api.mpoption(pol)); t0 = eng.last_kernel_ms()
        b = eng.nsq_accumulate(1, 0, n, api.mpoption(pol, screen=1)); t1 = eng.last_kernel_ms()
        b = eng.nsq_accumulate(1, 0, n, api.mpoption(pol, screen=1)); t1 = eng.last_kernel_ms()
        ai, ad = a.to_arrays(); bi, bd = b.to_arrays()
        print(name, pol, "ms %.3f -> %.3f" % (t0, t1), "n_screened", b.n_screened, "share %.4f" % (b.n_screened / n))
        print("   ints equal (but iters, screened):", np.array_equal(ai[:5], bi[:5]), np.array_equal(ai[6:-1], bi[6:-1]), "iters", ai[5], bi[5])
        print("   doubles max rel diff", np.max(np.abs(ad - bd) / np.maximum(np.abs(ad), 1e-300)), "sum_dns", ad[0], bd[0])
    eng.close()
