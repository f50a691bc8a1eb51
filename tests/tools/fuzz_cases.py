"""Fuzz run over random networks (the generator of tests/test_random_cases.py, other seeds and sizes): device against the C oracle on
sampled states, both policies; prints one line per case and every disagreement in status, dns (> 1e-5 MW) or iterations (> 1)."""
import importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
spec = importlib.util.spec_from_file_location("trc", os.path.join(ROOT, "tests/test_random_cases.py"))
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
from powersystemsreliabilityassessment_amd import api
from oracle import coracle
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ORDER = sys.argv[2] if len(sys.argv) > 2 else None           # "tune": every case under an elimination order tuned on the spot (relmc_tune_order)
tot = dict(states=0, status=0, dns=0, it1=0, it2=0, retried=0, dense=0, dense_conv=0, nc_device=0, nc_oracle=0)
for k, (seed, nb, chords, ng, lbs, tight, par, pminf) in enumerate(m.fuzz_stream(n_cases)):
    try:
        case = m.random_case(np.random.default_rng(seed), nb, chords, ng, lbs, tight, par, pminf)
        eng = api.Engine(case, elim_order=ORDER)
    except Exception as ex:
        print("case %2d nb %3d: not loaded (%s)" % (k, nb, str(ex)[-70:])); continue
    orc = coracle.Oracle(case)
    n = 2000 if nb <= 32 else 800
    st = eng.mc_sampling(None, n, seed=seed, first_index=0)
    line = "case %2d nb %3d nl %3d ninj %3d:" % (k, nb, case.nl, case.ninj)
    for pol in (0, 1):
        dns, nodal, info = eng.mc_simulation(st, mpopt=api.mpoption(pol), return_info=True)
        ref = orc.mc_simulation(st, pol, nthreads=16)
        bad = info["status"] != ref["status"]
        ok = ~bad & (ref["status"] == 0)
        dd = np.abs(dns - ref["dns"])[ok]; di = np.abs(info["iters"] - ref["iters"])[ok]
        ncd, nco = int(np.isin(info["status"], (1, 2)).sum()), int(np.isin(ref["status"], (1, 2)).sum())
        tot["nc_device"] += ncd; tot["nc_oracle"] += nco
        tot["states"] += n; tot["status"] += int(bad.sum()); tot["dns"] += int((dd > 1e-5).sum()); tot["it1"] += int((di == 1).sum()); tot["it2"] += int((di > 1).sum())
        line += " pol%d status!= %d, non-converged device %d / oracle %d, max|ddns| %.1e, iters +-1: %d, >1: %d;" % (pol, bad.sum(), ncd, nco, dd.max() if dd.size else 0.0, (di == 1).sum(), (di > 1).sum())
        for i in np.flatnonzero(bad)[:3]:
            line += " [state %d: device %d/%d it, oracle %d/%d it, dns %.6f vs %.6f]" % (i, info["status"][i], info["iters"][i], ref["status"][i], ref["iters"][i], dns[i], ref["dns"][i])
    tot["retried"] += eng.retry_stats()[0]; tot["dense"] += eng.retry_dense_stats()[0]; tot["dense_conv"] += eng.retry_dense_stats()[1]
    print(line + " second attempts %s dense %s order %s" % (eng.retry_stats(), eng.retry_dense_stats(), eng.case_order()), flush=True)
    eng.close()
print("total", tot)
