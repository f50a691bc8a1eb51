"""Converged indices of the three HL2 workloads on one GPU (for DESIGN.md section 5)."""
import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np
from powersystemsreliabilityassessment_amd import api, case96, seq, dist
for name, case in (("rts24", None), ("rts96", case96.rts96())):
    e = api.Engine(case) if case is not None else api.Engine()
    for pol, pn in ((api.REFERENCE_EMULATE, "emulate"), (api.PHYSICAL, "physical")):
        t=time.time(); n = 100_000_000 if name == "rts24" else 30_000_000
        acc = e.nsq_accumulate(1, 0, n, api.mpoption(pol)); dt=time.time()-t
        ix = dist.indices_from_acc(acc, e.case.nb, e.case.ncomp)
        top = np.argsort(-ix["comp_importance"])[:4]
        print(name, pn, "n=%d wall %.2fs EDNS %.4f MW PLC %.6f LOLE %.2f h/yr beta %.5f iters %.3f sing %d infeas %d nc %d" % (n, dt, ix["edns"], ix["plc"], ix["lole"], ix["beta"], ix["mean_iters"], acc.n_singular, acc.n_infeasible, acc.n_nonconverged),
              "top comps", top.tolist(), np.round(ix["comp_importance"][top],4).tolist(), "imp(L11)" if name=="rts24" else "", round(float(ix["comp_importance"][43]),5) if name=="rts24" else "")
        if name == "rts24": print("   nodal EENS MW", np.round(ix["nodal_eens"],4).tolist())
    e.close()
e = api.Engine(); s = seq.SeqEngine(e)
for pol, pn in ((api.REFERENCE_EMULATE, "emulate"), (api.PHYSICAL, "physical")):
    t=time.time(); ens=[]; dlc=[]; nlc=[]
    for b in range(20):
        a,d,n_,_,acc = s.seq_years(5, b*1000, 1000, api.mpoption(pol)); ens.append(a); dlc.append(d); nlc.append(n_)
    ens=np.concatenate(ens); dlc=np.concatenate(dlc); nlc=np.concatenate(nlc)
    print("seq", pn, "20000 years wall %.2fs EENS %.1f +- %.1f MWh/yr LOLE %.3f h/yr LOLF %.4f occ/yr" % (time.time()-t, ens.mean(), ens.std(ddof=1)/np.sqrt(ens.size), dlc.mean(), nlc.mean()))
