"""Sequential track on RTS-96: which simulated years contain an hour the primary elimination order does not converge on?
(relmc_retry_stats deltas per block of years; feeds tests/test_rts96.py::test_gpu96_sequential_retry)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powersystemsreliabilityassessment_amd import api, case96, seq
e = api.Engine(case96.rts96())
sq = seq.SeqEngine(e, reliability_data=case96.seqmeantime96())
blk = 25
t = time.perf_counter(); tot = 0
for y0 in range(0, 3000, blk):
    u0, c0 = e.retry_stats()
    ens, dlc, nlc, ncont, acc = sq.seq_years(1, y0, blk)
    u1, c1 = e.retry_stats()
    tot += int(ncont.sum())
    if u1 > u0:
        print("years [%d, %d): %d hour(s) re-evaluated, %d converged; non-converged left %d; LPs in block %d" % (y0, y0 + blk, u1 - u0, c1 - c0, acc.n_nonconverged, int(ncont.sum())), flush=True)
print("scanned %d hourly LPs in %.1f s, second attempts %s" % (tot, time.perf_counter() - t, e.retry_stats()))
