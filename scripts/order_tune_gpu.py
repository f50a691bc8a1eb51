"""Refines an elimination order with the GPU in the loop: local search (swap / move of two buses) where a candidate first has to be no more
than SLACK worse in the scheduler's cost model (host, 1-3 ms) and is then TIMED (case load + three launches of N samples); accepted when it
beats the best time by more than the noise margin, and the best is re-timed every so often.  Developer tool (round 3):
    python scripts/order_tune_gpu.py rts96 <order file> <seconds> [seed]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from powersystemsreliabilityassessment_amd import api, case24, case96
from tests import schedule_interp as si

name, f, budget = sys.argv[1], sys.argv[2], float(sys.argv[3]); seed = int(sys.argv[4]) if len(sys.argv) > 4 else 1
case = case24.rts24() if name == "rts24" else case96.rts96()
N = 1 << 19 if name == "rts24" else 1 << 18
SLACK, MARGIN = 3, 0.0015


def model(order):
    s = si.symbolic(case, 0, order)
    nf = s.npass_upd - s.npass_updh - s.npass_updq
    return nf * 10 + s.npass_updh * 7 + s.npass_updq * 6 + s.npass_inv * 6 + sum(6 if (s.bwd_half >> k) & 1 else 7 for k in range(s.npass_bwd)) + 4 * s.npass


def timed(order, reps=3):
    eng = api.Engine(case, elim_order=np.asarray(order, np.int32))
    eng.nsq_accumulate(1, 0, 65536)
    ts = []
    for k in range(reps):
        eng.nsq_accumulate(1, (k + 1) * N, N); ts.append(eng.last_kernel_ms())
    eng.close()
    return min(ts)


cur = [int(v) for v in open(f).read().strip().split(",")]
rnd = random.Random(seed); nb = case.nb
best_t = timed(cur, 5); best_m = model(cur); t0 = time.time(); n_eval = n_model = 0
print("start: %.3f ms for %d samples, model cost %d" % (best_t, N, best_m), flush=True)
while time.time() - t0 < budget:
    i, j = rnd.randrange(nb - 1), rnd.randrange(nb - 1)
    if i == j: continue
    new = list(cur)
    if rnd.random() < 0.5: new[i], new[j] = new[j], new[i]
    else: new.insert(j, new.pop(i))
    try:
        m = model(new)
    except RuntimeError:
        continue
    n_model += 1
    if m > best_m + SLACK: continue
    t = timed(new); n_eval += 1
    if t < best_t * (1.0 - MARGIN):
        t2 = timed(new, 5)                                   # confirm
        if t2 < best_t * (1.0 - MARGIN):
            cur, best_t, best_m = new, min(t, t2), min(m, best_m) if m < best_m else m
            print("%.0f s, %d timed of %d modelled: %.3f ms, model cost %d" % (time.time() - t0, n_eval, n_model, best_t, m), flush=True)
    if n_eval % 40 == 39:
        best_t = 0.5 * (best_t + timed(cur, 5))              # drift of the box
print("best %.3f ms  order=%s" % (best_t, ",".join(str(v) for v in cur)), flush=True)
