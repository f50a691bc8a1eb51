import sys, time, numpy as np
sys.path.insert(0, '.')
from powersystemsreliabilityassessment_amd import api, case24, case96, _abi, dist
def split(a):
    i, d = a.to_arrays(); return np.concatenate([i[:5], i[6:-1]]), int(i[-1]), d
for case, n, ch in ((case24.rts24(), 2 * 10**9, 10**8), (case96.rts96(), 2 * 10**8, 3 * 10**7)):
    eng = api.Engine(case)
    tot = [_abi.Acc(), _abi.Acc()]; tt = [0.0, 0.0]; nd = 0
    for lo in range(0, n, ch):
        m = min(ch, n - lo)
        t = time.time(); a = eng.nsq_accumulate(9, lo, m, api.mpoption(screen=1)); tt[0] += time.time() - t
        t = time.time(); b, k = eng.nsq_accumulate_distinct(9, lo, m, api.mpoption(screen=1)); tt[1] += time.time() - t
        tot[0] = dist.merge(tot[0], a); tot[1] = dist.merge(tot[1], b); nd += k
    ia, sa, da = split(tot[0]); ib, sb, db = split(tot[1])
    print(f"{case.nb} buses, {n:.3g} samples: fused screened {tt[0]:.1f} s, screened per-batch dedupe {tt[1]:.1f} s ({n / tt[1] / 1e6:.0f} M/s), {nd} distinct states solved; "
          f"integers identical: {bool(np.array_equal(ia, ib))}, n_screened equal: {sa == sb}, sums max rel diff {np.max(np.abs(da - db) / np.maximum(np.abs(da), 1e-300)):.2e}")
    eng.close()
