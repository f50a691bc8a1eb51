"""Newton steps of one state, device (RELMC_TRACE build) against the numpy MIPS restatement (pivoted LU of the unreduced system):
per iteration the largest relative difference of dtheta, dlambda, dp and where it sits.  usage: step_compare.py <nb> <state index>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("RELMC_LIB_PATH", os.path.join(ROOT, "powersystemsreliabilityassessment_amd/csrc/ablate/librelmc_trace.so"))
import ctypes as C
import importlib.util
import numpy as np
spec = importlib.util.spec_from_file_location("trc", os.path.join(ROOT, "tests/test_random_cases.py"))
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
from powersystemsreliabilityassessment_amd import api
from oracle import pyoracle as po
nbw, idx = int(sys.argv[1]), int(sys.argv[2])
s = [c for c in m.CASES if c[1] == nbw][0]
seed, nb, chords, ng, lbs, tight, par, pminf = s
case = m.random_case(np.random.default_rng(1000 + seed), nb, chords, ng, lbs, tight, par, pminf)
E = api.Engine(case)
E.L.relmc_debug_trace.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int32]
st = E.mc_sampling(None, idx + 1, seed=seed, first_index=0)[idx:idx + 1]
dns, nodal, info = E.mc_simulation(st, mpopt=api.mpoption(0), return_info=True)
N = 512 + 512 * 41
out = (C.c_double * N)(); E.L.relmc_debug_trace(E._h, out, N)
t = np.array(out)
bext = t[512 + 512 * 40: 512 + 512 * 40 + nb].astype(int)
nit = int(info["iters"][0])
# numpy trajectory with the steps recorded
steps = []
lp = po.build_lp(case, st[0], 0)
orig = np.linalg.solve
def rec(K, rhs):
    x = orig(K, rhs); steps.append(x.copy()); return x
np.linalg.solve = rec
x, f, eflag, it = po.mips_lp(lp["c"], lp["A"], lp["l"], lp["u"], lp["xmin"], lp["xmax"], lp["x0"])
np.linalg.solve = orig
nx = lp["x0"].size; inj = lp["inj_idx"]
print("device iterations", nit, "numpy", it, "failed components", np.flatnonzero(st[0]))
# rows of Ae: pinned variables first (from the identity block), then the balance rows in bus order
neq = steps[0].size - nx
pins = neq - (nb - len(lp["pre"]["drop_bal"]))
bal_rows = [i for i in range(nb) if i not in lp["pre"]["drop_bal"]]
for k in range(min(nit, it)):
    o = t[512 + 512 * (k + 1): 512 + 512 * (k + 2)]      # the kernel counts the step being computed from 1
    dth = np.zeros(nb); dla = np.zeros(nb); dth[bext] = o[:nb]; dla[bext] = o[128:128 + nb]
    dp = o[256:256 + case.ninj][inj]
    ref = steps[k]
    rth, rp, rl = ref[:nb], ref[nb:nx], np.zeros(nb)
    rl[bal_rows] = ref[nx + pins:]
    def rel(a, b): return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
    jl = int(np.argmax(np.abs(dla - rl))); jp = int(np.argmax(np.abs(dp - rp)))
    jt = int(np.argmax(np.abs(dth - rth)))
    print("step %2d  dtheta %.1e  dlambda %.1e (bus %d: %.6e vs %.6e)  dp %.1e (inj %d: %.6e vs %.6e)" % (
        k + 1, rel(dth, rth), rel(dla, rl), jl, dla[jl], rl[jl], rel(dp, rp), inj[jp], dp[jp], rp[jp]))
