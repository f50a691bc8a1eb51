"""Per-iteration termination trace (RELMC_TRACE build, `make -C .../csrc ablate/librelmc_trace.so`) of one state of one of the random
cases of tests/test_random_cases.py.  usage: random_debug.py <nb> <state index>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RELMC_LIB_PATH", os.path.join(ROOT, "powersystemsreliabilityassessment_amd/csrc/ablate/librelmc_trace.so"))
import ctypes as C
import importlib.util
import numpy as np
spec = importlib.util.spec_from_file_location("trc", os.path.join(ROOT, "tests/test_random_cases.py"))
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
from powersystemsreliabilityassessment_amd import api
nbw, idx = int(sys.argv[1]), int(sys.argv[2])
s = [c for c in m.CASES if c[1] == nbw][0]
seed, nb, chords, ng, lbs, tight, par, pminf = s
case = m.random_case(np.random.default_rng(1000 + seed), nb, chords, ng, lbs, tight, par, pminf)
E = api.Engine(case)
E.L.relmc_debug_trace.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int32]
st = E.mc_sampling(None, idx + 1, seed=seed, first_index=0)[idx:idx + 1]
dns, nodal, info = E.mc_simulation(st, mpopt=api.mpoption(0), return_info=True)
out = (C.c_double * (8 * 40))(); E.L.relmc_debug_trace(E._h, out, 8 * 40)
t = np.array(out).reshape(40, 8)
print(np.flatnonzero(st[0]), dns, info)
print("it   feascond   gradcond   compcond   costcond     alphap     alphad      gamma       f")
for it in range(int(info["iters"][0]) + 1):
    print(it, " ".join("%10.3e" % v for v in t[it]))
