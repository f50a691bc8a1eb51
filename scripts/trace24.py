"""Per-iteration termination quantities (feascond, gradcond, compcond, costcond, alpha_p, alpha_d, gamma, f) of single RTS-24 states
on the device (-DRELMC_TRACE build passed with RELMC_LIB_PATH):  python scripts/trace24.py 23,32 [22,32 ...]"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ctypes as C
from powersystemsreliabilityassessment_amd import case24, api
c = case24.rts24(); E = api.Engine(c)
E.L.relmc_debug_trace.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int32]
for arg in sys.argv[1:]:
    fl = [int(x) for x in arg.split(",")]
    st = np.zeros((1, c.ncomp), np.uint8); st[0, fl] = 1
    dns, nodal, info = E.mc_simulation(st, mpopt=api.mpoption(0), return_info=True)
    out = (C.c_double * (8 * 40))(); E.L.relmc_debug_trace(E._h, out, 8 * 40)
    t = np.array(out).reshape(40, 8)
    print(fl, "dns", dns, "iters", info["iters"], "nodal", np.round(nodal[0], 3).tolist())
    for it in range(int(info["iters"][0]) + 1): print(it, " ".join("%12.5e" % v for v in t[it]))
