import sys, os; sys.path.insert(0,'/root/repo')
os.environ['RELMC_LIB_PATH']='/root/repo/powersystemsreliabilityassessment_amd/csrc/ablate/librelmc_trace.so'
import numpy as np, ctypes as C
from powersystemsreliabilityassessment_amd import case96, api
c=case96.rts96(); E=api.Engine(c)
E.L.relmc_debug_trace.argtypes=[C.c_void_p, C.POINTER(C.c_double), C.c_int32]
for fl in ([12,31,33,56,70,78,88,89,96],[9,12,13,19,22,23,32,38,48,56,65,67,71,79,88],[22,32]):
    st=np.zeros((1,c.ncomp),np.uint8); st[0,fl]=1
    dns,nodal,info=E.mc_simulation(st,mpopt=api.mpoption(0),return_info=True)
    out=(C.c_double*(8*40))(); E.L.relmc_debug_trace(E._h,out,8*40)
    t=np.array(out).reshape(40,8)
    print(fl, dns, info)
    for it in range(int(info['iters'][0])+1): print(it, ' '.join('%10.3e'%v for v in t[it]))
