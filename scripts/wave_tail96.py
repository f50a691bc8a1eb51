import os, sys, ctypes as C
sys.path.insert(0,'/root/repo')
os.environ['RELMC_LIB_PATH']='/root/repo/powersystemsreliabilityassessment_amd/csrc/ablate/librelmc_pt.so'
import numpy as np
from powersystemsreliabilityassessment_amd import api, case96
e=api.Engine(case96.rts96()); e.nsq_accumulate(1,0,65536)
acc=e.nsq_accumulate(1,1000000,1000000)
buf=np.zeros(8*2048); e.L.relmc_debug_trace.argtypes=[C.c_void_p,C.POINTER(C.c_double),C.c_int32]
e.L.relmc_debug_trace(e._h, buf.ctypes.data_as(C.POINTER(C.c_double)), buf.size)
t=buf.view(np.uint64).reshape(256,8,8).sum(2).astype(float)
print("rts96 kernel ms", e.last_kernel_ms(), "max/mean %.4f"%(t.max()/t.mean()), "by wave-in-block", np.round(t.mean(0)/t.mean(),3))
