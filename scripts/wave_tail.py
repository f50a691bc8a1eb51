import os, sys, ctypes as C
sys.path.insert(0,''+os.path.dirname(os.path.dirname(os.path.abspath(__file__)))+'')
os.environ['RELMC_LIB_PATH']=''+os.path.dirname(os.path.dirname(os.path.abspath(__file__)))+'/powersystemsreliabilityassessment_amd/csrc/ablate/librelmc_pt.so'
import numpy as np
from powersystemsreliabilityassessment_amd import api
e=api.Engine(); e.nsq_accumulate(1,0,65536)
for n in (100000, 1000000):
    acc=e.nsq_accumulate(1,1000000,n)
    buf=np.zeros(8*2048); e.L.relmc_debug_trace.argtypes=[C.c_void_p,C.POINTER(C.c_double),C.c_int32]
    e.L.relmc_debug_trace(e._h, buf.ctypes.data_as(C.POINTER(C.c_double)), buf.size)
    t=buf.view(np.uint64).reshape(2048,8).sum(1).astype(float)
    print(n, "kernel ms", e.last_kernel_ms(), "per-wave cycles mean %.3e max %.3e min %.3e  max/mean %.4f"%(t.mean(), t.max(), t.min(), t.max()/t.mean()))
t=t.reshape(512,4)           # [block][wave in block]
print("by wave-in-block:", np.round(t.mean(0)/t.mean(),4))
bm=t.mean(1)
print("by XCD (block % 8):", np.round(np.array([bm[x::8].mean() for x in range(8)])/bm.mean(),4))
print("block-level max/mean %.4f  std/mean %.4f; within-block spread (max-min)/mean avg %.4f"%(bm.max()/bm.mean(), bm.std()/bm.mean(), ((t.max(1)-t.min(1))/t.mean(1)).mean()))
srt=np.argsort(bm); print("slowest blocks", srt[-8:], np.round(bm[srt[-8:]]/bm.mean(),3), "fastest", srt[:8], np.round(bm[srt[:8]]/bm.mean(),3))
# iteration-count imbalance alone would give:
