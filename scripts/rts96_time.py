import sys; sys.path.insert(0,'/root/repo')
import ctypes as C
from powersystemsreliabilityassessment_amd import api, case96
e=api.Engine(case96.rts96()); e.nsq_accumulate(1,0,200000)
ts=[]
for k in range(3):
    acc=e.nsq_accumulate(1,1000000*(k+1),1000000); ts.append(e.last_kernel_ms())
o=(C.c_int32*9)(); e.L.relmc_debug_schedule(e._h,o)
print("rts96 ms", min(ts), "iters", acc.sum_iters, "sum_dns %.6f"%acc.sum_dns, "nc", acc.n_nonconverged, "lds", o[6], "blocks/cu", o[7])
