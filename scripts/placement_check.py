import sys, os, time; sys.path.insert(0,'/root/repo')
os.environ['RELMC_VERBOSE']='1'
import ctypes as C
from powersystemsreliabilityassessment_amd import api, case96
for name,case in (("rts24",None),("rts96",case96.rts96())):
    t=time.time(); e=api.Engine(case) if case is not None else api.Engine(); dt=time.time()-t
    o=(C.c_int32*9)(); e.L.relmc_debug_schedule(e._h,o)
    e.nsq_accumulate(1,0,200000); ts=[]
    for k in range(3):
        acc=e.nsq_accumulate(1,1000000*(k+1),1000000); ts.append(e.last_kernel_ms())
    print(name,"load %.2fs"%dt,"kernel ms %.3f"%min(ts),"iters",acc.sum_iters,"sum_dns %.9f"%acc.sum_dns,"nfail",acc.n_fail)
