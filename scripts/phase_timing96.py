import os, sys, ctypes as C
sys.path.insert(0,'/root/repo')
os.environ['RELMC_LIB_PATH']='/root/repo/powersystemsreliabilityassessment_amd/csrc/ablate/librelmc_pt.so'
from powersystemsreliabilityassessment_amd import api, case96
e=api.Engine(case96.rts96()); e.nsq_accumulate(1,0,65536)
acc=e.nsq_accumulate(1,1000000,1000000); ms=e.last_kernel_ms()
out=(C.c_ulonglong*8)(); e.L.relmc_debug_phase_cycles.argtypes=[C.c_void_p,C.POINTER(C.c_ulonglong)]; e.L.relmc_debug_phase_cycles(e._h,out)
names=["init","evaluate","gather/assemble","conv test","UPD","INV+BWD","update","finish"]; tot=sum(out)
print("kernel_ms",ms)
for n_,v in zip(names,out): print(f"{n_:16s} {v/tot*100:6.2f} %   {v/acc.sum_iters:10.1f} cycles per (wave-)iteration")
