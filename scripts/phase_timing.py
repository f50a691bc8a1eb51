import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powersystemsreliabilityassessment_amd import api
eng = api.Engine()
eng.nsq_accumulate(1, 0, 65536)
acc = eng.nsq_accumulate(1, 1000000, 1000000)
ms = eng.last_kernel_ms()
out = (C.c_ulonglong * 8)()
eng.L.relmc_debug_phase_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
eng.L.relmc_debug_phase_cycles(eng._h, out)
names = ["init", "evaluate", "gather/assemble", "conv test", "UPD", "INV+BWD", "update", "finish"]
tot = sum(out)
print("kernel_ms", ms)
for n, v in zip(names, out):
    print(f"{n:16s} {v/tot*100:6.2f} %   {v/acc.sum_iters*4:10.1f} cycles per wave-iteration")
o9 = (C.c_int32 * 9)()
eng.L.relmc_debug_schedule.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
eng.L.relmc_debug_schedule(eng._h, o9)
print("schedule: upd/inv/bwd passes", o9[0], o9[1], o9[2], "noff", o9[3], "nzero", o9[4], "nws", o9[5], "lds_bytes", o9[6], "blocks/CU", o9[7], "tasks", o9[8])
