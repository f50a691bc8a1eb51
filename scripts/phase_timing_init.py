"""Setup shares of one scenario group (-DRELMC_PHASE_TIMING -DRELMC_PT_INIT build passed with RELMC_LIB_PATH)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powersystemsreliabilityassessment_amd import api
eng = api.Engine()
eng.nsq_accumulate(1, 0, 65536)
acc = eng.nsq_accumulate(1, 1000000, 1000000)
out = (C.c_ulonglong * 8)()
eng.L.relmc_debug_phase_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
eng.L.relmc_debug_phase_cycles(eng._h, out)
names = ["window sampling", "state from window", "status -> model", "topology", "susceptance entries", "start point", "interior-point loop", "output"]
tot = sum(out)
print("kernel_ms", eng.last_kernel_ms())
for n, v in zip(names, out):
    print(f"{n:22s} {v/tot*100:6.2f} %   {v/(acc.n/4):10.1f} cycles per scenario group")
