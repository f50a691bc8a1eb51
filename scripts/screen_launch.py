"""A few screened launches (relmc_solver_opts.screen = 1) for the kernel trace of scripts/screen_profile.sh:
    python scripts/screen_launch.py [rts24|rts96|seq] [samples / years]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powersystemsreliabilityassessment_amd import api, case24, case96, seq
what = sys.argv[1] if len(sys.argv) > 1 else "rts24"
n = int(sys.argv[2]) if len(sys.argv) > 2 else (125 if what == "seq" else 1000000)
eng = api.Engine(case96.rts96() if what == "rts96" else case24.rts24())
so = api.mpoption(screen=1)
if what == "seq":
    se = seq.SeqEngine(eng)
    for k in range(4):
        acc = se.seq_years(1, k * n, n, so)[4]
else:
    for k in range(4):
        acc = eng.nsq_accumulate(1, k * n, n, so)
print(what, "units", acc.n, "screened", acc.n_screened, "device ms of the last call", eng.last_kernel_ms())
