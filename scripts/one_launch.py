import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powersystemsreliabilityassessment_amd import api
eng = api.Engine()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
eng.nsq_accumulate(1, 0, 65536)
acc = eng.nsq_accumulate(1, 1000000, n)
print("kernel_ms", eng.last_kernel_ms(), "scen/s", n / eng.last_kernel_ms() * 1e3, "iters", acc.sum_iters / acc.n)
