"""Small driver for profiling the secondary kernels: rocprofv3 --kernel-trace --stats -- python3 scripts/aux_workloads.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powersystemsreliabilityassessment_amd import api, seq, hl1
e = api.Engine()
for k in range(4):
    e.nsq_accumulate_distinct(1, k * 1000000, 1000000)
s = seq.SeqEngine(e)
for k in range(4):
    s.seq_years(1, k * 125, 125)
g, l = hl1.rts24_generators(), hl1.rts24_load()
for k in range(4):
    hl1.run_non_sequential_mc(g, l, 100000, seed=k + 1, engine=e)
st = e.mc_sampling(None, 1000000, seed=1)
