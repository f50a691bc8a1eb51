import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys; sys.path.insert(0, %r)
from powersystemsreliabilityassessment_amd import api
eng = api.Engine()
eng.nsq_accumulate(1, 0, 200000)
ts=[]
for k in range(3):
    eng.nsq_accumulate(1, 1000000*(k+1), 1000000); ts.append(eng.last_kernel_ms())
print(min(ts))
''' % ROOT
for v in sys.argv[1:]:
    env = dict(os.environ, RELMC_LIB_PATH=os.path.join(ROOT, 'powersystemsreliabilityassessment_amd/csrc/ablate', v + '.so'))
    out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True)
    print(v, out.stdout.strip(), out.stderr.strip()[-300:])
