import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np
from powersystemsreliabilityassessment_amd import api
e=api.Engine(); n=1000000
st=e.mc_sampling(None,n,seed=1)
for rep in range(3):
    t=time.time(); dns,nodal=e.mc_simulation(st); dt=time.time()-t
    print("mc_simulation host buffers: n=%d wall %.1f ms -> %.2f M states/s (kernel %.1f ms)"%(n,dt*1e3,n/dt/1e6,e.last_kernel_ms()))
t=time.time(); st=e.mc_sampling(None,n,seed=2); print("mc_sampling to host %.1f ms"%((time.time()-t)*1e3))
