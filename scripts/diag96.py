import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from powersystemsreliabilityassessment_amd import case96, api
from oracle import coracle
c=case96.rts96(); O=coracle.Oracle(c); E=api.Engine(c)
n=1000000
st=O.mc_sampling(1,0,n)
bad=[]
for lo in range(0,n,250000):
    dns,nodal,info=E.mc_simulation(st[lo:lo+250000],mpopt=api.mpoption(0),return_info=True)
    b=np.flatnonzero((info['status']==1)|(info['status']==2)); bad+=[(lo+i,info['status'][i],info['iters'][i],dns[i]) for i in b]
print('nonconverged', bad)
idx=[b[0] for b in bad]
if idx:
    r=O.mc_simulation(st[idx],0); print('oracle', r['status'], r['iters'], r['dns'])
    r=O.mc_simulation(st[idx],1); print('oracle physical', r['status'], r['iters'], r['dns'])
    for i in idx: print(i, np.flatnonzero(st[i]))
