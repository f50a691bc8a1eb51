"""How many wavefront iterations do different row-grouping policies cost?  (model of relmc_eval_kernel's trip counts)"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from powersystemsreliabilityassessment_amd import api
e=api.Engine(); c=e.case; n=1_000_000
st=e.mc_sampling(None,n,seed=1,first_index=0)
it=np.zeros(n,dtype=np.int32)
for lo in range(0,n,250000):
    _,_,info=e.mc_simulation(st[lo:lo+250000],return_info=True); it[lo:lo+250000]=info["iters"]
cap=c.inj_pmax[:c.ng].sum()-st[:,:c.ng].astype(np.float64)@c.inj_pmax[:c.ng]
deficit=np.maximum(0.0,c.total_load-cap); hard=deficit>0
ideal=it.sum()/4
def cost(order):   # order: permutation; groups of 4 consecutive
    return it[order].reshape(-1,4).max(1).sum()
base=cost(np.arange(n))
def windowed(W, key):
    order=[]
    for lo in range(0,n,W):
        idx=np.arange(lo,min(lo+W,n)); order.append(idx[np.argsort(key[idx],kind="stable")])
    return cost(np.concatenate(order))
print("ideal (sum/4) %.0f ; i.i.d. rows %.4f ; mean iters %.3f"%(ideal, base/ideal, it.mean()))
for W in (64,128,256,1024):
    print("window %4d: hard-last %.4f ; by deficit %.4f ; by true iterations (oracle key) %.4f"%(W, windowed(W,hard.astype(int))/ideal, windowed(W,deficit)/ideal, windowed(W,it)/ideal))
nf=st[:,:c.ng].sum(1)
print("window 64 by (hard, #units out):", windowed(64, hard*100+nf)/ideal)
