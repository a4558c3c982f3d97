#!/usr/bin/env python3
"""How sensitive is BASELINE config 2 (1 024 TwoDBicycle, 200 m x 200 m) to a perturbation of the size of an fp32
rounding?  The fp64 CPU oracle is run twice, the second time from positions moved by N(0, 4e-6 m), and the distance
between the two runs is printed every 100 ticks.  Output committed as profiles/r2_chaos_sensitivity.txt; it is why the
full-length parity test re-anchors the oracle every 100 ticks (tests/test_gpu_large.py)."""
import sys, time, numpy as np
sys.path.insert(0,'/root/repo')
import bench
from oracle import csf_oracle as orc
n, box = 1024, 200.0
reach = tuple(50.0*k for k in range(1,14))
s0, off, dq = bench.synthetic_population(n, box, reach=reach)
rng=np.random.default_rng(1)
s1=s0.copy(); s1[:,0]+=rng.normal(0,4e-6,n); s1[:,1]+=rng.normal(0,4e-6,n)
p=orc.default_params("twod")
A=orc.Population(p,s0,5.0,off,dq); B=orc.Population(p,s1,5.0,off,dq)
t0=time.time()
for k in range(20):
    A.step(100); B.step(100)
    a,b=A.state(),B.state()
    dev=np.hypot(a[:,0]-b[:,0],a[:,1]-b[:,1])
    print((k+1)*100, f"median {np.median(dev):.2e} 99% {np.percentile(dev,99):.2e} max {dev.max():.2e}  n>2cm {(dev>0.02).sum()}  {time.time()-t0:.0f}s", flush=True)
