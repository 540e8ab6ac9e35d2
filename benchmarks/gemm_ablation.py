import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from neuronika_amd import capi as c
from benchmarks.microbench import timeit, rand
dev = c.Device(0)
n=4096
A=rand(dev,(n,n),0,0,1); B=rand(dev,(n,n),1,0,1); C=dev.zeros((n,n))
f=lambda: c.sgemm(dev,0,0,n,n,n,1.0,A,n,B,n,0.0,C,n)
timeit(dev,f,60)   # long warm-up: let the clocks settle
ms=timeit(dev,f,40)
print(os.environ.get("NK_GEMM_VARIANT","0"), round(ms,4), round(2*n**3/ms/1e9,1))
