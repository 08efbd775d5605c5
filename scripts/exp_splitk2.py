"""EXPERIMENT: split-K factor of the batched weight-gradient GEMMs (ASR_GEMM_SPLITK), stand-alone."""
import sys, os; sys.path.insert(0,".")
import torch
from e2e_asr_amd import ops
dev=torch.device("cuda:0")
def timed(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
for (M,N,K,batch) in ((80,1024,25600,2),(256,1024,25600,2),(1024,1024,12800,2),(256,1024,12800,2),(1024,1024,6400,2),(256,1024,6400,2)):
    a=torch.randn(batch,K,M,device=dev); b=torch.randn(K,batch*N,device=dev); c=torch.zeros(batch,M,N,device=dev)
    if batch==1:
        ms=timed(lambda: ops.gemm(a[0],b,None,True,False,out=c[0],accumulate=True))
    else:
        # as lstm_bwd.hip launches the two directions: A_d = a[d] [K,M] (stride K*M), B_d = columns d*N.. of b [K, 2N] (stride N)
        ms=timed(lambda: ops.gemm_batched(a, b, c, M, N, K, M, batch*N, N, K*M, N, M*N, batch, True, False, accumulate=True))
    print(os.environ.get("ASR_GEMM_SPLITK","default"),"TN",M,N,K,"x%d"%batch,"%.1f us"%(ms*1e3))
