import sys; sys.path.insert(0,'/root/repo')
import torch
from ccd_amd import ops
dev=torch.device('cuda:0')
for rows,N in ((131072,1152),(131072,384),(2097152,128),(131072,64)):
    x=torch.randn(rows,N,device=dev).to(torch.bfloat16); out=torch.zeros(N,device=dev)
    for _ in range(3): ops.colsum_bf16(x,out)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.colsum_bf16(x,out)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/20
    print(rows,N,round(ms*1000,1),"us",round(rows*N*2/ms/1e6,1),"GB/s")
