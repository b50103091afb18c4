import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time
from multimodal_plankton_recognition_amd import ops
dev='cuda'
for (M,N,K,ta,tb) in [(512,512,512,False,True),(512,512,512,False,False),(512,512,512,True,False),(512,512,514,False,True),(512,50,512,False,True)]:
    a=torch.randn((K,M) if ta else (M,K),device=dev); b=torch.randn((N,K) if tb else (K,N),device=dev)
    for _ in range(5): ops.gemm(a,b,trans_a=ta,trans_b=tb)
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(200): ops.gemm(a,b,trans_a=ta,trans_b=tb)
    e.record(); torch.cuda.synchronize()
    print(M,N,K,ta,tb,'%.1f us'%(s.elapsed_time(e)*1000/200))
