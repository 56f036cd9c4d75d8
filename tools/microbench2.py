import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from siu3r_amd import ops
from tools.microbench import timeit
def run(M,N,K,adt,odt,act=0,res=False):
    a=(torch.rand(M,K,device="cuda")*2-1).to(adt); w=torch.rand(N,K,device="cuda")*0.1
    pw=ops.pack_linear(w, torch.zeros(N,device="cuda"), False)
    r=torch.rand(M,N,device="cuda") if res else None
    t=timeit(lambda: ops.linear(a,pw,out_dtype=odt,act=act,residual=r), iters=30)
    print(f"M={M} N={N} K={K} a={str(adt)[6:]} out={str(odt)[6:]} act={act} res={res}: {t*1e6:7.1f} us {2.0*M*N*K/t/1e12:6.1f} TF/s")
for K in (64,256,1024,4096):
    run(2050,1024,K,torch.bfloat16,torch.bfloat16)
for K in (64,1024):
    run(2050,3072,K,torch.bfloat16,torch.bfloat16)
    run(2050,3072,K,torch.bfloat16,torch.float32)
run(2050,4096,1024,torch.bfloat16,torch.bfloat16,act=1)
run(2050,1024,4096,torch.bfloat16,torch.float32,res=True)
run(128,128,64,torch.bfloat16,torch.bfloat16)
run(128,128,4096,torch.bfloat16,torch.bfloat16)
run(16384,1024,1024,torch.bfloat16,torch.bfloat16)
# empty-kernel launch overhead reference
x=torch.zeros(1024,device="cuda"); 
print("torch add launch", timeit(lambda: x.add_(1), iters=100)*1e6, "us")
