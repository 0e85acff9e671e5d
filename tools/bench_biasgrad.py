import torch, time, sys
sys.path.insert(0, '.')
from psld_amd import ops
DEV='cuda'
def timeit(fn, it=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/it
for (b,hw,c) in ((128,1024,256),(128,256,256),(128,64,256)):
    x=torch.randn(b,hw,c,device=DEV); out=torch.empty(c,device=DEV); pim=torch.empty(b,c,device=DEV)
    t=timeit(lambda: ops.bias_grad(x,c,b,hw,c,out,1.0,pim,0))
    ref=x.double().sum((0,1))
    print(b,hw,c,f"{t*1e6:.1f} us", ((out.double()-ref).abs().max()/ref.abs().max()).item(), (pim.double()-x.double().sum(1)).abs().max().item())
