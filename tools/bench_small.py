import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from psld_amd import ops
ops.lib()
DEV="cuda"
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/iters*1e-3
B=128
M,c=B*256,256
hn=torch.randn(M,c,device=DEV); W=torch.randn(c,c,device=DEV)*0.05; y=torch.empty(M,c,device=DEV); bias=torch.zeros(c,device=DEV)
t=timeit(lambda: ops.gemm_raw(0,0,M,c,c,hn,c,0,W,c,0,y,c,0,epi=ops.epilogue(bias=bias)))
print(f"NIN fwd NN {M}x{c}x{c}: {t*1e6:.1f} us {2*M*c*c/t/1e12:.1f} TF")
t=timeit(lambda: ops.gemm_raw(0,1,M,c,c,hn,c,0,W,c,0,y,c,0))
print(f"NIN dgrad NT {M}x{c}x{c}: {t*1e6:.1f} us {2*M*c*c/t/1e12:.1f} TF")
hw=256
q=torch.randn(B,hw,c,device=DEV); k=torch.randn(B,hw,c,device=DEV); p=torch.empty(B,hw,hw,device=DEV); v=torch.randn(B,hw,c,device=DEV); ho=torch.empty(B,hw,c,device=DEV)
t=timeit(lambda: ops.gemm_raw(0,1,hw,hw,c,q,c,hw*c,k,c,hw*c,p,hw,hw*hw,B,ops.epilogue(alpha=0.0625)))
print(f"QK^T batched NT: {t*1e6:.1f} us {2*B*hw*hw*c/t/1e12:.1f} TF")
t=timeit(lambda: ops.gemm_raw(0,0,hw,c,hw,p,hw,hw*hw,v,c,hw*c,ho,c,hw*c,B))
print(f"PV batched NN: {t*1e6:.1f} us {2*B*hw*hw*c/t/1e12:.1f} TF")
t=timeit(lambda: ops.gemm_raw(1,0,hw,c,hw,p,hw,hw*hw,v,c,hw*c,ho,c,hw*c,B))
print(f"P^T dHo batched TN: {t*1e6:.1f} us {2*B*hw*hw*c/t/1e12:.1f} TF")
M2=B*1024
x=torch.randn(M2,512,device=DEV); W2=torch.randn(256,512,device=DEV)*0.05; y2=torch.empty(M2,256,device=DEV)
t=timeit(lambda: ops.gemm_raw(0,1,M2,256,512,x,512,0,W2,512,0,y2,256,0))
print(f"1x1 512->256 @32 NT: {t*1e6:.1f} us {2*M2*256*512/t/1e12:.1f} TF")
dx=torch.empty(M2,512,device=DEV)
t=timeit(lambda: ops.gemm_raw(0,0,M2,512,256,y2,256,0,W2,512,0,dx,512,0))
print(f"1x1 dgrad NN {M2}x512x256: {t*1e6:.1f} us {2*M2*256*512/t/1e12:.1f} TF")
sl=torch.empty(16,c,c,device=DEV)
t=timeit(lambda: ops.gemm_tn_splitk(c,c,M,hn,c,y,c,sl,16))
print(f"NIN wgrad TN splitk16: {t*1e6:.1f} us {2*M*c*c/t/1e12:.1f} TF")
x4=torch.randn(M,c,device=DEV)
t=timeit(lambda: ops.silu(x4))
print(f"silu {M}x{c}: {t*1e6:.1f} us {2*M*c*4/t/1e12:.2f} TB/s")
# pointwise limb kernel
fr=ops.gemm_frag(W, c, c, 1, c)
t=timeit(lambda: ops.gemm_split(hn, None, M, fr, c, y, ops.epilogue(bias=bias)))
print(f"NIN fwd bf16x6 {M}x{c}x{c}: {t*1e6:.1f} us {2*M*c*c/t/1e12:.1f} TF")
W3=torch.randn(c,3*c,device=DEV)*0.05; fr3=ops.gemm_frag(W3, 3*c, c, 1, 3*c); y3=torch.empty(M,3*c,device=DEV)
t=timeit(lambda: ops.gemm_split(hn, None, M, fr3, 3*c, y3))
print(f"QKV fused bf16x6 {M}x{3*c}x{c}: {t*1e6:.1f} us {2*M*3*c*c/t/1e12:.1f} TF")
frd=ops.gemm_frag(W3, c, 3*c, 3*c, 1)
t=timeit(lambda: ops.gemm_split(y3, None, M, frd, c, y))
print(f"QKV dgrad bf16x6 {M}x{c}x{3*c}: {t*1e6:.1f} us {2*M*3*c*c/t/1e12:.1f} TF")
f2=ops.gemm_frag(W2, 256, 512, 512, 1)
t=timeit(lambda: ops.gemm_split(x, None, M2, f2, 256, y2))
print(f"1x1 512->256 @32 bf16x6: {t*1e6:.1f} us {2*M2*256*512/t/1e12:.1f} TF")
f2d=ops.gemm_frag(W2, 512, 256, 1, 512)
t=timeit(lambda: ops.gemm_split(y2, None, M2, f2d, 512, dx))
print(f"1x1 dgrad bf16x6 {M2}x512x256: {t*1e6:.1f} us {2*M2*256*512/t/1e12:.1f} TF")
# pointwise weight gradients: fp32 engine (conv2d_wgrad 1x1 / TN split-K) vs the limb kernel
for (mm, nn_, kk, tag) in ((c, c, M, "NIN wgrad 256x256"), (256, 512, M2, "1x1 wgrad 512->256 @32"), (256, 512, M2 // 4, "1x1 wgrad 512->256 @16")):
    A = torch.randn(kk, mm, device=DEV); Bm = torch.randn(kk, nn_, device=DEV)
    tiles = (mm // 128) * (nn_ // 128)
    kt = kk // 32
    ns = max(1, min(512 // tiles, kt // 4)); per = -(-kt // ns); ns = -(-kt // per)
    sl2 = torch.empty(ns, mm, nn_, device=DEV)
    t = timeit(lambda: ops.gemm_tn_split(mm, nn_, kk, A, mm, Bm, nn_, sl2, nn_, ns))
    ns0 = 16
    sl0 = torch.empty(ns0, mm, nn_, device=DEV)
    t0 = timeit(lambda: ops.gemm_tn_splitk(mm, nn_, kk, A, mm, Bm, nn_, sl0, ns0))
    err = ((sl2.sum(0) - sl0.sum(0)).norm() / sl0.sum(0).norm()).item()
    print(f"{tag} TN: fp32 {t0*1e6:.1f} us {2*mm*nn_*kk/t0/1e12:.1f} TF | bf16x6 (split {ns}) {t*1e6:.1f} us {2*mm*nn_*kk/t/1e12:.1f} TF (rel {err:.1e})")
# attention products: fp32 engine vs limb kernel (both operands split in the kernel)
hw_ = 256; bb = M // hw_
qq = torch.randn(bb, hw_, c, device=DEV); kk_ = torch.randn(bb, hw_, c, device=DEV); pp = torch.empty(bb, hw_, hw_, device=DEV); pp2 = torch.empty_like(pp)
for (ta, tb, tag) in ((0, 1, "QK^T NT"), (0, 0, "PV NN"), (1, 0, "P^T dHo TN")):
    t0 = timeit(lambda: ops.gemm_raw(ta, tb, hw_, hw_, c, qq, c, hw_ * c, kk_, c, hw_ * c, pp, hw_, hw_ * hw_, bb))
    t = timeit(lambda: ops.bgemm_split(ta, tb, hw_, hw_, c, qq, c, hw_ * c, kk_, c, hw_ * c, pp2, hw_, hw_ * hw_, bb))
    err = ((pp2 - pp).norm() / pp.norm()).item()
    fl = 2 * bb * hw_ * hw_ * c
    print(f"{tag} batched: fp32 {t0*1e6:.1f} us {fl/t0/1e12:.1f} TF | bf16x6 {t*1e6:.1f} us {fl/t/1e12:.1f} TF (rel {err:.1e})")
