import sys, time
sys.path[:0]=['.','relightable-nr_amd']
import torch
from rnr_amd import ops
dev='cuda:0'
x=torch.randn(1,108,512,512,device=dev)
def t(f,n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
y=ops.nchw_to_nhwc(x,112)
print('nchw_to_nhwc 108->112 @512^2: %.3f ms'%t(lambda: ops.nchw_to_nhwc(x,112)))
raw=torch.randn(1,512,512,80,device=dev); b=torch.randn(80,device=dev)
print('nhwc_to_nchw 80->78 tanh: %.3f ms'%t(lambda: ops.nhwc_to_nchw(raw,78,bias=b,apply_tanh=True)))
print('torch permute+contig ref: %.3f ms'%t(lambda: x.permute(0,2,3,1).contiguous()))
a=torch.randn(1,512,512,3,13,device=dev); c=torch.randn(1,512,512,3,13,device=dev)
print('cat rays: %.3f ms'%t(lambda: torch.cat((a,c),-1)))
uv=torch.rand(1,512,512,2,26,device=dev); lt=torch.rand(1,26,3,512,512,device=dev); lp=torch.rand(1,100,200,3,device=dev); al=torch.rand(1,3,512,512,device=dev)
print('ops.ray_renderer (API layout, 26 rays, 512^2): %.3f ms'%t(lambda: ops.ray_renderer(uv, lt, lp, al, al, 13, False, True, 1.0)))
