import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/relightable-nr_amd'); sys.path.insert(0,'/root/repo/tests')
import torch
import test_gpu_unet as tu
cases=[(0,1,512,512,[64,64],78),(0,1,512,512,[64],64),(0,1,256,256,[128,128],128),(2,1,256,256,[64,64],64),(2,1,128,128,[128],128),(1,1,512,512,[64],128),(1,1,256,256,[128],256)]
tot=0
for (kind,N,H,W,cins,c_out) in cases:
    g = torch.Generator().manual_seed(1)
    srcs=[(torch.randn(N,C,H,W,generator=g),None,torch.randn(N,C,generator=g)*0.3,1) for C in cins]
    cin=sum(cins); k=4 if kind else 3
    w=(torch.randn(cin,c_out,4,4,generator=g) if kind==2 else torch.randn(c_out,cin,k,k,generator=g))/(cin*k*k)**0.5
    ref,_=tu.run_conv(kind,srcs,w,c_out,N,H,W)
    bad=0
    for rep in range(20):
        out,_=tu.run_conv(kind,srcs,w,c_out,N,H,W)
        bad += not torch.equal(out, ref)
    tot+=bad
    print((kind,N,H,W,cins,c_out),'runs differing from the first:',bad, flush=True)
print('TOTAL', tot)
