import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/relightable-nr_amd'); sys.path.insert(0,'/root/repo/tests')
import torch
import test_gpu_unet as tu
from rnr_amd import _lib
cases=[(0,1,512,512,[64,64],78),(0,1,512,512,[64],64),(0,1,256,256,[128,128],128),(2,1,256,256,[64,64],64),(2,1,128,128,[128],128),(1,1,512,512,[64],128),(2,1,256,256,[64],78)]
tot=0
for (kind,N,H,W,cins,c_out) in cases:
    g = torch.Generator().manual_seed(1)
    srcs=[]
    for j,C in enumerate(cins):
        raw=torch.randn(N,C,H,W,generator=g); srcs.append((raw,None,torch.randn(N,C,generator=g)*0.3,1))
    cin=sum(cins); k=4 if kind else 3
    w=(torch.randn(cin,c_out,4,4,generator=g) if kind==2 else torch.randn(c_out,cin,k,k,generator=g))/(cin*k*k)**0.5
    nat,_=tu.run_conv(kind,srcs,w,c_out,N,H,W)
    bad_runs=0; worst=0; first=None; unstable=0
    for rep in range(int(sys.argv[1]) if len(sys.argv)>1 else 20):
        emu,_=tu.run_conv(kind,srcs,w,c_out,N,H,W,flags=_lib.CONV_F32_EMU_BF16X6)
        e=(emu-nat).abs().max().item(); worst=max(worst,e)
        if first is None: first=emu
        unstable += not torch.equal(emu, first)
        bad_runs+= e>1e-3
    tot+=bad_runs
    print((kind,N,H,W,cins,c_out),'bad runs',bad_runs,'bitwise-unstable runs',unstable,'worst diff vs f32 %.2e'%worst, flush=True)
print('TOTAL BAD', tot)
