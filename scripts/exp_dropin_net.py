import sys, time
sys.path[:0]=['.','relightable-nr_amd']
import torch, network, bench
from rnr_amd import ops
dev='cuda:0'
args=bench.parse([]); sc=bench.build_scene(args)
net=network.RenderingNet(nf0=64,in_channels=108,out_channels=78,num_down_unet=5,out_channels_gcn=512)
full=net.state_dict(); full.update({k: torch.as_tensor(v) for k,v in sc['unet_sd'].items()}); net.load_state_dict(full, strict=True)
net.to(dev).eval()
for m in net.modules():
    if type(m)==torch.nn.BatchNorm2d: m.train()
x=torch.randn(1,108,512,512,device=dev); vf=torch.zeros(1,512,device=dev)
def ev(f,n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
with torch.no_grad():
    print('render_net(x, v): %.3f ms' % ev(lambda: net(x, vf)))
    plan=net.net._plan(1,512,512,x.device)
    xin=ops.nchw_to_nhwc(x,plan.in_c_pad)
    print('plan.forward only: %.3f ms' % ev(lambda: plan.forward(xin)))
    print('fused_single', plan.fused_single, 'fused', plan.fused, 'bn_mode', plan.bn_mode)
    from rnr_amd.unet import UNetPlan
    p2=UNetPlan({k: v.to(dev) for k,v in sc['unet_sd'].items()},108,78,64,5,(512,512),1,dev)
    print('pipeline-style plan (per-view BN): %.3f ms' % ev(lambda: p2.forward(xin)))
