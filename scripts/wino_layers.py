"""Layer-by-layer difference between UNetPlan(conv_algo='winograd') and conv_algo='direct' on the shapes of
tests/test_gpu_unet.py::test_unet_plan_winograd_vs_direct (debugging aid): raw outputs, BatchNorm scale / shift per step."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'relightable-nr_amd'))
import torch
from rnr_amd.unet import UNetPlan
from rnr_amd.scene import unet_state_dict
V = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
sd = unet_state_dict(30, 78, 64, 5, seed=3)
dev = torch.device('cuda:0')
wino = UNetPlan(sd, 30, 78, 64, 5, (S, S), V, dev, conv_algo='winograd')
direct = UNetPlan(sd, 30, 78, 64, 5, (S, S), V, dev, conv_algo='direct')
x = torch.randn(V, S, S, wino.in_c_pad, generator=torch.Generator().manual_seed(1)).to(dev)
x[..., 30:] = 0
wino.forward(x); direct.forward(x)
torch.cuda.synchronize()
for i, (a, b) in enumerate(zip(wino.steps, direct.steps)):
    xa, xb = a['out'].data[:V], b['out'].data[:V]
    d = (xa - xb).abs()
    algo = wino.L.rnr_conv_algorithm(ctypes.byref(a['desc']), V, *a['in_hw'])
    line = 'L%2d kind %d algo %d in %3dx%-3d c %4d -> %4d  |out| %.3e  diff %.3e  rel %.2e' % (
        i + 1, a['desc'].kind, algo, a['in_hw'][0], a['in_hw'][1], a['desc'].c_in0 + a['desc'].c_in1, a['desc'].c_out,
        xb.abs().max().item(), d.max().item(), (d.max() / xb.abs().max()).item())
    if a['out'].scale is not None:
        line += '   scale diff %.2e shift diff %.2e' % ((a['out'].scale[:V] - b['out'].scale[:V]).abs().max().item(),
                                                         (a['out'].shift[:V] - b['out'].shift[:V]).abs().max().item())
    print(line)
# where does the first differing layer differ?
for i, (a, b) in enumerate(zip(wino.steps, direct.steps)):
    xa, xb = a['out'].data[:V], b['out'].data[:V]
    bad = ((xa - xb).abs() > 1e-3 * xb.abs().max()).nonzero()
    if len(bad):
        print('first differing layer L%d: %d of %d elements differ' % (i + 1, len(bad), xa.numel()))
        print('views', bad[:, 0].unique().tolist()[:16])
        print('y mod 8', (bad[:, 1] % 8).unique().tolist(), ' x mod 16', (bad[:, 2] % 16).unique().tolist(), ' c', bad[:, 3].unique().tolist()[:70])
        print('y', bad[:, 1].unique().tolist()[:40]); print('x', bad[:, 2].unique().tolist()[:40])
        print(bad[:10].tolist())
        for t in bad[:16].tolist():
            print(t, 'wino %.6f direct %.6f' % (xa[tuple(t)].item(), xb[tuple(t)].item()))
        # again: is it reproducible?
        wino.forward(x); torch.cuda.synchronize()
        xa2 = wino.steps[i]['out'].data[:V]
        bad2 = ((xa2 - xb).abs() > 1e-3 * xb.abs().max()).nonzero()
        print('second run: %d differ' % len(bad2), bad2[:4].tolist())
        break
