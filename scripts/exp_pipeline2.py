"""Two steps in flight: two RNRPipeline instances on two HIP streams, steps submitted alternately, vs one pipeline on one
stream.  Frames/s over 8-view steps."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'relightable-nr_amd'))
import torch
import bench
from rnr_amd import scene
from rnr_amd.pipeline import RNRPipeline

class A: pass
args = A(); args.img_size = 512; args.nf0 = 64; args.tex_ch = 24
sc = bench.build_scene(args)
dev = torch.device('cuda:0')
V = 8
prec = sys.argv[1] if len(sys.argv) > 1 else 'f32'
mk = lambda: RNRPipeline(sc['mesh'], 512, sc['textures'], sc['unet_sd'], sc['pivots_spec'], sc['pivots_diff'], None, nf0=64,
                         max_views=V, device=dev, sh_coeff=sc['sh_coeff'], sh_lmax=10, skip_background_tiles=False, precision=prec)
pipes = [mk(), mk()]
steps = 24
poses = []
for s in range(steps):
    ids = [(s * V + i) * 7 % 720 for i in range(V)]
    poses.append({k: torch.from_numpy(x).to(dev) for k, x in scene.spiral_views(512, ids).items()})
call = lambda p, d: p.render(d['proj'], d['pose'], d['proj_inv'], d['R_inv'])
for s in range(4):
    call(pipes[s & 1], poses[s])
torch.cuda.synchronize()
t0 = time.perf_counter()
for s in range(steps):
    call(pipes[0], poses[s])
torch.cuda.synchronize()
t1 = (time.perf_counter() - t0) / steps
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for st in streams:
    st.wait_stream(torch.cuda.current_stream())
def run2():
    for s in range(steps):
        with torch.cuda.stream(streams[s & 1]):
            call(pipes[s & 1], poses[s])
run2(); torch.cuda.synchronize()
t0 = time.perf_counter()
run2()
torch.cuda.synchronize()
t2 = (time.perf_counter() - t0) / steps
print('%s  one in flight %.3f ms/step (%.1f frames/s)   two in flight %.3f ms/step (%.1f frames/s)  %+.1f %%' % (
    prec, t1 * 1e3, V / t1, t2 * 1e3, V / t2, (t1 / t2 - 1) * 100))
