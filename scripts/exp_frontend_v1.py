"""One view per call: what the launches in front of the U-Net cost and why (frame_prepare / setup_splat_faces / raster_tile /
shade_inputs).  Prints the rasterizer's list statistics of a few spiral views (wide list length, binned candidates per tile,
box sizes) and HIP-event times of the front end alone, looped; run under rocprofv3 --kernel-trace --stats for per-kernel times.
usage: python scripts/exp_frontend_v1.py [reps]    (RNR_HIP_LIB selects a library variant)"""
import os
import sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'relightable-nr_amd'))
import bench  # noqa: E402
from rnr_amd import ops, scene  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
args = bench.parse(['--no-cpu-baseline'])
dev = torch.device('cuda:0')
sc = bench.build_scene(args)
pipe = bench.make_pipeline(sc, args, dev, 1)
S, nf = args.img_size, pipe.mesh.num_faces
ids = np.arange(720)
pv = {k: torch.from_numpy(v).to(dev) for k, v in scene.spiral_views(S, ids).items()}


def pose(i):
    return tuple(pv[k][i:i + 1].contiguous() for k in ('proj', 'pose', 'proj_inv', 'R_inv'))


def align(x, a=256):
    return (x + a - 1) // a * a


ntiles = ((S + 15) // 16) ** 2
off_boxes = 0
off_counts = align(nf * 8) + 2 * align(nf * 36)
if os.environ.get('STATS', '1') == '1':
    for vid in [int(x) for x in os.environ.get('VIEWS', '0,111,300,552').split(',')]:
        pipe.render(*pose(vid))
        torch.cuda.synchronize()
        ws = pipe._lane_ws[0]
        counts = ws[off_counts:off_counts + 4 * (ntiles + 1)].view(torch.int32).cpu().numpy()
        boxes = ws[:nf * 8].view(torch.int16).reshape(nf, 4).cpu().numpy().astype(np.int64)
        tc, wide = counts[:ntiles], counts[ntiles]
        live = boxes[:, 0] <= boxes[:, 1]
        exact = boxes[:, 0] == -2
        area = (boxes[:, 1] - np.maximum(boxes[:, 0], 0) + 1) * (boxes[:, 3] - np.maximum(boxes[:, 2], 0) + 1)
        a = area[live & ~exact]
        wsz = ws.numel()
        rec_bytes = align(nf * 8 * 4)
        rec = ws[wsz - rec_bytes:wsz - rec_bytes + 32 * int(wide)].view(torch.float32).reshape(-1, 8).cpu().numpy()
        T = S // 16
        cx = lambda i: ((2 * i + 1 - S).astype(np.float32) / np.float32(S))
        lo, hi = cx(np.arange(T) * 16), cx(np.arange(T) * 16 + 15)

        def rej(xa, ya, xb, yb):
            dx, dy = (xb - xa)[:, None, None], (yb - ya)[:, None, None]
            p0, p1 = (lo[None, :, None] - ya[:, None, None]) * dx, (hi[None, :, None] - ya[:, None, None]) * dx
            q0, q1 = (lo[None, None, :] - xa[:, None, None]) * dy, (hi[None, None, :] - xa[:, None, None]) * dy
            return (p0 < q0) & (p0 < q1) & (p1 < q0) & (p1 < q1)
        x0, y0, x1, y1, x2, y2 = (rec[:, k] for k in range(6))
        touch = ~(rej(x0, y0, x1, y1) | rej(x1, y1, x2, y2) | rej(x2, y2, x0, y0))
        per_tile = touch.sum(0).ravel()
        nl = ((x0 == x1) & (y0 == y1)).astype(int) + ((x1 == x2) & (y1 == y2)) + ((x2 == x0) & (y2 == y0))
        print('   wide faces with exactly two coincident vertices %d, three %d, none %d; generic candidates per tile max %d' % (
            int((nl == 1).sum()), int((nl == 3).sum()), int((nl == 0).sum()), int(touch[nl != 1].sum(0).max()) if (nl != 1).any() else 0))
        print('   wide candidates per tile: mean %.1f median %d p90 %d p99 %d max %d; pairs %d' % (
            per_tile.mean(), int(np.median(per_tile)), int(np.percentile(per_tile, 90)), int(np.percentile(per_tile, 99)),
            int(per_tile.max()), int(per_tile.sum())))
        print('view %d: wide list %d, BOX_EXACT %d, live faces %d, tiles with binned candidates %d (max %d, sum %d); box pixels: '
              'mean %.1f median %d p90 %d p99 %d max %d, > 256: %d, 65..256: %d, sum %d'
              % (vid, wide, int(exact.sum()), int(live.sum()), int((tc > 0).sum()), int(tc.max()), int(tc.sum()),
                 a.mean(), int(np.median(a)), int(np.percentile(a, 90)), int(np.percentile(a, 99)), int(a.max()),
                 int((a > 256).sum()), int(((a > 64) & (a <= 256)).sum()), int(a[a <= 256].sum())), flush=True)

# the front end alone, looped (what the rocprofv3 kernel statistics of this run then show)
pb = pipe._prep(None)
gb = {m: pipe._gb[m][:1] for m in pipe._gb_maps}
ws = pipe._lane_ws[0]


def front(i, shade=True):
    proj, ps, proj_inv, R_inv = pose(i % 720)
    ops.frame_prepare(pipe.mesh, proj, ps, S, v_uvz=pb['v_uvz'][:1], tangents=pb['tangents'], lp_basis=pipe.sh_lighting.basis_recon,
                      lp_coeff=pipe.sh_coeff[0], light_probe=pb['lp'], workspace=ws)
    ops.rasterize_gbuffer(pipe.mesh, pb['v_uvz'][:1], None, S, pipe.near, pipe.far, maps=pipe._gb_maps, out=gb, workspace=ws, prepared=True)
    if shade:
        ops.shade_inputs(gb, pipe.mesh, proj_inv, R_inv, pipe.textures, pipe.pivots_spec, pipe.pivots_diff, pipe.sh_start_ch,
                         c_pad=pipe.unet.in_c_pad, net_in=pipe._net_in[:1], tangents=pb['tangents'])


with ops.on_device(dev):
    for shade in (True, False):
        for i in range(20):
            front(i, shade)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps):
            front(i, shade)
        e1.record()
        torch.cuda.synchronize()
        print('front end %s shade_inputs: %.1f us per view (%d views, lib %s)'
              % ('with' if shade else 'without', e0.elapsed_time(e1) * 1e3 / reps, reps, os.environ.get('RNR_HIP_LIB', 'in-tree')), flush=True)
