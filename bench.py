#!/usr/bin/env python
"""bench.py — rendered frames/s of the RNR deferred-render hot path on MI355X (BASELINE.json metric).

A "step" = one pass of the full HIP hot path (projection -> rasterize -> shade inputs -> U-Net -> ray render) over
one batch of `--views-per-step` synthetic 512x512 camera poses of the material_sphere-like scene,
inputs resident in HBM.  `value` = views rendered by ALL ranks / max-over-ranks wall time of exactly K steps.
Beside it, `single_view_mode`: the reference's own calling mode (test_rnr.py:265-393, one view per call, the 720
`spiral_step720` views in order) with its own roofline block.

Single GPU:  python bench.py [--steps K --warmup W]
N GPUs:      python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
             bench.py --gpus N --steps K --warmup W       (weak scaling: every rank renders its own pose slice and the
             frames are all-gathered over RCCL each step — the only collective of the path)
         or  python bench.py --gpus N ...                 (no launcher: this file starts the N ranks itself, launch_ranks)
Either way the ranks that took part are counted through the process group (`n_ranks_seen`) and a count other than --gpus is
an error exit, never a bench line.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'relightable-nr_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak (f32 in / f32 acc)
PEAK_HBM_GBS = 8000.0
EMU_PEAK = {'f32': PEAK_F32_MFMA_TFLOPS, 'bf16x6': 2500.0 / 6, 'f16x3': 2500.0 / 3}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--views-per-step', type=int, default=16,
                    help='camera poses per GPU per step (r04: 16; 8 until then — 2, 4, 8, 16, 32 views per step measure 483, 527, 547, 557, 561 frames/s)')
    ap.add_argument('--img-size', type=int, default=512)
    ap.add_argument('--windows', type=int, default=3,
                    help='timed windows of exactly --steps steps each (barrier + synchronize around every one); `value` is the '
                         'median window, all of them are printed')
    ap.add_argument('--prewarm-seconds', type=float, default=2.0,
                    help='untimed steps rendered for this long in front of the --warmup steps (a fresh box starts cold: clocks, '
                         'allocator, code objects); 0 = none')
    ap.add_argument('--nf0', type=int, default=64)
    ap.add_argument('--tex-ch', type=int, default=24)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-parity', action='store_true')
    ap.add_argument('--tile-skip', action='store_true',
                    help='let the out layer skip all-background pixel tiles in the timed loop (frames are bit-identical; '
                         'off by default so that `value` is the full RenderingNet on every pixel; the skip-enabled rate '
                         'is reported beside it as with_background_tile_skip)')
    ap.add_argument('--precision', default='f32', choices=['f32', 'bf16x6', 'f16x3'],
                    help="conv arithmetic of the timed loop: exact fp32 MFMA (default, the headline) or fp32 emulated on the bf16 "
                         "matrix cores (RNR_CONV_F32_EMU_BF16X6)")
    ap.add_argument('--conv-algo', default=None, choices=['winograd', 'winograd4', 'direct'],
                    help='U-Net convolution algorithm (rnr_amd.unet.UNetPlan conv_algo; default: the library default, winograd4 — fp32 Winograd '
                         'F(4x4,3x3) on the 3x3 layers whose grid fills the chip, F(2x2,3x3) / F(2x2,2x2) elsewhere, all on the f32 MFMA; '
                         'winograd = F(2x2, .) only (the r03 path); direct = every layer as a direct implicit GEMM)')
    ap.add_argument('--pmc-file', default=None,
                    help='merged.json written by scripts/pmc.sh (rocprofv3 --pmc passes over THIS command line at the same '
                         '--views-per-step / --precision): source of roofline.traffic.  Without it the newest committed '
                         'profiles/r*_pmc_per_kernel_*.json recorded at the same batch size and precision is used and the '
                         'field is labelled traffic_from_committed_profile')
    ap.add_argument('--check-gather', action='store_true',
                    help='after the timed loop verify on every rank that its slot of the gathered frame buffer equals the '
                         'frames it rendered (bitwise) and report it as gather_check (tests/test_gpu_dist.py)')
    ap.add_argument('--main-loop-only', action='store_true',
                    help='skip the per-stage and single-view extras after the timed loop (rocprofv3 runs: every kernel launch in the profile then belongs to a timed or warm-up step)')
    ap.add_argument('--no-calibration', action='store_true',
                    help='skip box_calibration (rocprofv3 runs: its register-resident MFMA loop would be half of the kernel statistics)')
    ap.add_argument('--single-views', type=int, default=720,
                    help='views of the single_view_mode block (default: the whole spiral_step720 trajectory, as test_rnr.py renders it)')
    ap.add_argument('--no-dropin-loop', action='store_true', help='skip the dropin_view_loop block (INTEGRATION.md Level 1 timing)')
    ap.add_argument('--stub-pipeline', action='store_true',
                    help='TEST HOOK (tests/test_bench_flow_cpu.py): run the control flow of this file — pose slicing, overlapped '
                         'frame gather, barrier + MAX-reduced timing, rank-0-only JSON, --check-gather — on CPU tensors with the '
                         'gloo backend and a stand-in pipeline whose frames encode their pose.  Never a measurement.')
    return ap.parse_args(argv)


def build_scene(args):
    from rnr_amd import scene
    from rnr_amd.rays import ray_pivots
    ps, pd = ray_pivots(6, 2, 5), ray_pivots(6, 2, 10)       # train_rnr.py:344-354 defaults
    n_rays = ps.shape[1] + pd.shape[1]
    c_in = 3 * n_rays + 6 + args.tex_ch
    return {
        'mesh': scene.uv_sphere(128, 256),                                    # 65 536 faces
        'textures': scene.synthetic_textures(512, args.tex_ch, 4, 0),
        'unet_sd': scene.unet_state_dict(c_in, 3 * n_rays, args.nf0, 5, 0),
        'pivots_spec': ps, 'pivots_diff': pd,
        'sh_coeff': torch.from_numpy(scene.synthetic_sh_coeff(2, 10, 1)),        # LightingSH coeff [2,121,3], lmax 10
        'c_in': c_in, 'n_rays': n_rays,
    }


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(sc, args, view_id, hip_image, emu_image=None, budget_s=150.0):
    """The oracle (a port of the reference's algorithm) timed on this box's host cores on a bounded sample of the same
    workload, BASELINE.md §3 protocol: 3 warm-up + 10 timed frames, median, one view per call like test_rnr.py:265; pose
    tensors and the view-independent lmax-10 lighting basis are prepared OUTSIDE the timer (the reference builds that
    basis once in LightingSH.__init__, network.py:574-582).  Threads: the protocol says every host core; torch-CPU's small
    ops oversubscribe badly on a 256-thread host (116 s per frame, measured once), so a quarter-resolution probe frame is
    rendered at {64, 32, 16, 8} threads first and the fastest setting is timed, the others are listed.  Also yields the parity figure (PSNR of the HIP
    frame vs the oracle frame of the same pose)."""
    from oracle import rnr_oracle as orc
    from oracle import raster as oras
    from rnr_amd import scene
    oras.build()
    ncpu = os.cpu_count() or 1
    try:
        omp = ctypes.CDLL('libgomp.so.1')
    except OSError:
        omp = None

    def set_threads(n):
        torch.set_num_threads(n)
        if omp is not None:
            omp.omp_set_num_threads(n)
    mesh_t = {k: torch.as_tensor(v) for k, v in sc['mesh'].items()}
    basis = torch.from_numpy(orc.sh_basis(10, orc.lp_recon_dirs().numpy()).astype(np.float32))      # init-time in the reference
    n_timed = 10
    vids = [(view_id + 97 * (k + 1)) % 720 for k in range(n_timed + 3)] + [view_id]   # the LAST one is the frame the HIP path rendered last
    views_all = [{k: torch.from_numpy(v) for k, v in scene.spiral_views(args.img_size, [vid]).items()} for vid in vids]

    def frame(views):
        lp = orc.reconstruct_lp(sc['sh_coeff'][0], basis)[None]                   # network.py:622-627, per view
        return orc.render_frame(mesh_t, views, args.img_size, sc['textures'], sc['unet_sd'], lp, sc['pivots_spec'],
                                sc['pivots_diff'])
    t_start = time.time()
    # thread sweep on a quarter-resolution probe frame (same network, 1/16 of the pixels): a full frame on all 256 logical
    # CPUs of the r03 box took 116 s (4.8 s on 32 threads) — oversubscription of torch-CPU's small ops, not a baseline
    probe = {k: torch.from_numpy(v) for k, v in scene.spiral_views(args.img_size // 4, [view_id]).items()}
    sweep = {}
    # (all 256 logical CPUs: 81 s for the PROBE frame, 116 s for a full one, measured once in r03 and not repeated on every run)
    for n in sorted({min(ncpu, 64), min(ncpu, 32), min(ncpu, 16), min(ncpu, 8)}, reverse=True):
        set_threads(n)
        t0 = time.time()
        lp = orc.reconstruct_lp(sc['sh_coeff'][0], basis)[None]
        orc.render_frame(mesh_t, probe, args.img_size // 4, sc['textures'], sc['unet_sd'], lp, sc['pivots_spec'], sc['pivots_diff'])
        sweep[n] = time.time() - t0
    cores = min(sweep, key=sweep.get)
    set_threads(cores)
    for i in range(3):                      # warm-up frames at full size
        frame(views_all[i])
    times, ref = [], None
    for i, views in enumerate(views_all[3:]):
        # bounded sample: stop early once the budget is spent, but always render the parity frame (the last one)
        last = i == len(views_all) - 4
        if not last and len(times) >= 3 and time.time() - t_start > budget_s:
            continue
        t0 = time.time()
        ref = frame(views)
        times.append(time.time() - t0)
    times = times[:n_timed] if len(times) > n_timed else times     # the parity frame is the 11th when nothing was skipped
    med = float(np.median(times))
    out = {'value': 1.0 / med, 'unit': 'frames/s', 'cores': cores, 'kind': 'port', 'cpu_model': cpu_model(),
           'host_logical_cpus': ncpu, 'seconds_per_frame_median': med, 'timed_frames': len(times), 'warmup_frames': 3,
           'thread_sweep_seconds_per_probe_frame': {str(k): v for k, v in sorted(sweep.items())},
           'sample': '3 warm-up + %d timed frames %dx%d (median; BASELINE.md §3 asks for 10, fewer only if the %d s budget ran '
                     'out), same scene/weights as the GPU run, one view per call like test_rnr.py:265.  kind "port": the oracle '
                     '(OpenMP C rasterizer + torch-CPU fp32 shading / U-Net) executes the LIVE 428.7 GFLOP of the U-Net only, '
                     'whereas the reference also runs a dead 928.8 GFLOP GCN pass per frame (1357 in total, '
                     'pytorch_prototyping.py:407-422).  %d threads = the fastest of a sweep over {64, 32, 16, 8} on a quarter-resolution '
                     'probe frame; BASELINE.md §3\'s "all host cores" oversubscribes torch-CPU on this host (256 threads: 116 s per '
                     'full frame, 81 s per probe frame, measured once in r03); pose tensors and the lmax-10 lighting basis prepared '
                     'outside the timer'
                     % (len(times), args.img_size, args.img_size, int(budget_s), cores)}
    parity = None
    if hip_image is not None:
        parity = {'psnr_db_vs_oracle': orc.psnr(hip_image.cpu(), ref['image']),
                  'max_abs_err': float((hip_image.cpu() - ref['image']).abs().max())}
        for prec, img in (emu_image or {}).items():
            parity[prec + '_psnr_db_vs_oracle'] = orc.psnr(img.cpu(), ref['image'])
            parity[prec + '_max_abs_err'] = float((img.cpu() - ref['image']).abs().max())
    return out, parity


def pmc_traffic_per_step(args, views_per_step=None):
    """HBM-side bytes of the conv kernels per step from rocprofv3 PMC passes (profiles/README.md):
    (2 x FETCH_SIZE + WRITE_SIZE) KB, FETCH doubled as MI355X_MICROARCH.md §HBM prescribes for gfx950.  PMC passes
    serialise kernels and cannot run inside a timed bench run, so the counters come from a separate run of the SAME
    command line (scripts/pmc.sh): either the file given by --pmc-file or the newest committed profile whose recorded
    batch size / precision / image size equal this run's.  Never rescaled from another batch size.
    Returns (bytes per step | None, info dict)."""
    import glob
    import re
    V = args.views_per_step if views_per_step is None else views_per_step
    own = args.pmc_file and views_per_step is None
    cands = [args.pmc_file] if own else sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_per_kernel_*.json')))[::-1]
    for f in cands:
        try:
            prof = json.load(open(f))
        except (OSError, ValueError):
            continue
        meta = prof.get('_meta')
        if meta is None:        # round-1 files carry no meta block: their name states the batch size, f32 unless tagged
            mv = re.search(r'views(\d+)', os.path.basename(f))
            ms = re.search(r'steps(\d+)', os.path.basename(f))
            meta = {'views_per_step': int(mv.group(1)) if mv else -1, 'steps': int(ms.group(1)) if ms else 2, 'warmup': 1,
                    'precision': 'bf16x6' if 'bf16x6' in os.path.basename(f) else 'f32', 'img_size': 512}
        if (meta['views_per_step'], meta['precision'], meta.get('img_size', 512)) != (V, args.precision, args.img_size):
            continue
        # profiles recorded before the Winograd kernels existed carry no conv_algo: they are the direct path's
        algo = (args.conv_algo or os.environ.get('RNR_CONV_ALGO') or 'winograd4') if args.precision == 'f32' else 'direct'
        if meta.get('conv_algo', 'direct') != algo:
            continue
        steps = meta['steps'] + meta['warmup']
        kb = sum(2.0 * v.get('FETCH_SIZE_total', 0.0) + v.get('WRITE_SIZE_total', 0.0)
                 for k, v in prof.items() if k.startswith('conv_') or 'conv_' in k.split('(')[0])
        if kb <= 0:
            continue
        return kb * 1024.0 / steps, {'traffic_source': os.path.basename(f), 'traffic_path': os.path.abspath(f),
                                     'traffic_from_committed_profile': not own,
                                     'traffic_profile': meta}
    return None, {'traffic_source': None, 'traffic_from_committed_profile': False,
                  'traffic_note': 'no PMC profile recorded at views_per_step=%d precision=%s: run scripts/pmc.sh and pass '
                                  '--pmc-file' % (V, args.precision)}


def sustained_mfma_tflops():
    """What the matrix cores SUSTAIN with random operands (scripts/micro/mfma_peak.hip: register-resident MFMA loops on every
    SIMD, ~0.25 s per case), from the newest committed profiles/r*_mfma_peak_micro.json: {instruction: TFLOP/s} or {}.  The
    16-bit MFMAs clock down to 1.6 - 1.8 GHz under random data (power), so the nominal 2.5 PFLOP/s is not reachable by any
    kernel with live operands; the f32 MFMA holds 2.35 GHz.  Reported BESIDE the nominal-peak fractions, never instead."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_mfma_peak_micro.json')) +
                   glob.glob(os.path.join(ROOT, 'profiles', 'archive', 'r*_mfma_peak_micro.json')), key=os.path.basename)
    if not files:
        return {}, None
    try:
        cases = json.load(open(files[-1]))['cases']
    except (OSError, ValueError, KeyError):
        return {}, None
    return {c['instruction']: c['tflops'] for c in cases if c['operands'] == 'unit_normal'}, os.path.basename(files[-1])


class _StubPipeline:
    """--stub-pipeline: a CPU stand-in with RNRPipeline.render's contract (two frame buffers used alternately) whose frames
    encode their pose, so that the gather check means something.  Control-flow tests only."""

    def __init__(self, V, S, dev):
        self._images = [torch.empty(V, 3, S, S, dtype=torch.float32, device=dev) for _ in range(2)]
        self._flip = 0

    def render(self, proj, pose, proj_inv, R_inv):
        img = self._images[self._flip][:proj.shape[0]]
        self._flip ^= 1
        code = pose[:, :3, 3] * 2.0 + proj[:, 0, 0:1] * 1e-3          # [n,3]
        img.copy_(code[:, :, None, None].expand_as(img))
        return img


def sustained_block(precision, achieved_tf):
    """{'sustained_peak': ..., 'frac_of_sustained': ...} for a roofline block (empty if no microbenchmark is committed)."""
    rates, src = sustained_mfma_tflops()
    ins, products = {'f32': ('v_mfma_f32_32x32x2_f32', 1), 'bf16x6': ('v_mfma_f32_32x32x16_bf16', 6),
                     'f16x3': ('v_mfma_f32_32x32x16_f16', 3)}[precision]
    if ins not in rates:
        return {}
    peak = rates[ins] / products
    return {'sustained_peak': peak, 'frac_of_sustained': achieved_tf / peak,
            'sustained_peak_source': '%s: %s with random operands sustains %.1f TFLOP/s%s' % (
                src, ins, rates[ins], '' if products == 1 else ' (/ %d partial products)' % products)}


def pmc_mfma_flops_per_step(info):
    """Executed MFMA FLOPs per step counted by the hardware: SUM SQ_INSTS_MFMA x FLOPs per wave instruction (4096 for
    v_mfma_f32_32x32x2_f32, 2048 for the out layer's v_mfma_f32_16x16x4_f32 kernel) over the conv kernels of the SAME PMC
    file roofline.traffic comes from, / (steps + warm-up) of that profile run.  None without such a file / counter."""
    path = info.pop('traffic_path', None)       # a path of this machine: used here, not printed
    if not path:
        return None
    try:
        prof = json.load(open(path))
    except (OSError, ValueError):
        return None
    meta = info.get('traffic_profile') or prof.get('_meta') or {}
    steps = meta.get('steps', 2) + meta.get('warmup', 1)
    tot = 0.0
    for k, v in prof.items():
        if k == '_meta' or 'SQ_INSTS_MFMA_total' not in v or 'conv_' not in k:      # the convolutions only (not calibrate_mfma_f32_kernel)
            continue
        tot += v['SQ_INSTS_MFMA_total'] * (2048.0 if 'conv_wino80' in k else 4096.0)
    return tot / steps if tot > 0 else None


def algo_block(unet, n_views, stage_ms, peak, masked_out_layer, direct_tf=None, traffic_info=None):
    """The MFMA roofline of a U-Net stage: `achieved` / `frac` count what the matrix cores EXECUTE (so frac <= 1 by construction);
    the direct-form (SURVEY 8(d) algorithmic) figure the Winograd kernels replace is reported beside it, never as frac."""
    ex = unet.mfma_flops_per_view(n_views, masked_out_layer) * n_views
    tf = ex / (stage_ms * 1e-3) / 1e12
    algos = [unet.L.rnr_conv_algorithm(ctypes.byref(s['desc']), n_views, *s['in_hw']) for s in unet.steps]
    if masked_out_layer and algos[-1] != 3:
        algos[-1] = 0
    blk = {'achieved': tf, 'peak': peak, 'unit': 'TFLOP/s', 'frac': tf / peak,
           'conv_algo': unet.conv_algo,
           'layers_direct_winograd3x3_winograd2x2': [algos.count(0), algos.count(1) + algos.count(3) + algos.count(4), algos.count(2)],
           'layers_winograd_f4x4_3x3': algos.count(4),
           'executed_mfma_flops': ex}
    pmc = pmc_mfma_flops_per_step(traffic_info or {})
    blk['executed_flops_from_pmc'] = pmc
    if pmc:
        blk['executed_flops_pmc_over_model'] = pmc / ex
    if direct_tf is not None:
        blk['effective_tflops_direct_form'] = direct_tf
        blk['algorithmic_speedup'] = direct_tf / tf
    blk['algo_note'] = ("achieved / frac = multiply-add FLOPs the matrix cores execute (padding columns / channels included: what "
                        "SQ_INSTS_MFMA counts, cross-checked by executed_flops_from_pmc = SUM SQ_INSTS_MFMA x 4096 | 2048 of the PMC "
                        "file named in traffic_source) / stage time / peak.  effective_tflops_direct_form = the ALGORITHMIC FLOPs of "
                        "SURVEY 8(d) (direct-form convolutions) / the same time: the Winograd 3x3 layers execute 16 instead of 36 "
                        "multiplications per 2x2 outputs (F(2x2,3x3)) or 36 instead of 144 per 4x4 outputs (F(4x4,3x3), conv_algo "
                        "'winograd4') and the 4x4 stride-2 / transposed ones 9 instead of 16 (fp32 operands, fp32 accumulation, same "
                        "MFMA instruction), algorithmic_speedup = their ratio")
    return blk


def step_series_block(series, dt, K):
    """Per-step attribution of the timed window (rank 0's events): where a step's wall time went, and which steps stand out.
      step_ms                  HIP-event interval start -> end of each step on the launch stream
      gap_ms                   end of step k -> start of step k + 1 on the GPU (idle unless the host is behind)
      unet_ms                  the U-Net stage inside each step (HIP events)
      non_unet_ms_per_step     mean(step_ms - unet_ms): rasterizer + shading inputs + ray renderer + whatever idle time
                               the stream saw inside the step
      gpu_idle_ms_per_step     (window wall time - sum(step_ms)) / K + mean over steps of (step_ms - min(step_ms)): wall time not
                               covered by a step's events plus what the steps took beyond the fastest one (same kernels, same
                               sizes: the excess is idle stream time or a clock dip)
      slow_steps               steps > 1.3 x the median step_ms, by index"""
    out = {'host_enqueue_ms': [round(x, 3) for x in series.get('host_enqueue_ms', [])]}
    if 'step_ms' not in series:
        return {'step_series': out}
    st, un = series['step_ms'], series['unet_ms']
    med = float(np.median(st))
    out.update({
        'step_ms': [round(x, 3) for x in st], 'gap_ms': [round(x, 3) for x in series['gap_ms']],
        'unet_ms': [round(x, 3) for x in un],
        'ms_per_step_median': med, 'ms_per_step_min': float(np.min(st)), 'ms_per_step_max': float(np.max(st)),
        'non_unet_ms_per_step': float(np.mean(np.asarray(st) - np.asarray(un))),
        'gpu_idle_ms_per_step': float((dt * 1e3 - np.sum(st)) / K + np.mean(np.asarray(st) - np.min(st))),
        'wall_minus_event_span_ms': dt * 1e3 - series['events_span_ms'],
        'slow_steps': [{'step': int(i), 'ms': round(float(x), 3), 'unet_ms': round(float(un[i]), 3)}
                       for i, x in enumerate(st) if x > 1.3 * med],
    })
    return {'step_series': out}


def make_pipeline(sc, args, dev, V, **kw):
    from rnr_amd.pipeline import RNRPipeline
    opts = dict(nf0=args.nf0, max_views=V, device=dev, sh_coeff=sc['sh_coeff'], sh_lmax=10,
                skip_background_tiles=args.tile_skip, precision=args.precision, conv_algo=args.conv_algo)
    opts.update(kw)
    return RNRPipeline(sc['mesh'], args.img_size, sc['textures'], sc['unet_sd'], sc['pivots_spec'], sc['pivots_diff'], None, **opts)


def hook_unet_events(unet, n_events, series=False):
    """HIP events around UNetPlan.forward on the launch stream; returns restore(), which un-hooks and gives the mean
    interval in ms (series=True: the list of per-call intervals)."""
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * n_events)]
    orig = unet.forward
    count = [0]

    def timed_forward(net_in, n_views=None, consumer_alpha=None):
        k = count[0]
        if k < n_events:
            ev[2 * k].record()
        r = orig(net_in, n_views, consumer_alpha)
        if k < n_events:
            ev[2 * k + 1].record()
        count[0] += 1
        return r
    unet.forward = timed_forward

    def restore():
        unet.forward = orig
        ms = [ev[2 * i].elapsed_time(ev[2 * i + 1]) for i in range(min(count[0], n_events))]
        return ms if series else float(np.mean(ms))
    return restore


def config5_block(args, dev):
    """BASELINE configs[4] / SURVEY 8(d) config 5 at its per-GPU shape, timed by the driver's run: 1024 x 1024, the 65 536-face
    sphere, 16-channel neural texture (U-Net 100 -> 78, nf0 as the headline), lighting = network.LightingLP on the 1600 x 3200
    synthetic environment map (8 Gaussians, seed 2) -> 4096 bilinear samples -> SH fit, lmax 10 (network.py:631-699), 2 views
    per step (one GPU's share of a small batch; the N > 1 form shards views exactly like the headline).  The light-probe
    front-end is timed once (it is per lighting, not per view).  Parity of this shape:
    tests/test_gpu_frame.py::test_config5_as_written_1024_c16_65536_faces_probe_1600x3200."""
    import network
    from rnr_amd import scene
    S, C, V5 = 2 * args.img_size, 16, 2
    a5 = argparse.Namespace(**vars(args))
    a5.img_size, a5.tex_ch = S, C
    sc5 = build_scene(a5)
    env = scene.synthetic_light_probe(1600, 3200, 2)[0]
    l_dir = torch.from_numpy(scene.sphere_samples(4096)).t().contiguous()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lp_model = network.LightingLP(l_dir, num_channel=3, lp_dataloader=[{'lp_img': env.permute(2, 0, 1)[None]}], fix_params=True,
                                  device=dev)
    lp_model.fit_sh(lmax=10)
    torch.cuda.synchronize()
    t_light = time.perf_counter() - t0
    sc5['sh_coeff'] = lp_model.sh_coeff
    pipe = make_pipeline(sc5, a5, dev, V5)
    K = max(4, min(args.steps, 10))
    ids = (np.arange((K + 2) * V5) * 7) % 720
    pv = {k: torch.from_numpy(v).to(dev) for k, v in scene.spiral_views(S, ids).items()}

    def st(s):
        sl = slice(s * V5, (s + 1) * V5)
        return pipe.render(pv['proj'][sl], pv['pose'][sl], pv['proj_inv'][sl], pv['R_inv'][sl])
    for s in range(2):
        st(s)
    restore = hook_unet_events(pipe.unet, K)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(2, 2 + K):
        st(s)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ums = restore()
    tf_direct = pipe.unet.flops_per_view * V5 / (ums * 1e-3) / 1e12
    blk = algo_block(pipe.unet, V5, ums, EMU_PEAK['f32'], False, tf_direct, {})
    blk.pop('algo_note', None)
    return {'workload': 'BASELINE configs[4] per-GPU shape: %dx%d, 65536 faces, 16-ch neural texture (U-Net %d -> %d, nf0=%d), '
                        'LightingLP(1600x3200 synthetic probe) -> 4096 samples -> SH lmax 10, %d views per step'
                        % (S, S, sc5['c_in'], 3 * sc5['n_rays'], args.nf0, V5),
            'frames_per_s': K * V5 / dt, 'ms_per_step': dt / K * 1e3, 'steps': K, 'views_per_step': V5,
            'light_probe_front_end_s': t_light,
            'roofline': {'bound': 'mfma', **blk, 'stage_ms_per_step': ums, 'alg_flops_per_view': pipe.unet.flops_per_view},
            'note': 'not the headline value; same pipeline object and kernels, exact fp32'}


def single_view_block(sc, args, dev):
    """The reference's calling mode (test_rnr.py:265-393): the spiral_step720 views in order, ONE view per call.
      sequential        every call on one stream, nothing else in flight: per-frame latency; HIP events around the U-Net of
                        every call give this mode's own roofline block;
      two_calls_in_flight   RNRPipeline(inflight=2).submit: call i on HIP stream i % 2 with private activations — the
                        rasterizer / shading kernels of view i+1 and the tails of the 22 short conv launches run under the
                        U-Net of view i.  Frames are the same."""
    from rnr_amd import scene
    n1 = max(8, int(args.single_views))
    ids = np.arange(n1) % 720
    pv = {k: torch.from_numpy(v).to(dev) for k, v in scene.spiral_views(args.img_size, ids).items()}
    pipe = make_pipeline(sc, args, dev, 1, inflight=2)

    def pose(i):
        return pv['proj'][i:i + 1], pv['pose'][i:i + 1], pv['proj_inv'][i:i + 1], pv['R_inv'][i:i + 1]
    for i in range(5):
        pipe.render(*pose(i))
    restore = hook_unet_events(pipe.unet, n1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n1):
        pipe.render(*pose(i))
    torch.cuda.synchronize()
    dt_seq = (time.perf_counter() - t0) / n1
    unet_ms = restore()
    seq_last = pipe.render(*pose(n1 - 1)).clone()
    for i in range(6):
        pipe.submit(*pose(i))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hs = None
    for i in range(n1):
        hs = pipe.submit(*pose(i))
    torch.cuda.synchronize()
    dt_fly = (time.perf_counter() - t0) / n1
    same = float((hs.image - seq_last).abs().max())
    flops_view = pipe.unet.flops_per_view
    traffic, tinfo = pmc_traffic_per_step(args, views_per_step=1)
    tf = flops_view / (unet_ms * 1e-3) / 1e12
    algo1 = (algo_block(pipe.unet, 1, unet_ms, EMU_PEAK[args.precision], args.tile_skip, tf, tinfo) if args.precision == 'f32'
             else {'achieved': tf, 'peak': EMU_PEAK[args.precision], 'unit': 'TFLOP/s', 'frac': tf / EMU_PEAK[args.precision]})
    del pipe
    # ... and three (one more private stream / activation set; a fourth would share a hardware queue: slower again)
    pipe3 = make_pipeline(sc, args, dev, 1, inflight=3)
    for i in range(6):
        pipe3.submit(*pose(i))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n1):
        hs = pipe3.submit(*pose(i))
    torch.cuda.synchronize()
    dt_fly3 = (time.perf_counter() - t0) / n1
    same3 = float((hs.image - seq_last).abs().max())
    del pipe3
    return {
        'views_per_call': 1, 'views': n1,
        'workload': 'test_rnr.py:265-393: spiral_step720 views in order, one view per call, %dx%d, full HIP RenderingNet' % (args.img_size, args.img_size),
        'frames_per_s': 1.0 / dt_seq, 'ms_per_frame': dt_seq * 1e3,
        'roofline': {'bound': 'mfma', 'kernel': 'conv_wino4_kernel / conv_wino_kernel / conv_wino2p_kernel / conv_wino2_kernel / conv_halo_kernel (22 conv launches per view; the split-K Winograd layers are '
                                                  'followed by a reduce launch and the one-workgroup-per-CU grids by a BatchNorm finalise launch of their own: '
                                                  '14 + 14 launches at 512^2, profiles/r06_frame_timeline_f32_views1.txt; HIP events bracket the U-Net stage of every call)',
                     **algo1, 'stage_ms_per_view': unet_ms, **sustained_block(args.precision, algo1['achieved']),
                     'alg_flops_per_view': flops_view, 'traffic': traffic,
                     'traffic_unit': 'bytes/view (HBM-side, PMC: (2 x FETCH_SIZE + WRITE_SIZE) KB of the conv kernels)',
                     **{k: v for k, v in tinfo.items() if k != 'traffic_path'}},
        'two_calls_in_flight': {'frames_per_s': 1.0 / dt_fly, 'ms_per_frame': dt_fly * 1e3,
                                'max_abs_diff_vs_sequential_last_frame': same,
                                'note': 'RNRPipeline(inflight=2).submit: throughput of the same one-view calls with two in flight '
                                        '(per-frame latency is the sequential figure)'},
        'three_calls_in_flight': {'frames_per_s': 1.0 / dt_fly3, 'ms_per_frame': dt_fly3 * 1e3,
                                  'max_abs_diff_vs_sequential_last_frame': same3,
                                  'note': 'RNRPipeline(inflight=3).submit'},
    }


def dropin_view_loop_block(sc, args, dev):
    """INTEGRATION.md Level 1 on the record: test_rnr.py:265-377's own call sequence through the DROP-IN modules
    (rnr_amd.view_loop.DropinViewLoop: network.Rasterizer -> render.get_TBN_map -> camera.get_view_dir_map -> torch.matmul ->
    sph_harm.evaluate_sh_basis -> TextureMapper -> 2 x RaySampler -> torch.cat -> RenderingNet -> post-scale -> RayRenderer), one
    view per call over the spiral_step720 views, nothing fused, the reference's intermediate tensors materialised in HBM.
      frames_per_s               wall clock over the views, test_rnr.py:322-328 verbatim (view directions to host numpy, basis back
                                 as float64 numpy, cast, upload: the host round trips the reference's numpy contract mandates);
      sh_basis_on_device         the same loop with the one-line variant evaluate_sh_basis(..., as_tensor=True);
      stage_ms                   HIP events at the call boundaries (mean over 48 views of each variant)."""
    import tempfile
    from rnr_amd import scene
    from rnr_amd.view_loop import DropinViewLoop, stage_table
    n1 = max(8, int(args.single_views))
    ids = np.arange(n1) % 720
    pv = {k: torch.from_numpy(v).to(dev) for k, v in scene.spiral_views(args.img_size, ids).items()}

    def pose(i):
        return pv['proj'][i:i + 1], pv['pose'][i:i + 1], pv['proj_inv'][i:i + 1], pv['R_inv'][i:i + 1]
    with tempfile.TemporaryDirectory() as td:
        obj = os.path.join(td, 'mesh.obj')
        scene.write_obj(obj, sc['mesh'])
        t0 = time.perf_counter()
        loop = DropinViewLoop(obj, args.img_size, sc['textures'], sc['unet_sd'], sc['sh_coeff'], nf0=args.nf0, device=dev)
        t_build = time.perf_counter() - t0
    out = {}
    last = {}
    for tag, on_dev in (('numpy_contract', False), ('sh_basis_on_device', True)):
        loop.sh_on_device = on_dev
        for i in range(5):
            loop.view(*pose(i))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        img = None
        for i in range(n1):
            img = loop.view(*pose(i))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n1
        last[tag] = img.clone()
        evs = []
        for i in range(48):
            e = []
            loop.view(*pose(i), events=e)
            evs.append(e)
        torch.cuda.synchronize()
        tab = stage_table(evs)
        out[tag] = {'frames_per_s': 1.0 / dt, 'ms_per_frame': dt * 1e3, 'stage_ms': tab, 'stage_ms_sum': float(sum(tab.values()))}
    # the same pose through the fused Level-2 pipeline
    pipe = make_pipeline(sc, args, dev, 1)
    ref = pipe.render(*pose(n1 - 1))
    diff = {k: float((v - ref).abs().max()) for k, v in last.items()}
    del pipe, loop
    worst = max(out['numpy_contract']['stage_ms'].items(), key=lambda kv: kv[1])
    return {
        'workload': 'test_rnr.py:265-377 call by call through the drop-in modules (INTEGRATION.md Level 1), %d spiral_step720 views '
                    'in order, one view per call, %dx%d, 65536 faces, nf0=%d' % (n1, args.img_size, args.img_size, args.nf0),
        'frames_per_s': out['numpy_contract']['frames_per_s'], 'ms_per_frame': out['numpy_contract']['ms_per_frame'],
        'stage_ms': out['numpy_contract']['stage_ms'], 'stage_ms_sum': out['numpy_contract']['stage_ms_sum'],
        'slowest_stage': {'name': worst[0], 'ms': worst[1]},
        'sh_basis_on_device': out['sh_basis_on_device'],
        'max_abs_diff_vs_fused_pipeline_last_frame': diff,
        'module_construction_s': t_build,
        'note': 'frames_per_s keeps the reference\'s numpy contract for the SH basis (sph_harm.py:41-71: host numpy in, float64 numpy '
                'out, then .astype(float32) and torch.from_numpy(...).to(device) in test_rnr.py:324-328): two host round trips and a '
                'stream drain per view that no drop-in can remove; sh_basis_on_device is the same loop with '
                'evaluate_sh_basis(..., as_tensor=True).  stage_ms are HIP-event intervals between the reference\'s own call boundaries '
                '(host-side waits of a stage appear in it as idle GPU time).  The script\'s batched torch.matmul of test_rnr.py:314 is '
                'answered by rnr_tbn_matvec (render.get_TBN_map returns a render.TBNMap; INTEGRATION.md Level 1).  Not the headline value.',
    }


def workload_name(args, world, V, sc):
    """config.workload: which BASELINE.json config the line measures.  N = 1: configs[2] (spiral_step720 at 512^2, full HIP
    RenderingNet on one MI355X) in V-view batches.  N > 1: configs[3] (spiral_step720 views sharded across the GPUs of one node,
    frames all-gathered over RCCL / xGMI) with the global batch stated: configs[3] quotes batch = 64 on 8 GPUs (8 per GPU); the
    headline keeps the N = 1 line's per-GPU batch so that per-GPU work is the same at every N (weak scaling), and the exact
    configs[3] batch (8 views per GPU) is timed in the same run as `with_8_views_per_gpu`."""
    net = ('%dx%d, full HIP RenderingNet (f32 MFMA convs + SH relight), UV-sphere 65536 faces, neural texture 512^2 x %d ch x 4 '
           'levels, U-Net %d->%d nf0=%d' % (args.img_size, args.img_size, args.tex_ch, sc['c_in'], 3 * sc['n_rays'], args.nf0))
    if world == 1:
        return ('BASELINE configs[2] in %d-view batches (the reference renders 1 view per call; that mode is reported as '
                'single_view_mode with its own roofline): test_rnr.py spiral_step720 views, %s' % (V, net))
    return ('BASELINE configs[3]: spiral_step720 views sharded across %d x MI355X, one process per GPU, frames all-gathered over RCCL / '
            'xGMI every step; GLOBAL batch = %d views per step = %d per GPU x %d GPUs (configs[3] quotes batch = 64 = 8 per GPU on 8 '
            'GPUs; this line keeps the N = 1 headline\'s %d views per GPU so that per-GPU work is fixed across N — weak scaling — and '
            'the 8-per-GPU batch of configs[3] is timed in the same run as with_8_views_per_gpu); %s' % (world, world * V, V, world, V, net))


def launch_ranks(n, argv):
    """`python bench.py --gpus N` WITHOUT a launcher: become the launcher.  Starts N copies of this command line, one per GPU,
    with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set (127.0.0.1, a port the kernel hands out), passes their
    stdout / stderr through (rank 0 prints the JSON line), and returns the worst exit code; when one rank dies the others are
    terminated by their exact PIDs so that a broken rendezvous cannot hang the launch."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RNR_BENCH_SELF_LAUNCHED='1')
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.05)
        for p in list(live):
            c = p.poll()
            if c is None:
                continue
            live.remove(p)
            if c != 0 and rc == 0:
                rc = c
                for q in live:
                    q.terminate()
    return rc


def main(argv=None):
    args = parse(argv)
    if args.gpus < 1:
        raise SystemExit('bench.py: --gpus must be >= 1')
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # plain launch asking for N GPUs: this process becomes the launcher of N ranks (one per GPU)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:] if argv is None else argv))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks: refusing to report a number for a '
                         'job of another size' % (args.gpus, world))
    stub = args.stub_pipeline
    if stub:
        dev = torch.device('cpu')
        sync = lambda: None
    else:
        if not torch.cuda.is_available():
            raise SystemExit('bench.py needs an MI355X: the hot path has no CPU fallback')
        torch.cuda.set_device(local_rank)
        dev = torch.device('cuda', local_rank)
        sync = torch.cuda.synchronize
    use_dist = world > 1 or os.environ.get('RNR_BENCH_FORCE_DIST') == '1'     # the env switch exercises the RCCL path on 1 GPU
    dist = None
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        if stub:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)       # "nccl" is RCCL on ROCm
    from rnr_amd import scene
    V = args.views_per_step
    if stub:
        sc = {'c_in': 0, 'n_rays': 0}
        pipe = _StubPipeline(V, args.img_size, dev)
    else:
        sc = build_scene(args)
        pipe = make_pipeline(sc, args, dev, V)
    # pose slices: step s, rank r renders spiral views (s*world + r)*V ... +V  (mod 720)
    n_total = (args.steps + args.warmup) * world * V
    ids = (np.arange(n_total) * 7) % 720
    allv = scene.spiral_views(args.img_size, ids)
    poses = {k: torch.from_numpy(v).to(dev) for k, v in allv.items()}
    # frames over xGMI — the only exchange of the path, issued asynchronously so that the gather of step s overlaps the
    # rendering of step s+1 (rnr_amd.dist.OverlappedFrameGather; the pipeline alternates two frame buffers)
    gather = None
    if use_dist:
        from rnr_amd.dist import OverlappedFrameGather
        gather = OverlappedFrameGather(world, (V, 3, args.img_size, args.img_size), torch.float32, dev)

    def step(s):
        lo = (s * world + rank) * V
        sl = slice(lo, lo + V)
        img = pipe.render(poses['proj'][sl], poses['pose'][sl], poses['proj_inv'][sl], poses['R_inv'][sl])
        if use_dist:
            gather.submit(img)
        return img

    def drain():
        if use_dist:
            gather.drain()

    # pre-warm (untimed, in FRONT of the W warm-up steps): a fresh box hands over a GPU in its idle power state and a process
    # whose allocator, code objects and clocks have seen nothing yet; W = 3..5 steps are 0.1 s.  Steps are rendered until
    # --prewarm-seconds have passed (host clock, synchronised) — the timed region below is untouched: exactly K steps.
    prewarm_steps = 0
    prewarm_s = args.prewarm_seconds if not stub else min(args.prewarm_seconds, 0.2)      # the stub run only exercises the control flow
    if prewarm_s > 0:
        tp = time.perf_counter()
        while prewarm_steps < 200:
            go = time.perf_counter() - tp < prewarm_s
            if use_dist:        # every rank must run the same number of steps (each step is a collective): rank-agreed decision
                flag = torch.tensor([1 if go else 0], device=dev, dtype=torch.int32)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                go = bool(flag.item())
            if not go:
                break
            step(prewarm_steps % max(1, args.warmup + args.steps))
            prewarm_steps += 1
            sync()
        drain()
    for s in range(args.warmup):
        step(s)
    drain()
    # ---- HIP events on the launch stream: around every step and around the dominant stage (U-Net convs) of every step ----
    active_tiles, out_tiles_per_step, out_step = None, 0, None
    unet_forward_plain = None
    if not stub:
        out_step = pipe.unet.steps[-1]
        out_tiles_per_step = pipe.unet.L.rnr_conv_tile_count(ctypes.byref(out_step['desc']), V, args.img_size, args.img_size)
        active_tiles = torch.zeros(1, dtype=torch.int64, device=dev)     # out-layer pixel tiles actually computed
        unet_forward_plain = pipe.unet.forward

    def timed_window():
        """EXACTLY K steps between barrier + synchronize on both sides.  Returns (wall seconds, series) where series holds,
        per step, the HIP-event interval start -> end of the step on the launch stream (step_ms), the interval from its end
        to the next step's start (gap_ms: GPU idle between steps, 0 while the host runs ahead), the U-Net stage inside it
        (unet_ms) and the host time spent enqueueing it (host_enqueue_ms) — test_rnr.py:265, 374 prints the same per-view
        stamps (t_prep t_raster t_preproc t_sh t_network t_render)."""
        K = args.steps
        ev_s = ev_e = None
        restore_u = None
        if not stub:
            ev_s = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
            ev_e = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
            pipe.unet.forward = unet_forward_plain
            restore_u = hook_unet_events(pipe.unet, K, series=True)
            hooked = pipe.unet.forward

            def counting_forward(net_in, n_views=None, consumer_alpha=None):
                r = hooked(net_in, n_views, consumer_alpha)
                if consumer_alpha is not None and out_tiles_per_step:
                    active_tiles.add_(pipe.unet._tile_mask[:out_tiles_per_step].sum())
                return r
            pipe.unet.forward = counting_forward
        host_ms = []
        if use_dist:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        img_ = None
        for k, s in enumerate(range(args.warmup, args.warmup + K)):
            th = time.perf_counter()
            if ev_s is not None:
                ev_s[k].record()
            img_ = step(s)
            if ev_e is not None:
                ev_e[k].record()
            host_ms.append((time.perf_counter() - th) * 1e3)
        drain()
        sync()
        if use_dist:
            dist.barrier()
        sync()
        dt_ = time.perf_counter() - t0
        ser = {'host_enqueue_ms': host_ms}
        if ev_s is not None:
            ser['step_ms'] = [ev_s[k].elapsed_time(ev_e[k]) for k in range(K)]
            ser['gap_ms'] = [ev_e[k].elapsed_time(ev_s[k + 1]) for k in range(K - 1)]
            ser['unet_ms'] = restore_u()
            pipe.unet.forward = unet_forward_plain
            ser['events_span_ms'] = ev_s[0].elapsed_time(ev_e[K - 1])
        return dt_, ser, img_

    def calibrate():
        """What every rank's GPU sustains right now in a register-resident f32-MFMA loop (rnr_calibrate_mfma_f32, ~0.1 s):
        [TFLOP/s per rank] or None.  Outside every timed region.  Boxes of one pool differ by several per cent in the
        clock they hold under matrix load (r06: 552 / 577 / 580 frames/s from the same commit on three boxes): the
        headline is only comparable across boxes beside this figure."""
        if stub or args.no_calibration:
            return None
        from rnr_amd import ops
        tf = ops.calibrate_mfma_f32(dev, 0.1)['tflops']
        # ... and a device-to-device copy of 512 MiB (read + write counted), best of five: the HBM side of the same question
        a = torch.empty(128 << 20, dtype=torch.float32, device=dev)
        b = torch.empty_like(a)
        b.copy_(a)
        best = 0.0
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            b.copy_(a)
            e1.record()
            e1.synchronize()
            best = max(best, 2.0 * a.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9)
        del a, b
        if not use_dist:
            return [(tf, best)]
        mine = torch.tensor([tf, best], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        return [(float(t[0].item()), float(t[1].item())) for t in allr]

    n_windows = 1 if stub else max(1, args.windows)
    windows = []
    img = None
    calib_before = calibrate()
    for w in range(n_windows):
        dt_w, ser_w, img = timed_window()
        if use_dist:
            tmax = torch.tensor([dt_w], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt_w = float(tmax.item())
        windows.append((dt_w, ser_w))
    calib_after = calibrate()
    # value policy: the MEDIAN window (by max-over-ranks wall time) of n_windows windows of exactly K steps each; every
    # window's rate is printed (`windows`), the series below belong to the window `value` is taken from
    order = sorted(range(n_windows), key=lambda i: windows[i][0])
    w_sel = order[(n_windows - 1) // 2]
    dt, series = windows[w_sel]
    unet_ms = float(np.mean(series['unet_ms'])) if 'unet_ms' in series else dt / args.steps * 1e3
    gather_check = None
    if args.check_gather and use_dist:
        got = gather.latest[rank * V:(rank + 1) * V]
        bad = torch.tensor([0 if torch.equal(got, img) else 1, int(gather.latest.shape[0] != world * V)], device=dev)
        dist.all_reduce(bad, op=dist.ReduceOp.SUM)
        gather_check = {'ok': int(bad.sum().item()) == 0, 'ranks_with_mismatch': int(bad[0].item()),
                        'gathered_shape': list(gather.latest.shape), 'backend': dist.get_backend()}
    last_frame = img[V - 1:V].clone()       # the frame buffers are reused by the extra renders below
    # N > 1: BASELINE configs[3]'s own batch — 8 views per GPU (64 at 8 GPUs) — through the same pipeline, gather and timing
    # protocol, every rank taking part; reported beside the headline as with_8_views_per_gpu
    dt8 = None
    V8 = 8 if not stub else max(1, V // 2)
    if use_dist and world > 1 and V > V8 and os.environ.get('RNR_BENCH_FAST') != '1':
        from rnr_amd.dist import OverlappedFrameGather
        gather8 = OverlappedFrameGather(world, (V8, 3, args.img_size, args.img_size), torch.float32, dev)

        def step8(s_):
            lo = ((s_ * world + rank) * V8) % (n_total - V8 + 1)
            sl = slice(lo, lo + V8)
            gather8.submit(pipe.render(poses['proj'][sl], poses['pose'][sl], poses['proj_inv'][sl], poses['R_inv'][sl]))
        for s_ in range(max(1, args.warmup)):
            step8(s_)
        gather8.drain()
        dist.barrier()
        sync()
        t8 = time.perf_counter()
        for s_ in range(args.steps):
            step8(s_)
        gather8.drain()
        sync()
        dist.barrier()
        sync()
        dt8 = time.perf_counter() - t8
        t8max = torch.tensor([dt8], device=dev, dtype=torch.float64)
        dist.all_reduce(t8max, op=dist.ReduceOp.MAX)
        dt8 = float(t8max.item())
        del gather8
    # ranks that actually took part: every rank adds 1 through the process group (not dist.get_world_size(), which only
    # repeats the environment); a mismatch with --gpus is a failed run, not a number
    n_ranks_seen = 1
    if use_dist:
        cnt = torch.ones(1, device=dev, dtype=torch.int64)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        n_ranks_seen = int(cnt.item())
    if n_ranks_seen != args.gpus:
        raise SystemExit('bench.py: --gpus %d but %d ranks took part' % (args.gpus, n_ranks_seen))
    # executed FLOPs: the live U-Net minus the out-layer tiles skipped because no pixel of them is ever read
    skipped, flops_step, n_conv, flops_view = 0.0, 0.0, 0, 0.0
    if not stub:
        d_out, (oh, ow) = out_step['desc'], out_step['in_hw']
        out_flops_view = 2 * oh * ow * 9 * (d_out.c_in0 + d_out.c_in1) * d_out.c_out
        if args.tile_skip and out_tiles_per_step:
            skipped = 1.0 - float(active_tiles.item()) / (out_tiles_per_step * args.steps * n_windows)
        flops_view = pipe.unet.flops_per_view
        flops_step = (flops_view - skipped * out_flops_view) * V
        n_conv = len(pipe.unet.steps)
    achieved_tf = flops_step / (unet_ms * 1e-3) / 1e12
    peak_tf = EMU_PEAK[args.precision]
    algo8 = {'achieved': achieved_tf, 'peak': peak_tf, 'unit': 'TFLOP/s', 'frac': achieved_tf / peak_tf}
    dtype = {'f32': 'f32', 'bf16x6': 'f32 emulated on bf16 MFMA (bf16x6)', 'f16x3': 'f32 emulated on f16 MFMA (f16x3)'}[args.precision]

    res = None
    if rank == 0:
        traffic, traffic_info = (None, {}) if stub else pmc_traffic_per_step(args)
        if not stub and args.precision == 'f32':
            algo8 = algo_block(pipe.unet, V, unet_ms, peak_tf, args.tile_skip, achieved_tf, traffic_info)
        try:
            rccl = '.'.join(str(x) for x in torch.cuda.nccl.version()) if not stub else None
        except Exception:       # noqa: BLE001 - version probing must never fail a run
            rccl = None
        res = {
            'metric': 'rendered frames/sec at %dx%d (material_sphere-like synthetic scene), full HIP RNR path'
                      % (args.img_size, args.img_size),
            'value': args.steps * world * V / dt, 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': dtype, 'data': 'synthetic',
            'n_ranks_seen': n_ranks_seen, 'rccl_version': rccl,
            'value_policy': 'median of %d timed windows of exactly %d steps each (barrier + synchronize around every window, MAX over '
                            'ranks per window); `windows` lists every window, the per-step series are those of the median window'
                            % (n_windows, args.steps),
            'windows': [{'frames_per_s': args.steps * world * V / w[0], 'ms_per_step': w[0] / args.steps * 1e3,
                         'is_value': i == w_sel} for i, w in enumerate(windows)],
            'prewarm_steps': prewarm_steps,
            **step_series_block(series, dt, args.steps),
            'config': {'workload': workload_name(args, world, V, sc),
                       'views_per_step_per_gpu': V, 'global_views_per_step': world * V,
                       'parallelism': 'views sharded x%d, all_gather of frames' % world,
                       'conv_algo': None if stub else pipe.unet.conv_algo},
            'roofline': {'bound': 'mfma', 'kernel': '%s (%d conv launches/step, BatchNorm finalise inside them; HIP events bracket the U-Net stage)' % ('conv_wino4_kernel / conv_wino2p_kernel / conv_wino2_kernel / conv_wino80_kernel' if args.precision == 'f32' else 'conv_halo_emu_kernel', n_conv),
                         **algo8, 'traffic': traffic,
                         'traffic_unit': 'bytes/step (HBM-side, PMC: (2 x FETCH_SIZE + WRITE_SIZE) KB of the conv kernels)',
                         **{k: v for k, v in traffic_info.items() if k != 'traffic_path'},
                         'alg_flops_per_step': flops_step, 'stage_ms_per_step': unet_ms,
                         **sustained_block(args.precision, algo8['achieved']),
                         'out_layer_tiles_skipped': skipped,
                         'flops_note': 'alg_flops_per_step (direct form) = %.1f GFLOP/view live U-Net minus the out-layer pixel tiles that hold no '
                                       'foreground pixel when --tile-skip is given (never read: the ray renderer zeroes background)'
                                       % (flops_view / 1e9)},
        }
        if calib_before is not None:
            res['box_calibration'] = {
                'instruction': 'v_mfma_f32_32x32x2_f32, register-resident loop on every SIMD (rnr_calibrate_mfma_f32, ~0.1 s per measurement)',
                'nominal_tflops': EMU_PEAK['f32'], 'tflops_per_rank_before_windows': [c[0] for c in calib_before],
                'tflops_per_rank_after_windows': [c[0] for c in calib_after],
                'frac_of_nominal': min(c[0] for c in calib_before + calib_after) / EMU_PEAK['f32'],
                'hbm_copy_GBps_per_rank_before_windows': [c[1] for c in calib_before],
                'hbm_copy_GBps_per_rank_after_windows': [c[1] for c in calib_after],
                'hbm_copy': 'torch device-to-device copy of 512 MiB, bytes read + written, best of five',
                'note': 'what THIS box sustains on the U-Net\'s instruction around the timed windows; the same commit measured 552, 577 and '
                        '580 frames/s on three boxes of the pool in r06 — compare `value` across boxes beside this number'}
        if stub:
            res['stub'] = True
            res['cpu_baseline'] = None
        if gather_check is not None:
            res['gather_check'] = gather_check
        if dt8 is not None:
            res['with_8_views_per_gpu'] = {
                'frames_per_s': args.steps * world * V8 / dt8, 'ms_per_step': dt8 / args.steps * 1e3,
                'views_per_step_per_gpu': V8, 'global_views_per_step': world * V8,
                'note': 'BASELINE configs[3] batch (8 views per GPU; 64 at 8 GPUs) with the same barrier + MAX-over-ranks protocol; '
                        'not the headline value'}
    if rank == 0 and not stub:
        extras = not args.main_loop_only
        fast = os.environ.get('RNR_BENCH_FAST') == '1'      # scripts/stage.sh: headline loop + per-stage figures only
        # per-stage HIP events (5 extra steps outside the timed region): the non-conv stages against the HBM roofline
        P_px = args.img_size * args.img_size
        m = sc['mesh']
        T_tex = sum(int(t.shape[-3]) * int(t.shape[-2]) for t in sc['textures'])
        mesh_b = 12 * len(m['v']) + 8 * len(m['vt']) + 12 * len(m['vn']) + 36 * len(m['f_v_idx'])
        alg = {       # algorithmic bytes per VIEW: compulsory reads + writes of each fused stage
            'raster': mesh_b + 28 * P_px,                                           # idx 4 + alpha 4 + uv 8 + normal 12
            'shade_inputs': 28 * P_px + 4 * args.tex_ch * T_tex + 4 * sc['c_in'] * P_px,
            'ray_render': 4 * 3 * sc['n_rays'] * P_px + 4 * (3 * sc['n_rays'] + 6) * P_px + 4 * P_px + 12 * P_px,
        }
        acc = {}
        n_prof = 5 if extras else 0
        for s in range(n_prof):
            evs = []
            lo = (s % (args.steps + args.warmup)) * world * V
            sl = slice(lo, lo + V)
            pipe.render(poses['proj'][sl], poses['pose'][sl], poses['proj_inv'][sl], poses['R_inv'][sl], stage_events=evs)
            torch.cuda.synchronize()
            for (_, e0), (name, e1) in zip(evs[:-1], evs[1:]):
                acc[name] = acc.get(name, 0.0) + e0.elapsed_time(e1) / n_prof
        if extras:
            res['stages'] = {k: {'ms_per_step': acc[k],
                             **({'alg_bytes_per_step': alg[k] * V, 'GB/s': alg[k] * V / (acc[k] * 1e-3) / 1e9,
                                 'frac_of_hbm_peak': alg[k] * V / (acc[k] * 1e-3) / 1e9 / PEAK_HBM_GBS} if k in alg else {})}
                         for k in acc}
            res['stages']['raster']['note'] = ('not an HBM-bound stage: 65 k faces are binned and edge-tested per view '
                                               '(VALU / latency bound); the HBM figure is given for completeness')
            res['stages']['shade_inputs']['note'] = ('arithmetic-bound (PMC: vector ALUs issue 74 % of the kernel cycles, '
                                                     'profiles/README.md); the HBM figure is the SURVEY 8(d) yardstick')
            res['stages']['ray_render']['note'] = ('arithmetic-bound (PMC: vector ALUs saturated: 26 rays per pixel with '
                                                   'atan2 / acos / tanh each, profiles/README.md); the HBM figure is the '
                                                   'SURVEY 8(d) yardstick')

        def timed(p):
            def st(s):
                lo = (s % (args.steps + args.warmup)) * V
                sl = slice(lo, lo + V)
                return p.render(poses['proj'][sl], poses['pose'][sl], poses['proj_inv'][sl], poses['R_inv'][sl])
            for s in range(2):
                st(s)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for s in range(args.warmup, args.warmup + args.steps):
                st(s)
            torch.cuda.synchronize()
            return time.perf_counter() - t1

        if extras and not fast and world == 1 and V > 8:
            # the batch size of the rounds before r04, same pipeline object (a call may carry fewer views than max_views)
            def st8(s):
                lo = (s % (args.steps + args.warmup)) * 8
                sl = slice(lo, lo + 8)
                return pipe.render(poses['proj'][sl], poses['pose'][sl], poses['proj_inv'][sl], poses['R_inv'][sl])
            for s in range(2):
                st8(s)
            torch.cuda.synchronize()
            t8 = time.perf_counter()
            for s in range(args.steps):
                st8(s)
            torch.cuda.synchronize()
            t8 = time.perf_counter() - t8
            res['with_8_views_per_step'] = {'frames_per_s': args.steps * 8 / t8, 'ms_per_step': t8 / args.steps * 1e3,
                                            'note': 'the headline batch size of r02 / r03; not the headline value'}
        if extras and not fast and world == 1 and not args.tile_skip and out_tiles_per_step:
            # product-tuned configuration of RNRPipeline, reported beside the headline (frames are bit-identical /
            # equal to 1e-6): out-layer pixel tiles without a foreground pixel are not computed, and the batch is split
            # over two HIP streams so that kernel tails overlap
            pipe.skip_background_tiles = True
            dts = timed(pipe)
            frac = float(pipe.unet._tile_mask[:out_tiles_per_step].float().mean().item())
            pipe.skip_background_tiles = False
            res['with_background_tile_skip'] = {'frames_per_s': args.steps * V / dts, 'ms_per_step': dts / args.steps * 1e3,
                                                'active_out_layer_tiles_last_step': frac,
                                                'note': 'bit-identical frames; not the headline value'}
            if V >= 2:
                pipe2 = make_pipeline(sc, args, dev, V, skip_background_tiles=True, streams=2)
                dt2 = timed(pipe2)
                res['with_tile_skip_and_2_streams'] = {'frames_per_s': args.steps * V / dt2,
                                                       'ms_per_step': dt2 / args.steps * 1e3,
                                                       'note': 'RNRPipeline(streams=2, skip_background_tiles=True); not the headline value'}
                del pipe2
        if extras and not fast and world == 1 and args.precision == 'f32' and pipe.unet.conv_algo != 'direct':
            # every convolution as a direct implicit GEMM (the r01 - r03 path: conv_halo_kernel only), same frames to fp32 rounding
            try:
                pd = make_pipeline(sc, args, dev, V, conv_algo='direct')
                restore_d = hook_unet_events(pd.unet, args.steps + 2)
                dtd = timed(pd)
                ums = restore_d()
                tfd = pd.unet.flops_per_view * V / (ums * 1e-3) / 1e12
                lo = (args.warmup + args.steps - 1) * V
                sl = slice(lo, lo + V)
                dimg = pd.render(poses['proj'][sl], poses['pose'][sl], poses['proj_inv'][sl], poses['R_inv'][sl])
                wimg = pipe.render(poses['proj'][sl], poses['pose'][sl], poses['proj_inv'][sl], poses['R_inv'][sl])
                res['with_direct_convolutions'] = {
                    'frames_per_s': args.steps * V / dtd, 'ms_per_step': dtd / args.steps * 1e3,
                    'max_abs_diff_vs_headline_frames': float((dimg - wimg).abs().max()),
                    'roofline': {'bound': 'mfma', 'kernel': 'conv_halo_kernel (U-Net stage, HIP events)', 'achieved': tfd, 'peak': peak_tf,
                                 'unit': 'TFLOP/s', 'frac': tfd / peak_tf, 'stage_ms_per_step': ums, **sustained_block('f32', tfd)},
                    'note': "RNRPipeline(conv_algo='direct'): every layer a direct implicit GEMM on the f32 MFMA; not the headline value"}
                del pd
            except Exception as e:          # noqa: BLE001 - an extra must never fail the bench line
                res['with_direct_convolutions'] = {'error': str(e)[:200]}
        if extras and not fast and world == 1 and args.precision == 'f32' and pipe.unet.conv_algo == 'winograd4':
            # the r03 product path: F(2x2, 3x3) / F(2x2, 2x2) only (no F(4x4, 3x3) layers)
            try:
                p2 = make_pipeline(sc, args, dev, V, conv_algo='winograd')
                restore_2 = hook_unet_events(p2.unet, args.steps + 2)
                dt2 = timed(p2)
                ums2 = restore_2()
                res['with_winograd_f2x2_only'] = {
                    'frames_per_s': args.steps * V / dt2, 'ms_per_step': dt2 / args.steps * 1e3, 'unet_stage_ms_per_step': ums2,
                    'note': "RNRPipeline(conv_algo='winograd'): the r03 product path, every Winograd layer F(2x2, .); not the headline value"}
                del p2
            except Exception as e:          # noqa: BLE001 - an extra must never fail the bench line
                res['with_winograd_f2x2_only'] = {'error': str(e)[:200]}
        if extras and not fast and world == 1:
            # the same steps with two of them in flight (RNRPipeline(inflight=2).submit: private activations per slot, the
            # non-conv stages and kernel tails of one step run under the convolutions of the other); full compute, same frames
            pf = make_pipeline(sc, args, dev, V, inflight=2)

            def sub(s):
                lo = (s % (args.steps + args.warmup)) * V
                sl = slice(lo, lo + V)
                return pf.submit(poses['proj'][sl], poses['pose'][sl], poses['proj_inv'][sl], poses['R_inv'][sl])
            for s in range(4):
                sub(s)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for s in range(args.warmup, args.warmup + args.steps):
                sub(s)
            torch.cuda.synchronize()
            dtf = time.perf_counter() - t1
            res['with_two_steps_in_flight'] = {'frames_per_s': args.steps * V / dtf, 'ms_per_step': dtf / args.steps * 1e3,
                                               'note': 'RNRPipeline(inflight=2).submit, %d views per call, everything else as the '
                                                       'headline; not the headline value' % V}
            del pf
        if extras and not fast and world == 1 and args.precision == 'f32':
            # the ray renderer folded into the out layer (RNRPipeline(fuse_ray=True): rnr_ray_weights + rnr_conv2d_ray).  Measured
            # and NOT the default: the weights kernel costs more than ray_render_kernel did
            try:
                pr = make_pipeline(sc, args, dev, V, fuse_ray=True)
                dtr = timed(pr)
                res['with_ray_renderer_in_out_layer_epilogue'] = {
                    'frames_per_s': args.steps * V / dtr, 'ms_per_step': dtr / args.steps * 1e3,
                    'note': 'RNRPipeline(fuse_ray=True): no ray_render_kernel, the out layer writes 12 B/px instead of 320; '
                            'everything else as the headline; slower than the separate kernel, not the headline value'}
                del pr
            except Exception as e:          # noqa: BLE001 - an extra must never fail the bench line
                res['with_ray_renderer_in_out_layer_epilogue'] = {'error': str(e)[:200]}
        emu_last = {}
        if extras and not fast and world == 1 and args.precision == 'f32':
            # fp32 emulated on the 16-bit matrix cores (RNR_CONV_F32_EMU_BF16X6 / _F16X3): opt-in configurations of the same
            # pipeline, each with its OWN roofline block — the peak is the dense 16-bit MFMA rate divided by the partial
            # products per multiply-add (2.5 PFLOP/s / 6 resp. / 3), the achieved figure the same algorithmic fp32 FLOPs
            lo = (args.warmup + args.steps - 1) * V
            sl = slice(lo, lo + V)
            native = pipe.render(poses['proj'][sl], poses['pose'][sl], poses['proj_inv'][sl], poses['R_inv'][sl]).clone()
            for prec, products, what in [
                    ('bf16x6', 6, 'every conv operand split exactly into 3 bf16 terms, 6 partial products accumulated in fp32 on '
                                  'v_mfma_f32_32x32x16_bf16'),
                    ('f16x3', 3, 'every conv operand split into 2 fp16 terms (22 significand bits, weights pre-scaled per layer by '
                                 'a power of two), 3 partial products accumulated in fp32 on v_mfma_f32_32x32x16_f16; error vs '
                                 'float64 0.32-0.87x the exact-fp32 kernels on 21 of 22 layer shapes, 1.01x on the last (profiles/archive/r02_emu_layer_table.md)')]:
                pe = make_pipeline(sc, args, dev, V, skip_background_tiles=False, precision=prec)
                emu_img = pe.render(poses['proj'][sl], poses['pose'][sl], poses['proj_inv'][sl], poses['R_inv'][sl])
                diff = float((emu_img - native).abs().max())
                emu_last[prec] = emu_img[V - 1:V].clone()
                for s0 in range(2):     # warm-up outside the hooked region
                    pe.render(poses['proj'][:V], poses['pose'][:V], poses['proj_inv'][:V], poses['R_inv'][:V])
                restore_e = hook_unet_events(pe.unet, args.steps)      # U-Net stage by HIP events on the launch stream, as for the headline
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for s0 in range(args.warmup, args.warmup + args.steps):
                    lo2 = (s0 % (args.steps + args.warmup)) * V
                    sl2 = slice(lo2, lo2 + V)
                    pe.render(poses['proj'][sl2], poses['pose'][sl2], poses['proj_inv'][sl2], poses['R_inv'][sl2])
                torch.cuda.synchronize()
                dte = time.perf_counter() - t1
                ums = restore_e()
                peak = 2500.0 / products
                tf = pe.unet.flops_per_view * V / (ums * 1e-3) / 1e12
                res['with_f32_emulation_' + prec] = {
                    'frames_per_s': args.steps * V / dte, 'ms_per_step': dte / args.steps * 1e3,
                    'max_abs_diff_vs_f32_mfma_frames': diff,
                    'roofline': {'bound': 'mfma', 'kernel': 'conv_halo_emu_kernel<%s> (U-Net stage, HIP events)' % prec,
                                 'achieved': tf, 'peak': peak, 'unit': 'TFLOP/s (fp32-equivalent: algorithmic FLOPs of the exact '
                                 'convolution; peak = 2500 dense 16-bit MFMA TFLOP/s / %d partial products)' % products,
                                 'frac': tf / peak, 'stage_ms_per_step': ums, **sustained_block(prec, tf)},
                    'note': 'RNRPipeline(precision="%s"): %s; full compute on every pixel, one stream; not the headline value'
                            % (prec, what)}
                if prec == 'f16x3':
                    pe.skip_background_tiles = True
                    dt4 = timed(pe)
                    res['with_all_opt_in_fast_paths'] = {
                        'frames_per_s': args.steps * V / dt4, 'ms_per_step': dt4 / args.steps * 1e3,
                        'note': 'RNRPipeline(precision="f16x3", skip_background_tiles=True), one stream; not the headline value'}
                del pe
        if world == 1 and extras and (not fast or os.environ.get('RNR_BENCH_SINGLE') == '1'):
            res['single_view_mode'] = single_view_block(sc, args, dev)
        if world == 1 and extras and not fast and args.precision == 'f32' and args.img_size == 512:
            try:
                res['config5_shape'] = config5_block(args, dev)
            except Exception as e:          # noqa: BLE001 - an extra must never fail the bench line
                res['config5_shape'] = {'error': repr(e)[:300]}
        if world == 1 and extras and not fast and not args.no_dropin_loop:
            try:
                res['dropin_view_loop'] = dropin_view_loop_block(sc, args, dev)
            except Exception as e:          # noqa: BLE001 - an extra must never fail the bench line
                res['dropin_view_loop'] = {'error': repr(e)[:300]}
        if not args.no_cpu_baseline and world == 1:
            last_id = int(ids[(args.warmup + args.steps - 1) * V + V - 1])
            hip_last = None if args.no_parity else last_frame
            cb, parity = cpu_baseline(sc, args, last_id, hip_last, emu_last)
            res['cpu_baseline'] = cb
            if parity:
                res['parity'] = parity
        else:
            res['cpu_baseline'] = None
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        ctypes.CDLL(None).fflush(None)     # RCCL's banner sits in C stdio's buffer when stdout is a pipe
        print(json.dumps(res), flush=True)
    return res


if __name__ == '__main__':
    main()
