"""-m gpu: the multi-GPU launch path on the one GPU this suite gets: `python -m torch.distributed.run --nproc-per-node 1
bench.py` with the RCCL path forced on (backend "nccl" = RCCL).  What is checked: process-group start-up on the GPU,
the asynchronous double-buffered `all_gather_into_tensor` of frames (rnr_amd.dist.OverlappedFrameGather) delivering
bit-identical frames, the bench JSON contract under the launcher.  Scaling numbers are the driver's to measure."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env):
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith('{') and '"metric"' in l]
    assert lines, p.stdout[-2000:]
    return json.loads(lines[-1])


def test_bench_under_torchrun_rccl_gather():
    env = dict(os.environ, RNR_BENCH_FORCE_DIST='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    port = 29600 + os.getpid() % 300
    res = _run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr',
                '127.0.0.1', '--master-port', str(port), 'bench.py', '--gpus', '1', '--steps', '2', '--warmup', '1',
                '--views-per-step', '2', '--no-cpu-baseline', '--main-loop-only', '--check-gather'], env)
    assert res['n_gpus'] == 1 and res['steps'] == 2 and res['warmup'] == 1
    assert res['gather_check']['ok'] is True, res['gather_check']
    assert res['gather_check']['backend'] == 'nccl'
    assert res['gather_check']['gathered_shape'] == [2, 3, 512, 512]
    assert res['value'] > 10.0 and res['scaling'] == 'weak'
    assert 0.1 < res['roofline']['frac'] <= 1.0 and res['cpu_baseline'] is None and res['n_ranks_seen'] == 1
    # r06: three windows of exactly K steps, value = the median one, and its per-step attribution (HIP events on the launch stream)
    ws = res['windows']
    assert len(ws) == 3 and sum(w['is_value'] for w in ws) == 1 and 'median of 3' in res['value_policy']
    assert sorted(w['frames_per_s'] for w in ws)[1] == [w for w in ws if w['is_value']][0]['frames_per_s'] == res['value']
    ss = res['step_series']
    assert len(ss['step_ms']) == 2 and len(ss['gap_ms']) == 1 and len(ss['unet_ms']) == 2 and len(ss['host_enqueue_ms']) == 2
    assert all(u < t for u, t in zip(ss['unet_ms'], ss['step_ms'])) and ss['ms_per_step_min'] <= ss['ms_per_step_median'] <= ss['ms_per_step_max']
    assert ss['non_unet_ms_per_step'] > 0 and isinstance(ss['slow_steps'], list) and res['prewarm_steps'] >= 1
    # r06: what this box sustains on the U-Net's MFMA instruction, measured around the timed windows (one figure per rank)
    bc = res['box_calibration']
    assert len(bc['tflops_per_rank_before_windows']) == len(bc['tflops_per_rank_after_windows']) == 1
    assert 0.5 < bc['frac_of_nominal'] < 1.02 and bc['nominal_tflops'] == 157.3
    assert len(bc['hbm_copy_GBps_per_rank_before_windows']) == 1 and 1000.0 < bc['hbm_copy_GBps_per_rank_after_windows'][0] < 8000.0
    assert res['roofline']['frac'] < bc['frac_of_nominal']          # no kernel beats the register-resident loop


def test_bench_plain_launch_forced_dist():
    """Same path without the launcher (RANK/WORLD_SIZE absent): bench.py must default to a 1-rank group on 127.0.0.1."""
    env = dict(os.environ, RNR_BENCH_FORCE_DIST='1', HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_PORT=str(29900 + os.getpid() % 90))
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        env.pop(k, None)
    res = _run([sys.executable, 'bench.py', '--steps', '2', '--warmup', '1', '--views-per-step', '1', '--no-cpu-baseline',
                '--main-loop-only', '--check-gather'], env)
    assert res['gather_check']['ok'] is True


def test_bench_single_view_block():
    """The reference's calling mode as a full bench block: `single_view_mode` carries frames/s of the sequential one-view
    calls, its own roofline (HIP events around the U-Net of every call, nominal and sustained fractions) and the throughput
    with two calls in flight, whose last frame equals the sequential one."""
    env = dict(os.environ, RNR_BENCH_FAST='1', RNR_BENCH_SINGLE='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'RNR_BENCH_FORCE_DIST'):
        env.pop(k, None)
    res = _run([sys.executable, 'bench.py', '--steps', '2', '--warmup', '1', '--views-per-step', '2', '--no-cpu-baseline',
                '--single-views', '24'], env)
    sv = res['single_view_mode']
    assert sv['views_per_call'] == 1 and sv['views'] == 24
    assert sv['frames_per_s'] > 100.0 and abs(sv['frames_per_s'] * sv['ms_per_frame'] - 1000.0) < 1.0
    r = sv['roofline']
    # frac counts what the matrix cores EXECUTE (fewer multiplications in the Winograd layers): a utilisation, <= 1 by
    # construction, nominal and sustained; the algorithmic (direct-form) figure is reported beside it
    assert r['bound'] == 'mfma' and 0.3 < r['frac'] < r['frac_of_sustained'] < 1.0 and r['peak'] == 157.3
    assert abs(r['achieved'] - r['executed_mfma_flops'] / (r['stage_ms_per_view'] * 1e-3) / 1e12) < 1e-6 * r['achieved']
    assert abs(r['effective_tflops_direct_form'] - r['alg_flops_per_view'] / (r['stage_ms_per_view'] * 1e-3) / 1e12) < 1e-6 * r['achieved']
    assert r['conv_algo'] == 'winograd4' and sum(r['layers_direct_winograd3x3_winograd2x2']) == 22 and r['layers_winograd_f4x4_3x3'] >= 4
    assert r['layers_direct_winograd3x3_winograd2x2'][1] >= 8 and r['layers_direct_winograd3x3_winograd2x2'][2] >= 3
    assert r['executed_mfma_flops'] < r['alg_flops_per_view'] and 1.5 < r['algorithmic_speedup'] < 4.0
    if r.get('executed_flops_from_pmc'):
        assert 0.95 < r['executed_flops_pmc_over_model'] < 1.05
    fly = sv['two_calls_in_flight']
    assert fly['frames_per_s'] > 100.0 and fly['max_abs_diff_vs_sequential_last_frame'] < 2e-6
    assert res['n_ranks_seen'] == 1 and 'stages' in res
