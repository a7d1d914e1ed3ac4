"""CPU, world_size 2, gloo: bench.py's OWN control flow for N > 1 — pose slicing (s * world + rank) * V, the overlapped
double-buffered frame all-gather, barrier + MAX-reduced timing, rank-0-only JSON, --check-gather — driven through
`bench.py --stub-pipeline` (a stand-in pipeline whose frames encode their pose; the HIP path needs a GPU).  What an 8-GPU
launch adds to this is the RCCL transport itself.  Plus: two processes that import rnr_amd._lib while librnr_hip.so is
missing build it exactly once (the start-up race of `torchrun --nproc-per-node N` on a source checkout)."""
import json
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(world, extra, port):
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port))
        env.pop('RNR_BENCH_FORCE_DIST', None)
        procs.append(subprocess.Popen([sys.executable, 'bench.py', '--gpus', str(world), '--stub-pipeline', '--img-size', '16']
                                      + extra, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    return [so for so, _ in outs]


@pytest.mark.parametrize('V,steps,warm', [(2, 3, 1), (1, 4, 0), (3, 2, 2)])
def test_bench_control_flow_world2_gloo(V, steps, warm):
    port = 29700 + (os.getpid() % 200) + 7 * V
    outs = _launch(2, ['--steps', str(steps), '--warmup', str(warm), '--views-per-step', str(V), '--check-gather'], port)
    lines0 = [l for l in outs[0].splitlines() if l.startswith('{')]
    assert len(lines0) == 1, outs[0]
    assert not [l for l in outs[1].splitlines() if l.startswith('{')], 'only rank 0 prints the JSON line'
    res = json.loads(lines0[0])
    assert res['stub'] is True and res['n_gpus'] == 2 and res['n_ranks_seen'] == 2
    assert res['steps'] == steps and res['warmup'] == warm and res['scaling'] == 'weak'
    assert res['config']['views_per_step_per_gpu'] == V
    # the multi-GPU line names ITS config: BASELINE configs[3] with the global batch stated, and the exact
    # configs[3] per-GPU batch timed beside it (the stub stands in with V // 2 views per rank)
    assert res['config']['global_views_per_step'] == 2 * V
    assert 'configs[3]' in res['config']['workload'] and 'configs[2]' not in res['config']['workload']
    assert 'GLOBAL batch = %d views per step' % (2 * V) in res['config']['workload']
    if V > 1:
        w8 = res['with_8_views_per_gpu']
        assert w8['views_per_step_per_gpu'] == max(1, V // 2) and w8['global_views_per_step'] == 2 * max(1, V // 2)
        assert abs(w8['frames_per_s'] - steps * w8['global_views_per_step'] / (w8['ms_per_step'] * 1e-3 * steps)) < 1e-6 * w8['frames_per_s']
    else:
        assert 'with_8_views_per_gpu' not in res
    # value is the whole-job aggregate over both ranks on the MAX-reduced time
    assert abs(res['value'] - steps * 2 * V / (res['ms_per_step'] * 1e-3 * steps)) < 1e-6 * res['value']
    # r06: the value is one window of EXACTLY `steps` steps (the stub runs a single window), listed with its policy, and the line
    # carries the per-step series (host enqueue times on the stub; HIP-event intervals on a GPU) and the agreed pre-warm count
    assert 'exactly %d steps' % steps in res['value_policy'] and len(res['windows']) == 1 and res['windows'][0]['is_value'] is True
    assert abs(res['windows'][0]['frames_per_s'] - res['value']) < 1e-9 * res['value']
    assert len(res['step_series']['host_enqueue_ms']) == steps and res['prewarm_steps'] >= 1
    gc = res['gather_check']
    assert gc['ok'] is True and gc['ranks_with_mismatch'] == 0 and gc['backend'] == 'gloo'
    assert gc['gathered_shape'] == [2 * V, 3, 16, 16]


def _plain_env():
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'RNR_BENCH_FORCE_DIST'):
        env.pop(k, None)
    return env


def test_plain_launch_with_gpus_2_starts_two_ranks():
    """`python bench.py --gpus 2` with NO launcher and no rank environment (the form the driver uses for N = 1): the file starts
    the two ranks itself; the JSON says n_gpus 2 and n_ranks_seen 2 (counted through the process group) and the gathered frame
    buffer holds both ranks' poses (--gpus used to be parsed and ignored — one rank, n_gpus 1)."""
    p = subprocess.run([sys.executable, 'bench.py', '--gpus', '2', '--stub-pipeline', '--img-size', '16', '--steps', '3',
                        '--warmup', '1', '--views-per-step', '2', '--check-gather'], cwd=ROOT, env=_plain_env(),
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, p.stdout
    res = json.loads(lines[0])
    assert res['n_gpus'] == 2 and res['n_ranks_seen'] == 2 and res['stub'] is True
    assert res['gather_check']['ok'] is True and res['gather_check']['gathered_shape'] == [4, 3, 16, 16]
    assert abs(res['value'] - 3 * 2 * 2 / (res['ms_per_step'] * 1e-3 * 3)) < 1e-6 * res['value']


def test_plain_launch_with_gpus_1_is_one_process():
    p = subprocess.run([sys.executable, 'bench.py', '--gpus', '1', '--stub-pipeline', '--img-size', '16', '--steps', '2',
                        '--warmup', '1', '--views-per-step', '2'], cwd=ROOT, env=_plain_env(), capture_output=True, text=True,
                       timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][0])
    assert res['n_gpus'] == 1 and res['n_ranks_seen'] == 1
    assert 'configs[2]' in res['config']['workload'] and res['config']['global_views_per_step'] == 2
    assert 'with_8_views_per_gpu' not in res


def test_launcher_world_size_must_equal_gpus():
    """A launcher that started another number of ranks than --gpus says is an error exit, never a bench line."""
    env = dict(_plain_env(), RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT='29691')
    p = subprocess.run([sys.executable, 'bench.py', '--gpus', '2', '--stub-pipeline', '--img-size', '16', '--steps', '1',
                        '--warmup', '0'], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and 'WORLD_SIZE=1' in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith('{')]


def test_plain_launch_propagates_a_failing_rank():
    """The ranks of a self-launched job die (a negative image size passes the launcher's argparse and raises in the ranks when
    they allocate their frame buffers): the launcher returns non-zero and prints no JSON line."""
    p = subprocess.run([sys.executable, 'bench.py', '--gpus', '2', '--stub-pipeline', '--img-size', '-1', '--steps', '1',
                        '--warmup', '0'], cwd=ROOT, env=_plain_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.startswith('{')]


def test_bench_gathered_frames_are_the_right_poses():
    """The frames rank r contributes at step s are poses (s * world + r) * V ... + V of the shared pose list, and the gathered
    buffer holds rank 0's block first: checked against the stub's pose encoding, inside one process group of two ranks."""
    code = r'''
import os, sys, json
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'relightable-nr_amd'))
import bench
from rnr_amd import scene
from rnr_amd.dist import OverlappedFrameGather
rank, world, V, S = int(os.environ['RANK']), 2, 2, 8
dist.init_process_group('gloo', rank=rank, world_size=world)
steps = 3
ids = (np.arange(steps * world * V) * 7) %% 720
poses = {k: torch.from_numpy(v) for k, v in scene.spiral_views(S, ids).items()}
pipe = bench._StubPipeline(V, S, torch.device('cpu'))
g = OverlappedFrameGather(world, (V, 3, S, S), torch.float32, torch.device('cpu'))
ok = True
for s in range(steps):
    lo = (s * world + rank) * V
    g.submit(pipe.render(poses['proj'][lo:lo + V], poses['pose'][lo:lo + V], poses['proj_inv'][lo:lo + V], poses['R_inv'][lo:lo + V]))
    done = g.latest
    if done is not None:            # the gather of step s - 1 (depth 2)
        ps = s - 1
        ref = bench._StubPipeline(world * V, S, torch.device('cpu'))
        sl = slice(ps * world * V, (ps + 1) * world * V)
        want = ref.render(poses['proj'][sl], poses['pose'][sl], poses['proj_inv'][sl], poses['R_inv'][sl])
        ok = ok and torch.equal(done, want)
last = g.drain()
ref = bench._StubPipeline(world * V, S, torch.device('cpu'))
sl = slice((steps - 1) * world * V, steps * world * V)
ok = ok and torch.equal(last, ref.render(poses['proj'][sl], poses['pose'][sl], poses['proj_inv'][sl], poses['R_inv'][sl]))
dist.destroy_process_group()
print('OK' if ok else 'MISMATCH')
''' % (ROOT, ROOT)
    port = 29950 + os.getpid() % 40
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, '-c', code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for p in procs:
        so, se = p.communicate(timeout=300)
        assert p.returncode == 0, se[-3000:]
        assert so.strip().endswith('OK'), so


@pytest.mark.skipif(shutil.which('hipcc') is None and not os.path.isfile('/opt/rocm/bin/hipcc'), reason='needs hipcc')
def test_missing_library_is_built_once_by_two_ranks(tmp_path):
    """A source checkout without librnr_hip.so under a 2-rank launch: both processes load the library, `make` runs once
    (rnr_amd/_lib.py serialises the ranks with an exclusive file lock and the Makefile renames the library into place).
    Only objparse.hip is really compiled here: the other objects are copied in up to date, the link is the real one."""
    dst = tmp_path / 'checkout'
    shutil.copytree(os.path.join(ROOT, 'include'), dst / 'include')
    pkg = dst / 'relightable-nr_amd'
    shutil.copytree(os.path.join(ROOT, 'relightable-nr_amd', 'rnr_amd'), pkg / 'rnr_amd', ignore=shutil.ignore_patterns('__pycache__'))
    shutil.copytree(os.path.join(ROOT, 'relightable-nr_amd', 'csrc'), pkg / 'csrc', ignore=shutil.ignore_patterns('.build.lock'))
    build = pkg / 'csrc' / 'build'
    if not (build / 'conv.o').is_file():
        pytest.skip('in-tree objects not built (run __graft_entry__.build() first)')
    os.remove(build / 'objparse.o')         # one cheap translation unit is out of date; make compiles it and links
    for f in build.iterdir():
        if f.name != 'objparse.o':
            os.utime(f)                      # newer than the copied sources
    assert not (pkg / 'librnr_hip.so').exists()
    code = r'''
import os, sys, subprocess, json
sys.path.insert(0, %r)
calls = []
real_run = subprocess.run
def counting_run(cmd, *a, **k):
    if cmd and cmd[0] == 'make':
        calls.append(cmd)
    return real_run(cmd, *a, **k)
subprocess.run = counting_run
from rnr_amd import _lib
L = _lib.load()
print(json.dumps({'make_calls': len(calls), 'abi': L.rnr_abi_version(), 'path': _lib.LIB_PATH}))
''' % str(pkg)
    env = dict(os.environ)
    env.pop('RNR_HIP_LIB', None)
    procs = [subprocess.Popen([sys.executable, '-c', code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for _ in range(2)]
    res = []
    for p in procs:
        so, se = p.communicate(timeout=900)
        assert p.returncode == 0, se[-3000:]
        res.append(json.loads(so.strip().splitlines()[-1]))
    assert all(r['abi'] == 1 and r['path'].startswith(str(pkg)) for r in res)
    assert sum(r['make_calls'] for r in res) == 1, res
    assert (pkg / 'librnr_hip.so').is_file()
