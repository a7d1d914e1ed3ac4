"""CPU, world_size 2, gloo: the N>1 path of the renderer (view sharding + the single all-gather).  The render
function here is a deterministic stand-in (the HIP path needs a GPU); what is tested is the partitioning, the ragged
padding, ordering and that every rank ends up with every frame."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_render(v):
    # a frame that encodes its pose uniquely: [b,3,4,4]
    p = v['pose'][:, :3, 3]
    return p[:, :, None, None].expand(-1, -1, 4, 4).contiguous() * 2.0 + 1.0


def _worker(rank, world, port, B, q):
    sys.path.insert(0, os.path.join(ROOT, 'relightable-nr_amd'))
    from rnr_amd import dist as rdist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)
        views = {'pose': torch.randn(B, 4, 4, generator=g), 'proj': torch.randn(B, 3, 3, generator=g)}
        out = rdist.render_views_sharded(_fake_render, views)
        ref = _fake_render(views)
        ok = torch.equal(out, ref)
        lo, hi = rdist.shard_bounds(B, world, rank)
        local = rdist.render_views_sharded(_fake_render, views, gather=False)
        ok = ok and ((local is None and hi == lo) or torch.equal(local, ref[lo:hi]))
        # non-float32 frames with an empty shard on one rank: the dtype must be agreed, not assumed (uint8 frames)
        as_u8 = lambda v: (_fake_render(v).abs() * 10).clamp(max=255).to(torch.uint8)
        out8 = rdist.render_views_sharded(as_u8, views)
        ok = ok and out8.dtype == torch.uint8 and torch.equal(out8, as_u8(views))
        # caller-stated shape / dtype: no agreement round
        outh = rdist.render_views_sharded(lambda v: _fake_render(v).half(), views, frame_shape=(3, 4, 4), frame_dtype=torch.float16)
        ok = ok and outh.dtype == torch.float16 and torch.equal(outh, ref.half())
        q.put((rank, bool(ok), (lo, hi)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('B', [8, 5, 1])
def test_sharded_render_gloo(B):
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000) + B
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    bounds = sorted(b for _, _, b in res)
    assert bounds[0][0] == 0 and bounds[-1][1] == B and bounds[0][1] == bounds[1][0]


def _worker_overlap(rank, world, port, steps, q):
    sys.path.insert(0, os.path.join(ROOT, 'relightable-nr_amd'))
    from rnr_amd import dist as rdist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        b = 3
        frame = lambda s, r: torch.full((b, 2, 4, 4), float(100 * s + r)) + torch.arange(b, dtype=torch.float32)[:, None, None, None]
        g = rdist.OverlappedFrameGather(world, (b, 2, 4, 4), torch.float32, 'cpu')
        mine = [torch.empty(b, 2, 4, 4) for _ in range(2)]          # two frame buffers used in turn, like RNRPipeline
        ok, seen = True, []
        for s in range(steps):
            buf = mine[s % 2]
            buf.copy_(frame(s, rank))
            done = g.submit(buf)
            if done is not None:                                     # the gather of step s-1 has completed
                want = torch.cat([frame(s - 1, r) for r in range(world)], 0)
                ok = ok and torch.equal(done, want)
                seen.append(s - 1)
        last = g.drain()
        ok = ok and torch.equal(last, torch.cat([frame(steps - 1, r) for r in range(world)], 0))
        q.put((rank, bool(ok), seen))
    finally:
        dist.destroy_process_group()


def test_overlapped_frame_gather_gloo():
    """The asynchronous, double-buffered frame all-gather bench.py uses for N > 1: every step's frames arrive intact and
    in order although step s+1 overwrites the other frame buffer while the gather of step s is in flight."""
    world, steps = 2, 5
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000) + 77
    procs = [ctx.Process(target=_worker_overlap, args=(r, world, port, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert all(seen == list(range(steps - 1)) for _, _, seen in res), res


def test_shard_bounds_properties():
    sys.path.insert(0, os.path.join(ROOT, 'relightable-nr_amd'))
    from rnr_amd.dist import shard_bounds
    for n in (0, 1, 7, 64, 720):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker_mismatch(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, 'relightable-nr_amd'))
    from rnr_amd import dist as rdist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        views = {'pose': torch.randn(4, 4, 4)}
        # rank 1 renders half-precision frames although the caller stated float32: BOTH ranks must raise (before the fix only
        # rank 1 did and rank 0 hung in the all-gather)
        render = (lambda v: _fake_render(v).half()) if rank == 1 else _fake_render
        try:
            rdist.render_views_sharded(render, views, frame_shape=(3, 4, 4), frame_dtype=torch.float32)
            q.put((rank, 'no error'))
        except ValueError as e:
            q.put((rank, 'raised: ' + str(e)[:40]))
    finally:
        dist.destroy_process_group()


def test_frame_mismatch_on_one_rank_raises_everywhere():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000) + 131
    procs = [ctx.Process(target=_worker_mismatch, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(msg.startswith('raised') for _, msg in res), res
