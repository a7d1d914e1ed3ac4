"""View-sharded rendering over one node: one process per GPU, `torch.distributed` (backend "nccl" = RCCL on ROCm,
xGMI underneath; "gloo" for the CPU tests).  The path has exactly one exchange: an all-gather of rendered frames.

There is no reference counterpart (the reference is single-process); partitioning is by camera view
because views are independent — weights/textures/mesh/lighting are read-only and BatchNorm statistics are per view.
"""
import torch
import torch.distributed as dist


def shard_bounds(num_items, world_size, rank):
    """Contiguous, balanced slice [lo, hi) of `num_items` for `rank`; the first (num_items % world) ranks get one more."""
    base, extra = divmod(int(num_items), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def render_views_sharded(render_fn, views, group=None, gather=True):
    """views: dict of per-view tensors with a leading view dimension B (proj, pose, proj_inv, R_inv ...).
    render_fn(view_slice_dict) -> frames [b, ...] for the local slice (b may be 0).
    Returns frames for ALL B views on every rank (gather=True) or the local slice.

    Ragged shards (B not divisible by the world size) are padded to the largest shard for the collective and trimmed.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    B = next(iter(views.values())).shape[0]
    lo, hi = shard_bounds(B, world, rank)
    local = {k: v[lo:hi] for k, v in views.items()}
    frames = render_fn(local) if hi > lo else None
    if not gather or world == 1:
        return frames
    max_b = (B + world - 1) // world
    # every rank needs the frame shape even if its shard is empty
    shape = torch.zeros(8, dtype=torch.int64, device=_dev(views))
    if frames is not None:
        shape[0] = frames.dim() - 1
        shape[1:frames.dim()] = torch.tensor(frames.shape[1:], dtype=torch.int64)
    dist.all_reduce(shape, op=dist.ReduceOp.MAX, group=group)
    tail = tuple(int(x) for x in shape[1:1 + int(shape[0])])
    ref = frames if frames is not None else torch.zeros((0,) + tail, device=_dev(views))
    send = torch.zeros((max_b,) + tail, dtype=ref.dtype, device=ref.device)
    send[:hi - lo] = ref
    out = torch.empty((world * max_b,) + tail, dtype=ref.dtype, device=ref.device)
    dist.all_gather_into_tensor(out, send, group=group)
    out = out.reshape((world, max_b) + tail)
    pieces = []
    for r in range(world):
        rlo, rhi = shard_bounds(B, world, r)
        pieces.append(out[r, :rhi - rlo])
    return torch.cat(pieces, 0)


class OverlappedFrameGather:
    """The all-gather of step k issued asynchronously so that it overlaps the rendering of step k+1 (RCCL runs it on its
    own stream over xGMI).  `depth` gather buffers are used in turn; `submit` retires the oldest gather before a buffer
    is reused, so a caller that alternates `depth` frame buffers of its own (RNRPipeline does, with two) never has a
    frame overwritten while it is still being sent.

        g = OverlappedFrameGather(world, frames.shape, frames.dtype, frames.device)
        for poses in steps:
            g.submit(pipe.render(*poses))      # returns at once
        all_frames = g.drain()                 # [world * b, ...] of the last step
    """

    def __init__(self, world_size, frames_shape, dtype, device, group=None, depth=2):
        self.group, self.depth = group, int(depth)
        shape = (int(world_size) * int(frames_shape[0]),) + tuple(int(x) for x in frames_shape[1:])
        self.buffers = [torch.empty(shape, dtype=dtype, device=device) for _ in range(self.depth)]
        self.pending = []          # (work, buffer), oldest first
        self.submitted = 0
        self.latest = None         # buffer of the most recently completed gather

    def submit(self, frames):
        buf = self.buffers[self.submitted % self.depth]
        work = dist.all_gather_into_tensor(buf, frames, group=self.group, async_op=True)
        self.pending.append((work, buf))
        self.submitted += 1
        while len(self.pending) >= self.depth:
            self._retire()
        return self.latest

    def _retire(self):
        work, buf = self.pending.pop(0)
        work.wait()
        self.latest = buf

    def drain(self):
        while self.pending:
            self._retire()
        return self.latest


def _dev(views):
    return next(iter(views.values())).device


def broadcast_state(tensors, src=0, group=None):
    """One-time replication of weights / textures / mesh from `src` (start-up only)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        for t in tensors:
            dist.broadcast(t, src=src, group=group)
    return tensors
