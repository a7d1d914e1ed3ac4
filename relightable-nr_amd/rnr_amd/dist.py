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


_DTYPE_CODES = [torch.float32, torch.float16, torch.bfloat16, torch.float64, torch.uint8, torch.int8, torch.int16,
                torch.int32, torch.int64, torch.bool]


def render_views_sharded(render_fn, views, group=None, gather=True, frame_shape=None, frame_dtype=None):
    """Convenience helper for ONE batch of views (ragged shards, empty shards, any frame dtype).  The steady-state
    product path — what bench.py runs — is `RNRPipeline.render` + `OverlappedFrameGather` below (fixed shard size, no
    shape exchange, gather overlapped with the next step's rendering).

    views: dict of per-view tensors with a leading view dimension B (proj, pose, proj_inv, R_inv ...).
    render_fn(view_slice_dict) -> frames [b, ...] for the local slice (b may be 0).
    Returns frames for ALL B views on every rank (gather=True) or the local slice.

    Ragged shards (B not divisible by the world size) are padded to the largest shard for the collective and trimmed.
    A rank whose shard is empty does not know the frame shape / dtype: unless the caller states them
    (frame_shape = shape of ONE frame, frame_dtype), they are agreed with one small all_reduce.
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
    if frame_shape is not None and frame_dtype is not None:
        tail, dtype = tuple(int(x) for x in frame_shape), frame_dtype
    else:
        # every rank needs the frame shape AND dtype even if its shard is empty: [ndim, d0..d5, dtype code]
        meta = torch.zeros(8, dtype=torch.int64, device=_dev(views))
        if frames is not None:
            if frames.dim() - 1 > 6 or frames.dtype not in _DTYPE_CODES:
                raise ValueError('frames of %d dims / dtype %s: pass frame_shape and frame_dtype' % (frames.dim(), frames.dtype))
            meta[0] = frames.dim() - 1
            meta[1:frames.dim()] = torch.tensor(frames.shape[1:], dtype=torch.int64)
            meta[7] = _DTYPE_CODES.index(frames.dtype) + 1
        dist.all_reduce(meta, op=dist.ReduceOp.MAX, group=group)
        meta = meta.tolist()
        if meta[7] == 0:
            return None                        # B == 0: nobody rendered anything
        tail, dtype = tuple(int(x) for x in meta[1:1 + int(meta[0])]), _DTYPE_CODES[int(meta[7]) - 1]
    # a rank whose frames disagree with the agreed shape / dtype must not be the only one to raise: the others would go on
    # into the all-gather and hang.  One more small all_reduce makes the failure collective.
    mine_bad = frames is not None and (tuple(frames.shape[1:]) != tail or frames.dtype != dtype)
    bad = torch.tensor([1 if mine_bad else 0], dtype=torch.int64, device=_dev(views))
    dist.all_reduce(bad, op=dist.ReduceOp.SUM, group=group)
    if int(bad.item()):
        raise ValueError('%d rank(s) render frames that differ from the agreed %s %s (rank %d renders %s)'
                         % (int(bad.item()), tail, dtype, rank,
                            'nothing' if frames is None else '%s %s' % (tuple(frames.shape[1:]), frames.dtype)))
    send = torch.zeros((max_b,) + tail, dtype=dtype, device=_dev(views))
    if frames is not None:
        send[:hi - lo] = frames
    ref = send
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
        """Start the all-gather of `frames` [b, ...] and return the buffer of the most recently COMPLETED gather (None
        until one has completed).  The returned tensor is one of the `depth` internal buffers: it is overwritten by the
        gather issued `depth - 1` submits later — consume or clone it before that.  `frames` itself must stay untouched
        until its gather has been retired (alternate `depth` frame buffers, as RNRPipeline does)."""
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
