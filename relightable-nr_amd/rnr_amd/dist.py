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


def _dev(views):
    return next(iter(views.values())).device


def broadcast_state(tensors, src=0, group=None):
    """One-time replication of weights / textures / mesh from `src` (start-up only)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        for t in tensors:
            dist.broadcast(t, src=src, group=group)
    return tensors
