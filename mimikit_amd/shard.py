"""Clip sharding across the GPUs of one node.

Generation shards embarrassingly over clips: every clip's auto-regressive chain is
independent, so rank r of R takes a contiguous slice of the batch and the
only communication on the path is ONE broadcast of the flattened fp32 weight blob from rank 0
(RCCL over xGMI when the process group backend is "nccl"; gloo on CPU in the tests).  There is
no per-step collective.  ``gather_clips`` optionally brings the finished clips back to rank 0.
"""
from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist

__all__ = ["clip_slice", "broadcast_weights", "gather_clips", "timed_passes"]

# True: the collectives run on an initialised process group of ONE rank as well (bench.py --force-dist: the RCCL calls of an N-GPU run,
# exercised on a 1-GPU box); by default a single rank skips them
FORCE_COLLECTIVES = False


def _collectives(group=None) -> bool:
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return FORCE_COLLECTIVES or dist.get_world_size(group) > 1


def clip_slice(n_clips: int, rank: int, world_size: int) -> Tuple[int, int]:
    """[start, stop) of the clips owned by `rank`: contiguous, sizes differ by at most one"""
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError(f"bad rank {rank} / world size {world_size}")
    base, extra = divmod(n_clips, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def broadcast_weights(module: torch.nn.Module, src: int = 0, group=None) -> int:
    """Make every rank's parameters and buffers equal to rank `src`'s with a single broadcast of
    one flat fp32 buffer.  Returns the number of bytes broadcast."""
    tensors = [t for t in list(module.parameters()) + list(module.buffers()) if torch.is_floating_point(t)]
    if not tensors:
        return 0
    flat = torch.cat([t.detach().reshape(-1).to(torch.float32) for t in tensors])
    if _collectives(group):
        dist.broadcast(flat, src=src, group=group)
    offset = 0
    with torch.no_grad():
        for t in tensors:
            n = t.numel()
            t.copy_(flat[offset:offset + n].view_as(t).to(t.dtype))
            offset += n
    return flat.numel() * 4


def gather_clips(local: torch.Tensor, dst: int = 0, group=None) -> Optional[torch.Tensor]:
    """concatenate every rank's (clips, ...) tensor along dim 0 on rank `dst` (None elsewhere): a true gather - only `dst`
    receives data (the optional gather of the finished clips, 4.9 MB per rank at BASELINE config 4)"""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    # `dst` is a GLOBAL rank (what torch.distributed.gather takes); inside a sub-group the group rank differs from it
    world, is_dst = dist.get_world_size(group), dist.get_rank() == dst
    n_local = torch.tensor([local.size(0)], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=local.device) for _ in range(world)] if is_dst else None
    dist.gather(n_local, sizes, dst=dst, group=group)
    # gather needs one shape on every rank: all pad to the largest share (a max over the ranks - clip_slice shares differ by
    # at most one clip, but the callers' tensors need not come from clip_slice)
    biggest = torch.tensor([local.size(0)], dtype=torch.int64, device=local.device)
    dist.all_reduce(biggest, op=dist.ReduceOp.MAX, group=group)
    padded = torch.zeros((int(biggest), *local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[:local.size(0)] = local
    parts: Optional[List[torch.Tensor]] = [torch.empty_like(padded) for _ in range(world)] if is_dst else None
    dist.gather(padded, parts, dst=dst, group=group)
    if not is_dst:
        return None
    return torch.cat([p[:int(n)] for p, n in zip(parts, sizes)], dim=0)


def timed_passes(one_pass: Callable[[], None], steps: int, warmup: int, sync: Callable[[], None], group=None) -> float:
    """the bench contract's timing (bench.py): `warmup` untimed passes, then EXACTLY `steps` passes between two fences
    (device sync + barrier + device sync), elapsed time = MAX over ranks.  `sync` waits for the local device."""
    import time
    multi = _collectives(group)

    def fence():
        sync()
        if multi:
            dist.barrier(group=group)
        sync()

    for _ in range(warmup):
        one_pass()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        one_pass()
    fence()
    elapsed = time.perf_counter() - t0
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64)
        backend = dist.get_backend(group)
        if backend == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        elapsed = float(t.item())
    return elapsed
