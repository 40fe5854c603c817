"""Clip sharding across the GPUs of one node.

Generation shards embarrassingly over clips: every clip's auto-regressive chain is
independent, so rank r of R takes a contiguous slice of the batch and the
only communication on the path is ONE broadcast of the flattened fp32 weight blob from rank 0
(RCCL over xGMI when the process group backend is "nccl"; gloo on CPU in the tests).  There is
no per-step collective.  ``gather_clips`` optionally brings the finished clips back to rank 0.
"""
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

__all__ = ["clip_slice", "broadcast_weights", "gather_clips"]


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
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat, src=src, group=group)
    offset = 0
    with torch.no_grad():
        for t in tensors:
            n = t.numel()
            t.copy_(flat[offset:offset + n].view_as(t).to(t.dtype))
            offset += n
    return flat.numel() * 4


def gather_clips(local: torch.Tensor, dst: int = 0, group=None) -> Optional[torch.Tensor]:
    """concatenate every rank's (clips, ...) tensor along dim 0 on rank `dst` (None elsewhere)"""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    sizes = [torch.zeros(1, dtype=torch.int64, device=local.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([local.size(0)], dtype=torch.int64, device=local.device), group=group)
    biggest = int(max(int(s) for s in sizes))
    padded = torch.zeros((biggest, *local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[:local.size(0)] = local
    parts: List[torch.Tensor] = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    if dist.get_rank(group) != dst:
        return None
    return torch.cat([p[:int(n)] for p, n in zip(parts, sizes)], dim=0)
